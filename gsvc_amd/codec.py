"""Entropy coding of quantised anchor attributes: host side of csrc/ans.hip.

``encoder_gaussian`` / ``decoder_gaussian`` keep the interface of reference utils/encodings.py:102-245 (symbols that are
already quantised to integers, the context model's ``mean`` / ``scale`` and the step ``Q`` per element, the symbol range)
— the coder behind them is this library's rANS instead of the external ``gsvc_cuda_ans.ANSCoder``, and a stream is a
``bytes`` object (optionally also written to ``file_name``) instead of a file only.

Stream layout (little endian): magic ``GSA1`` | n (u64) | seg_len (u32) | min (i32) | max (i32) | n_seg (u64) |
seg_bytes[n_seg] (u32) | the segments back to back.
"""
from __future__ import annotations

import struct

import numpy as np
import torch

from . import _lib

MAGIC = b"GSA1"
SEG_LEN = 4096          # symbols per independent segment: 64 bits of state + size per segment = 0.016 bit per symbol
_HEADER = struct.Struct("<4sQIiiQ")


def _model(mean, scale, Q):
    mu = (mean / Q).reshape(-1).float().contiguous()
    sigma = (scale / Q).reshape(-1).float().contiguous()
    return mu, sigma


def ans_encode(symbols: torch.Tensor, mu: torch.Tensor, sigma: torch.Tensor, min_symbol: int, max_symbol: int,
               seg_len: int = SEG_LEN) -> bytes:
    """int32 symbols in [min_symbol, max_symbol] + per-symbol Normal(mu, sigma) (CUDA tensors of equal length) -> stream."""
    if not symbols.is_cuda:
        raise _lib.GsvcError("ans_encode runs on the HIP kernels of csrc/ans.hip; CPU tensors are not supported")
    L = _lib.lib()
    dev = symbols.device
    sym = symbols.reshape(-1).to(torch.int32).contiguous()
    n = sym.numel()
    assert mu.numel() == n and sigma.numel() == n
    n_seg = int(L.gsvc_ans_segments(n, seg_len))
    cap = max(int(L.gsvc_ans_scratch_bytes(n, seg_len)), 1)
    scratch = torch.empty(cap, dtype=torch.uint8, device=dev)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    seg_bytes = torch.zeros(max(n_seg, 1), dtype=torch.int32, device=dev)
    seg_offsets = torch.zeros(n_seg + 1, dtype=torch.int64, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(L.gsvc_ans_encode(_lib.ptr(sym), _lib.ptr(mu), _lib.ptr(sigma), n, int(min_symbol), int(max_symbol), seg_len,
                                 _lib.ptr(scratch), _lib.ptr(seg_bytes), _lib.ptr(seg_offsets), _lib.ptr(out), _lib.ptr(err),
                                 _lib.current_stream(dev)), "gsvc_ans_encode")
    if int(err.item()) != 0:
        raise _lib.GsvcError(f"ans_encode: symbol outside [{min_symbol}, {max_symbol}] or zero-frequency symbol (code {int(err.item())})")
    total = int(seg_offsets[-1].item()) if n_seg else 0
    sizes = seg_bytes[:n_seg].cpu().numpy().astype("<u4").tobytes()
    return _HEADER.pack(MAGIC, n, seg_len, int(min_symbol), int(max_symbol), n_seg) + sizes + out[:total].cpu().numpy().tobytes()


def ans_decode(stream: bytes, mu: torch.Tensor, sigma: torch.Tensor) -> torch.Tensor:
    """Inverse of ans_encode with the same per-symbol model; returns int32 symbols on the model's device."""
    magic, n, seg_len, smin, smax, n_seg = _HEADER.unpack_from(stream, 0)
    if magic != MAGIC:
        raise _lib.GsvcError("ans_decode: not a GSA1 stream")
    if mu.numel() != n or sigma.numel() != n:
        raise _lib.GsvcError(f"ans_decode: stream holds {n} symbols, the model {mu.numel()}")
    L = _lib.lib()
    dev = mu.device
    off = _HEADER.size
    # the kernel derives the segment count from (n, seg_len) and reads seg_offsets[seg + 1] for each: a header that
    # disagrees, or sizes that run past the end of the stream, must never reach the device
    if seg_len <= 0 or n < 0 or n_seg != int(L.gsvc_ans_segments(n, seg_len)):
        raise _lib.GsvcError(f"ans_decode: malformed header (n={n}, seg_len={seg_len}, n_seg={n_seg})")
    if off + 4 * n_seg > len(stream):
        raise _lib.GsvcError("ans_decode: truncated stream (segment table)")
    sizes = np.frombuffer(stream, dtype="<u4", count=n_seg, offset=off).astype(np.int64)
    off += 4 * n_seg
    offsets = np.zeros(n_seg + 1, dtype=np.int64)
    np.cumsum(sizes, out=offsets[1:])
    if off + int(offsets[-1]) > len(stream):
        raise _lib.GsvcError("ans_decode: truncated stream (payload shorter than its segment table says)")
    payload = np.frombuffer(stream, dtype=np.uint8, count=int(offsets[-1]), offset=off)
    bytes_d = torch.from_numpy(payload.copy()).to(dev) if payload.size else torch.empty(1, dtype=torch.uint8, device=dev)
    offs_d = torch.from_numpy(offsets).to(dev)
    sym = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(L.gsvc_ans_decode(_lib.ptr(bytes_d), _lib.ptr(offs_d), _lib.ptr(mu), _lib.ptr(sigma), n, smin, smax, seg_len,
                                 _lib.ptr(sym), _lib.ptr(err), _lib.current_stream(dev)), "gsvc_ans_decode")
    if int(err.item()) != 0:
        raise _lib.GsvcError(f"ans_decode: corrupt stream or a model that differs from the encoder's (code {int(err.item())})")
    return sym[:n]


def encoder_gaussian(x, mean, scale, Q, min_value, max_value, file_name=None):
    """reference utils/encodings.py:102-222.  ``x``: quantised symbols (integers stored as float), same shape as ``mean`` /
    ``scale``; ``Q`` a number or a tensor of that shape.  Returns (bit_len, local_min, local_max, stream)."""
    if not isinstance(Q, torch.Tensor):
        Q = torch.full_like(mean, float(Q))
    assert x.shape == mean.shape == scale.shape == Q.shape
    local_min, local_max = int(x.min().item()), int(x.max().item())
    assert local_min >= min_value and local_max <= max_value
    if local_min == local_max:
        local_max += 1
    mu, sigma = _model(mean, scale, Q)
    stream = ans_encode(x.reshape(-1), mu, sigma, local_min, local_max)
    if file_name is not None:
        assert str(file_name).endswith(".b")
        with open(file_name, "wb") as f:
            f.write(stream)
    return 8 * len(stream), local_min, local_max, stream


def decoder_gaussian(mean, scale, Q, stream=None, file_name=None, min_value=None, max_value=None):
    """reference utils/encodings.py:225-262: returns the de-quantised values ``symbols * Q`` (shape of ``mean``).  The symbol
    range travels inside the stream; ``min_value`` / ``max_value`` are accepted for signature compatibility."""
    if stream is None:
        with open(file_name, "rb") as f:
            stream = f.read()
    if not isinstance(Q, torch.Tensor):
        Q = torch.full_like(mean, float(Q))
    mu, sigma = _model(mean, scale, Q)
    sym = ans_decode(stream, mu, sigma)
    return sym.to(mean.dtype).view(mean.shape) * Q
