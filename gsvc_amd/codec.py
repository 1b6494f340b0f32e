"""Entropy coding of quantised anchor attributes: host side of csrc/ans.hip.

``encoder_gaussian`` / ``decoder_gaussian`` keep the interface of reference utils/encodings.py:102-245 (symbols that are
already quantised to integers, the context model's ``mean`` / ``scale`` and the step ``Q`` per element, the symbol range)
— the coder behind them is this library's rANS instead of the external ``gsvc_cuda_ans.ANSCoder``, and a stream is a
``bytes`` object (optionally also written to ``file_name``) instead of a file only.

Stream layout (little endian): magic ``GSA4`` | Phi-table checksum (u32) | n (u64) | seg_len (u32) | min (i32) | max (i32) | n_seg (u64) |
seg_bytes[n_seg] (u32) | the segments back to back.
"""
from __future__ import annotations

import struct

import numpy as np
import torch

from . import _lib

MAGIC = b"GSA4"          # GSA1 / GSA2: Phi from double erfc; GSA3: Phi from the fixed-point table (checksum in the header); GSA4: frequency
                         # floor 2^-16 per symbol (the rate model's likelihood floor) instead of 2^-20
SEG_LEN = 4096          # symbols per independent segment: 64 bits of state + size per segment = 0.016 bit per symbol
_HEADER = struct.Struct("<4sIQIiiQ")


def _model(mean, scale, Q):
    mu = (mean / Q).reshape(-1).float().contiguous()
    sigma = (scale / Q).reshape(-1).float().contiguous()
    return mu, sigma


def ans_encode(symbols: torch.Tensor, mu: torch.Tensor, sigma: torch.Tensor, min_symbol: int, max_symbol: int,
               seg_len: int | None = None) -> bytes:
    """int32 symbols in [min_symbol, max_symbol] + per-symbol Normal(mu, sigma) (CUDA tensors of equal length) -> stream.

    ``seg_len`` None: segments of SEG_LEN symbols, shortened for streams that spend many bits per symbol — a segment is the
    decoder's unit of parallelism (one lane decodes it serially) and costs 9 bytes of its own (final state, size, flush), so
    a stream of ~7 bits per symbol (GSVC's scaling streams) can afford 512-symbol segments (8x the lanes, 8x shorter serial
    chains) for 2 % more bytes, while a 0.1-bit-per-symbol feature stream keeps 4 096."""
    if seg_len is None:
        first = ans_encode(symbols, mu, sigma, min_symbol, max_symbol, seg_len=SEG_LEN)
        n = max(int(symbols.numel()), 1)
        bits_per_symbol = 8.0 * len(first) / n
        short = SEG_LEN
        while short > 512 and 72.0 / (short // 2) <= 0.025 * bits_per_symbol:      # 9 bytes per segment <= 2.5 % of the stream
            short //= 2
        return first if short == SEG_LEN or n <= short else ans_encode(symbols, mu, sigma, min_symbol, max_symbol, seg_len=short)
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
    return (_HEADER.pack(MAGIC, int(L.gsvc_ans_table_checksum()), n, seg_len, int(min_symbol), int(max_symbol), n_seg) + sizes +
            out[:total].cpu().numpy().tobytes())


class DeferredChecks:
    """Error flags of decode launches whose check is postponed: a decoder that reads its error word after every launch
    synchronises 26 times per model; collecting them lets the launches queue back to back (and run side by side on several
    streams) with ONE read at the end."""

    def __init__(self):
        self.flags = []

    def add(self, err, what):
        self.flags.append((err, what))

    def check(self):
        if not self.flags:
            return
        codes = torch.stack([e.reshape(()) for e, _ in self.flags]).tolist()      # the one synchronisation
        flags, self.flags = self.flags, []
        for code, (_, what) in zip(codes, flags):
            if code != 0:
                raise _lib.GsvcError(f"{what}: corrupt stream or a model that differs from the encoder's (code {code})")


class PreparedStream:
    """A stream whose header has been parsed and validated on the host and whose payload + segment offsets are on the device."""
    __slots__ = ("n", "seg_len", "smin", "smax", "n_seg", "bytes_d", "offs_d")


def _parse(stream: bytes):
    """Header fields, segment sizes and payload of one stream; every inconsistency raises here, on the host: the kernel derives
    the segment count from (n, seg_len) and reads seg_offsets[seg + 1] for each — a header that disagrees, or sizes that run
    past the end of the stream, must never reach the device."""
    if len(stream) < _HEADER.size:
        raise _lib.GsvcError("ans_decode: truncated stream (header)")
    magic, crc, n, seg_len, smin, smax, n_seg = _HEADER.unpack_from(stream, 0)
    if magic != MAGIC:
        # GSA1-GSA3 are earlier layouts of THIS project's development rounds (the frequency floor changed with GSA4: another CDF, so an
        # old stream cannot be decoded by this table, and nothing outside the repository ever held one): refused by name, not misread
        raise _lib.GsvcError(f"ans_decode: not a {MAGIC.decode()} stream" + (f" (found {magic.decode(errors='replace')}: streams written before the "
                             "GSA4 frequency floor must be re-encoded from their model, see INTEGRATION.md section 3)" if magic[:3] == b"GSA" else ""))
    if crc != int(_lib.lib().gsvc_ans_table_checksum()):
        raise _lib.GsvcError("ans_decode: the stream was coded with a different Phi table than this build's (checksum mismatch)")
    off = _HEADER.size
    if seg_len <= 0 or n < 0 or n_seg != int(_lib.lib().gsvc_ans_segments(n, seg_len)):
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
    return (n, seg_len, smin, smax, n_seg), offsets, payload


def prepare_streams(streams, device):
    """Parse a list of streams and move ALL their payloads and segment-offset tables to the device with one copy each (a
    decoder that uploads per stream pays a blocking host-to-device copy behind every decode kernel).  Empty streams -> None."""
    parsed = [(_parse(st) if len(st) else None) for st in streams]
    pay_at, off_at, pay_len, off_len = [], [], 0, 0
    for pr in parsed:
        pay_at.append(pay_len)
        off_at.append(off_len)
        if pr is not None:
            pay_len += (pr[2].size + 15) // 16 * 16      # every payload starts 16-byte aligned
            off_len += pr[1].size
    pay = np.zeros(pay_len + 16, dtype=np.uint8)         # + 16: the kernel's aligned 16-byte reads may run past the last byte
    offs = np.zeros(max(off_len, 1), dtype=np.int64)
    for pr, pa, oa in zip(parsed, pay_at, off_at):
        if pr is not None:
            pay[pa:pa + pr[2].size] = pr[2]
            offs[oa:oa + pr[1].size] = pr[1]
    pay_d, offs_d = torch.from_numpy(pay).to(device), torch.from_numpy(offs).to(device)
    out = []
    for pr, pa, oa in zip(parsed, pay_at, off_at):
        if pr is None:
            out.append(None)
            continue
        ps = PreparedStream()
        ps.n, ps.seg_len, ps.smin, ps.smax, ps.n_seg = pr[0]
        ps.bytes_d = pay_d[pa:]
        ps.offs_d = offs_d[oa:oa + pr[1].size]
        out.append(ps)
    return out


def ans_decode(stream, mu: torch.Tensor, sigma: torch.Tensor, defer: "DeferredChecks | None" = None) -> torch.Tensor:
    """Inverse of ans_encode with the same per-symbol model; returns int32 symbols on the model's device.  ``stream``: bytes, or
    a PreparedStream (prepare_streams).  With ``defer`` the kernel's error word is not read here (no synchronisation): the
    caller runs ``defer.check()`` before trusting the symbols."""
    ps = stream if isinstance(stream, PreparedStream) else prepare_streams([stream], mu.device)[0]
    n = ps.n if ps is not None else 0
    if mu.numel() != n or sigma.numel() != n:
        raise _lib.GsvcError(f"ans_decode: stream holds {n} symbols, the model {mu.numel()}")
    dev = mu.device
    sym = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    if n == 0:
        return sym[:0]
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = torch.empty(int(_lib.lib().gsvc_ans_decode_scratch_bytes(n, ps.seg_len)), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().gsvc_ans_decode(_lib.ptr(ps.bytes_d), _lib.ptr(ps.offs_d), _lib.ptr(mu), _lib.ptr(sigma), n, ps.smin,
                                          ps.smax, ps.seg_len, _lib.ptr(sym), _lib.ptr(err), _lib.ptr(scratch),
                                          _lib.current_stream(dev)), "gsvc_ans_decode")
    if defer is not None:
        defer.add(err, "ans_decode")
    elif int(err.item()) != 0:
        raise _lib.GsvcError(f"ans_decode: corrupt stream or a model that differs from the encoder's (code {int(err.item())})")
    return sym[:n]


def ans_decode_many(jobs, defer: "DeferredChecks | None" = None):
    """``ans_decode`` of several streams in ONE launch: jobs = [(stream or PreparedStream or None, mu, sigma), ...] -> list of int32
    symbol tensors (an empty stream gives an empty tensor).  A decode launch lasts as long as one lane's serial segment
    whatever the number of streams it carries, and separate launches of a few waves each do not overlap reliably."""
    import ctypes as C
    if not jobs:
        return []
    dev = jobs[0][1].device
    raw = [j[0] for j in jobs]
    need = [i for i, st in enumerate(raw) if st is not None and not isinstance(st, PreparedStream)]
    if need:
        prep = prepare_streams([raw[i] for i in need], dev)
        for i, ps in zip(need, prep):
            raw[i] = ps
    errs = torch.zeros(len(jobs), dtype=torch.int32, device=dev)
    arr = (_lib.AnsDecodeJobC * len(jobs))()
    out, keep = [], []
    L = _lib.lib()
    for k, (ps, (_, mu, sigma)) in enumerate(zip(raw, jobs)):
        n = ps.n if ps is not None else 0
        if mu.numel() != n or sigma.numel() != n:
            raise _lib.GsvcError(f"ans_decode: stream holds {n} symbols, the model {mu.numel()}")
        sym = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        out.append(sym[:n])
        d = arr[k]
        d.n = n
        d.min_symbol, d.max_symbol, d.seg_len = (ps.smin, ps.smax, ps.seg_len) if ps is not None else (0, 1, 1)
        if n == 0:
            continue
        mu, sigma = mu.contiguous(), sigma.contiguous()
        scratch = torch.empty(int(L.gsvc_ans_decode_scratch_bytes(n, ps.seg_len)), dtype=torch.uint8, device=dev)
        keep += [mu, sigma, scratch, sym]
        d.bytes, d.seg_offsets, d.mu, d.sigma = ps.bytes_d.data_ptr(), ps.offs_d.data_ptr(), mu.data_ptr(), sigma.data_ptr()
        d.symbols, d.error_flag, d.scratch = sym.data_ptr(), errs[k:].data_ptr(), scratch.data_ptr()
    _lib.check(L.gsvc_ans_decode_many(arr, len(jobs), _lib.current_stream(dev)), "gsvc_ans_decode_many")
    for t in keep:
        t.record_stream(torch.cuda.current_stream(dev))
    if defer is not None:
        defer.add(errs.max(), "ans_decode")
    else:
        code = int(errs.max().item())
        if code != 0:
            raise _lib.GsvcError(f"ans_decode: corrupt stream or a model that differs from the encoder's (code {code})")
    return out


def decoder_gaussian_many(jobs, defer=None):
    """``decoder_gaussian`` of several streams in one launch: jobs = [(mean, scale, Q, stream), ...] -> de-quantised tensors."""
    models = []
    for mean, scale, Q, stream in jobs:
        if not isinstance(Q, torch.Tensor):
            Q = torch.full_like(mean, float(Q))
        mu, sigma = _model(mean, scale, Q)
        models.append((stream, mu, sigma))
    syms = ans_decode_many(models, defer=defer)
    return [sym.to(mean.dtype).view(mean.shape) * (Q if isinstance(Q, torch.Tensor) else float(Q))
            for sym, (mean, scale, Q, _) in zip(syms, jobs)]


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


def decoder_gaussian(mean, scale, Q, stream=None, file_name=None, min_value=None, max_value=None, defer=None):
    """reference utils/encodings.py:225-262: returns the de-quantised values ``symbols * Q`` (shape of ``mean``).  The symbol
    range travels inside the stream; ``min_value`` / ``max_value`` are accepted for signature compatibility; ``defer``: see
    ans_decode."""
    if stream is None:
        with open(file_name, "rb") as f:
            stream = f.read()
    if not isinstance(Q, torch.Tensor):
        Q = torch.full_like(mean, float(Q))
    mu, sigma = _model(mean, scale, Q)
    sym = ans_decode(stream, mu, sigma, defer=defer)      # bytes or a PreparedStream
    return sym.to(mean.dtype).view(mean.shape) * Q
