"""MLP weight stage of the stream codec: 8-bit quantisation of every ``mlp*`` tensor, the non-zero
mask as packed bits through zlib, the symbols through a Huffman code.

Follows reference utils/param_utils.py:4-61 (``quantize_per_tensor`` / ``quantize_per_dimension`` / ``quantize_tensor`` /
``dequantize_tensor``), utils/mask.py:9-48 (``encode_mask`` / ``decode_mask``) and scene/gaussian_model.py:1727-1835
(``quantize_model`` / ``encode_mlp``): weight matrices are quantised row by row (first dimension), biases per tensor, to
``round((t - t_min) / (scale + 1e-19))`` with ``scale = (t_max - t_min) / 2**8`` (so symbols run 0 .. 256), zeros are left out
through the mask, and the model is left holding the de-quantised values (``replace=True``).

What differs: the reference builds its code with the third-party ``dahuffman`` package and pickles the result
(``mlp.pkl``); here the code is a canonical Huffman code built in this file (only the code LENGTHS travel) and the container
is a JSON header + raw sections (``mlp.b``) — unpickling a downloaded stream would execute whatever it contains.  The reference
has no decoder for this file (its decoder loads ``checkpoint.pth``); ``decode_mlp`` here is the inverse of ``encode_mlp``.
Host-side code: 0.3 M weights once per video, not on the fitting path.
"""
from __future__ import annotations

import heapq
import json
import struct
import zlib

import numpy as np
import torch

EPS = 1e-19
MAGIC = b"GSVM1\0"


# ---------------------------------------------------------------------------------------------- quantisation
def quantize_per_tensor(t: torch.Tensor, bit: int = 8, eps: float = EPS):
    valid_mask = (t != 0)
    t_min, t_max = float(t[valid_mask].min()), float(t[valid_mask].max())
    scale = (t_max - t_min) / 2 ** bit
    quant_t = ((t - t_min) / (scale + eps)).round()
    rescaled_t = t_min + scale * quant_t
    new_t = torch.zeros_like(t)
    new_t[valid_mask] = rescaled_t[valid_mask]
    return quant_t, valid_mask, new_t, {"t_min": t_min, "scale": scale}


def quantize_per_dimension(t: torch.Tensor, bit: int = 8, axis: int = 0, eps: float = EPS):
    assert axis == 0, "Only quantization along first dimension is supported for now"
    parts = [quantize_per_tensor(t[i:i + 1, ...], bit=bit, eps=eps) for i in range(t.size(axis))]
    meta = {"t_min": [p[3]["t_min"] for p in parts], "scale": [p[3]["scale"] for p in parts]}
    return torch.cat([p[0] for p in parts], 0), torch.cat([p[1] for p in parts], 0), torch.cat([p[2] for p in parts], 0), meta


def quantize_tensor(t, bit=16, axis=-1, eps=EPS):
    return quantize_per_tensor(t, bit, eps) if axis == -1 else quantize_per_dimension(t, bit, axis, eps)


def dequantize_tensor(quant_t, mask, meta):
    new_dims = len(quant_t.shape) - 1
    t_min = torch.tensor(meta["t_min"]).view(-1, *([1] * new_dims))
    scale = torch.tensor(meta["scale"]).view(-1, *([1] * new_dims))
    return (t_min + scale * quant_t) * mask


# ---------------------------------------------------------------------------------------------- mask bits
def mask_to_bytes(mask) -> bytes:
    """Bits MSB first, zero-padded to a whole byte (reference utils/mask.py:9-21)."""
    return np.packbits(np.asarray(mask, dtype=np.uint8).reshape(-1) != 0).tobytes()


def encode_mask(mask, level=9) -> bytes:
    return zlib.compress(mask_to_bytes(mask.cpu().numpy() if isinstance(mask, torch.Tensor) else mask), level)


def decode_mask(compressed: bytes) -> torch.Tensor:
    """All bits of the stored bytes (the padding included, as the reference returns it), as int64."""
    bits = np.unpackbits(np.frombuffer(zlib.decompress(compressed), dtype=np.uint8))
    return torch.from_numpy(bits.astype(np.int64))


# ---------------------------------------------------------------------------------------------- canonical Huffman
class HuffmanCode:
    """Canonical Huffman code over small non-negative integer symbols.  Only the code lengths define it: symbols are
    ordered by (length, symbol) and numbered consecutively, so ``lengths`` is all a decoder needs."""

    MAX_LEN = 16      # code lengths are limited (from_data): the decoder's table has 2^L entries

    def __init__(self, lengths: dict):
        self.lengths = {int(s): int(l) for s, l in lengths.items()}
        if not self.lengths or min(self.lengths.values()) < 1 or max(self.lengths.values()) > self.MAX_LEN:
            raise ValueError(f"HuffmanCode: code lengths must be 1 .. {self.MAX_LEN}")
        if sum(1 << (self.MAX_LEN - l) for l in self.lengths.values()) > (1 << self.MAX_LEN):
            raise ValueError("HuffmanCode: the lengths violate Kraft's inequality (not a prefix code)")
        order = sorted(self.lengths, key=lambda s: (self.lengths[s], s))
        self.codes, code, prev = {}, 0, self.lengths[order[0]]
        for s in order:
            code <<= self.lengths[s] - prev
            prev = self.lengths[s]
            self.codes[s] = code
            code += 1

    @classmethod
    def from_data(cls, symbols: np.ndarray):
        vals, counts = np.unique(np.asarray(symbols).reshape(-1), return_counts=True)
        if vals.size == 1:
            return cls({int(vals[0]): 1})
        heap = [(int(c), int(v), (int(v),)) for v, c in zip(vals, counts)]      # (weight, tie-break, symbols below)
        heapq.heapify(heap)
        depth = {int(v): 0 for v in vals}
        while len(heap) > 1:
            wa, ta, sa = heapq.heappop(heap)
            wb, tb, sb = heapq.heappop(heap)
            for s in sa + sb:
                depth[s] += 1
            heapq.heappush(heap, (wa + wb, min(ta, tb), sa + sb))
        return cls(cls._limit_lengths(depth, {int(v): int(c) for v, c in zip(vals, counts)}))

    @classmethod
    def _limit_lengths(cls, depth: dict, count: dict) -> dict:
        """Lengths <= MAX_LEN that still satisfy Kraft's inequality.  A very skewed histogram (a few hundred thousand quantised
        weights over 257 values) can put rare symbols deeper than the limit: those are cut to MAX_LEN and the excess is paid for
        by lengthening the deepest shorter codes of the rarest symbols by one bit at a time; then codes are shortened again
        wherever a whole bit is free (most frequent first).  Unlimited codes are returned as they are (optimal)."""
        L = cls.MAX_LEN
        if max(depth.values()) <= L:
            return depth
        if len(depth) > (1 << L):
            raise ValueError(f"HuffmanCode: {len(depth)} symbols do not fit codes of at most {L} bits")
        ln = {s: min(d, L) for s, d in depth.items()}
        kraft = sum(1 << (L - l) for l in ln.values())          # in units of 2^-L; a prefix code needs <= 2^L
        full = 1 << L
        while kraft > full:
            s = min((x for x in ln if ln[x] < L), key=lambda x: (-ln[x], count[x], x))
            kraft -= 1 << (L - ln[s] - 1)
            ln[s] += 1
        for s in sorted(ln, key=lambda x: (-count[x], x)):
            while ln[s] > 1 and kraft + (1 << (L - ln[s])) <= full:
                kraft += 1 << (L - ln[s])
                ln[s] -= 1
        return ln

    def encode(self, symbols: np.ndarray) -> bytes:
        sym = np.asarray(symbols).reshape(-1).astype(np.int64)
        if sym.size == 0:
            return b""
        top = max(self.lengths) + 1
        len_of, code_of = np.zeros(top, np.int64), np.zeros(top, np.int64)
        for s, l in self.lengths.items():
            len_of[s], code_of[s] = l, self.codes[s]
        lens = len_of[np.clip(sym, 0, top - 1)]
        codes = code_of[np.clip(sym, 0, top - 1)]
        if sym.min() < 0 or sym.max() >= top or lens.min() <= 0:
            raise ValueError("HuffmanCode.encode: symbol outside the code")
        starts = np.cumsum(lens) - lens
        total = int(starts[-1] + lens[-1])
        within = np.arange(total, dtype=np.int64) - np.repeat(starts, lens)
        bits = (np.repeat(codes, lens) >> (np.repeat(lens, lens) - 1 - within)) & 1
        return np.packbits(bits.astype(np.uint8)).tobytes()

    def decode(self, data: bytes, n: int) -> np.ndarray:
        if n == 0:
            return np.zeros(0, np.int64)
        L = max(self.lengths.values())
        tab_sym, tab_len = np.zeros(1 << L, np.int64), np.zeros(1 << L, np.int64)
        for s, l in self.lengths.items():
            lo = self.codes[s] << (L - l)
            tab_sym[lo:lo + (1 << (L - l))] = s
            tab_len[lo:lo + (1 << (L - l))] = l
        bits = np.unpackbits(np.frombuffer(data, dtype=np.uint8)).astype(np.int64)
        bits = np.concatenate([bits, np.zeros(L, np.int64)])
        window = np.zeros(bits.size - L + 1, np.int64)          # value of the L bits starting at every position
        for j in range(L):
            window = (window << 1) | bits[j:j + window.size]
        sym_at, len_at = tab_sym[window].tolist(), tab_len[window].tolist()
        out, pos = [0] * n, 0
        for i in range(n):
            if len_at[pos] == 0:
                raise ValueError("corrupt Huffman stream")
            out[i] = sym_at[pos]
            pos += len_at[pos]
        return np.asarray(out, np.int64)


# ---------------------------------------------------------------------------------------------- model level
@torch.no_grad()
def quantize_model(pc, replace=True):
    """8-bit quantisation of every tensor whose state_dict key starts with ``mlp`` (reference :1727-1764).  Returns
    (valid_mask_list, quant_weight_list, meta_info_list) and caches it on the model as the reference does."""
    cur = pc.state_dict()
    valid_mask_list, quant_weight_list, meta_info_list = [], [], []
    for k, v in cur.items():
        if not k.startswith("mlp"):
            continue
        large_tf = v.dim() in {2, 4} and "bias" not in k
        quant_t, valid_mask, new_t, meta = quantize_tensor(v, 8, 0 if large_tf else -1)
        cur[k] = new_t
        quant_weight_list.append(quant_t[valid_mask].flatten())          # only the non-zero weights are coded
        valid_mask_list.append(valid_mask.flatten())
        meta["key"], meta["shape"] = k, [int(i) for i in v.shape]
        meta_info_list.append(meta)
    if replace:
        pc.load_state_dict(cur)
    pc.quantized_model_cache = (valid_mask_list, quant_weight_list, meta_info_list)
    return pc.quantized_model_cache


def encode_mlp(pc, file_path) -> int:
    """Write the quantised MLPs (``quantize_model`` must have run); returns the file size in bits (reference :1767-1835)."""
    valid_mask_list, quant_weight_list, meta_info_list = pc.quantized_model_cache
    compressed_mask = encode_mask(torch.cat(valid_mask_list))
    sym = torch.cat(quant_weight_list).to(torch.int64).cpu().numpy()
    code = HuffmanCode.from_data(sym)
    params = code.encode(sym)
    meta = {"code_lengths": {str(s): l for s, l in code.lengths.items()}, "meta_list": meta_info_list, "n_symbols": int(sym.size)}
    compressed_meta = zlib.compress(json.dumps(meta).encode("utf-8"), 9)
    with open(file_path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<QQQ", len(compressed_meta), len(compressed_mask), len(params)))
        f.write(compressed_meta)
        f.write(compressed_mask)
        f.write(params)
    import os
    return os.path.getsize(file_path) * 8


def decode_mlp(file_path) -> dict:
    """{state_dict key: de-quantised tensor} of an ``encode_mlp`` file."""
    with open(file_path, "rb") as f:
        blob = f.read()
    if blob[:len(MAGIC)] != MAGIC:
        raise ValueError("not an MLP stream of this library")
    at = len(MAGIC)
    n_meta, n_mask, n_par = struct.unpack_from("<QQQ", blob, at)
    at += 24
    if at + n_meta + n_mask + n_par != len(blob):
        raise ValueError("MLP stream: section sizes do not add up to the file size")
    meta = json.loads(zlib.decompress(blob[at:at + n_meta]).decode("utf-8"))
    mask_bits = decode_mask(blob[at + n_meta:at + n_meta + n_mask])
    code = HuffmanCode({int(s): l for s, l in meta["code_lengths"].items()})
    sym = torch.from_numpy(code.decode(blob[at + n_meta + n_mask:], int(meta["n_symbols"]))).to(torch.float32)
    out, m_at, s_at = {}, 0, 0
    for info in meta["meta_list"]:
        shape = tuple(info["shape"])
        numel = int(np.prod(shape)) if shape else 1
        mask = mask_bits[m_at:m_at + numel].view(shape).to(torch.float32)
        m_at += numel
        n_valid = int(mask.sum())
        quant = torch.zeros(numel)
        quant[mask.flatten() > 0] = sym[s_at:s_at + n_valid]
        s_at += n_valid
        quant = quant.view(shape)
        if isinstance(info["t_min"], list):
            out[info["key"]] = dequantize_tensor(quant, mask, info)
        else:
            out[info["key"]] = (info["t_min"] + info["scale"] * quant) * mask
    if s_at != sym.numel():
        raise ValueError("MLP stream: symbol count does not match the masks")
    return out
