"""Independent host statement of the attribute-stream coder — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference codes the quantised anchor attributes through the external ``gsvc_cuda_ans.ANSCoder`` (reference README.md:51;
call sites utils/encodings.py:102-245 ``encoder_gaussian`` / ``decoder_gaussian``, common/ans_coder.py:18-54), whose source is
not in the reference tree, so the bitstream is this library's own (gsvc_amd/csrc/ans.hip, container in gsvc_amd/codec.py).  There
is therefore no reference byte stream to pin; what CAN be pinned is that the HIP kernels implement the written specification and
nothing else — a symmetric bug or a silent format drift passes every encode -> decode round trip on the same kernels.  This file
restates that specification in plain Python integers / IEEE doubles, sharing no code with the kernels:

  model      per symbol i: mu = float32 mean / Q, sigma = float32 scale / Q, symbols are integers in [min, max], R = max - min + 1
  Phi        4 097-entry table over z in [-8, 8]: round(0.5 erfc(-z / sqrt 2) 2^32), clamped to [0, 2^32 - 1], made monotone;
             its FNV-1a checksum (over the entries' little-endian bytes) travels in the stream header
  C(s)       0 for s <= min, 2^20 for s > max, otherwise with z = ((double) s - 0.5 - mu) * (1.0 / sigma), t = (z + 8) * 256:
             p32 = table[floor t] + (((table[floor t + 1] - table[floor t]) * floor((t - floor t) 65536)) >> 16)
             (0 for t <= 0 — 2^31 for a NaN z —, 2^32 - 1 for t >= 4096);  C = ((p32 * (2^20 - F R)) >> 32) + F (s - min), F = floor_of(R) = min(16, max(1, 2^13 // R))
  coder      rANS, 32-bit state x in [2^23, 2^31), 20-bit frequencies, byte renormalisation.  A segment of seg_len symbols is coded
             from its LAST symbol to its first: while x >= 2048 freq: emit x & 255, x >>= 8;  x = (x // freq) << 20 | ... + start.
             Segment bytes = final state (big endian) followed by the emitted bytes in REVERSE order of emission (= the order the
             decoder consumes them).  Decoder: slot = x & (2^20 - 1), s = the symbol with C(s) <= slot < C(s + 1),
             x = freq (x >> 20) + slot - start, then while x < 2^23: x = x << 8 | next byte.
  container  "GSA4" | checksum u32 | n u64 | seg_len u32 | min i32 | max i32 | n_seg u64 | seg_bytes u32[n_seg] | segments  (little endian)
"""
from __future__ import annotations

import math
import struct

import numpy as np

SCALE_BITS = 20
M = 1 << SCALE_BITS
L = 1 << 23
PHI_STEPS = 4096
MAGIC = b"GSA4"
HEADER = struct.Struct("<4sIQIiiQ")


def phi_table():
    tab = []
    for i in range(PHI_STEPS + 1):
        z = -8.0 + i * (16.0 / PHI_STEPS)
        p = math.floor(0.5 * math.erfc(-z * 0.70710678118654752440) * 4294967296.0 + 0.5)
        v = 0xFFFFFFFF if p >= 4294967295.0 else (0 if p <= 0 else int(p))
        if tab and v < tab[-1]:
            v = tab[-1]
        tab.append(v)
    return tab


def table_checksum(tab) -> int:
    h = 2166136261
    for v in tab:
        for b in range(4):
            h = ((h ^ ((v >> (8 * b)) & 0xFF)) * 16777619) & 0xFFFFFFFF
    return h


_TAB = phi_table()
CHECKSUM = table_checksum(_TAB)


def floor_of(R: int) -> int:
    """Frequency floor of every symbol in units of 2^-20: 16 (the 2^-16 likelihood floor of the rate model) while the alphabet's
    floors take at most 1/128 of the mass, down to 1 for very wide alphabets (csrc/ans.hip ans_floor)."""
    return min(16, max(1, (M >> 7) // max(R, 1)))


def cdf(s: int, mu: np.float64, inv_sigma: np.float64, smin: int, smax: int) -> int:
    if s <= smin:
        return 0
    if s > smax:
        return M
    R = smax - smin + 1
    with np.errstate(all="ignore"):
        z = (np.float64(s) - np.float64(0.5) - mu) * inv_sigma
        t = (z + np.float64(8.0)) * np.float64(256.0)
    if not (t > 0.0):
        p32 = 0x80000000 if z != z else 0
    elif t >= float(PHI_STEPS):
        p32 = 0xFFFFFFFF
    else:
        i = int(t)
        f = int((t - np.float64(i)) * np.float64(65536.0))
        a, b = _TAB[i], _TAB[i + 1]
        p32 = a + (((b - a) * f) >> 16)
    F = floor_of(R)
    return ((p32 * (M - F * R)) >> 32) + F * (s - smin)


def _model(mu, sigma):
    mu = np.asarray(mu, dtype=np.float32).reshape(-1).astype(np.float64)
    with np.errstate(all="ignore"):
        inv = np.float64(1.0) / np.asarray(sigma, dtype=np.float32).reshape(-1).astype(np.float64)
    return mu, inv


def encode_segment(sym, mu, inv, smin, smax) -> bytes:
    x = L
    emitted = bytearray()
    for i in range(len(sym) - 1, -1, -1):
        s = int(sym[i])
        if s < smin or s > smax:
            raise ValueError("ans_oracle: symbol outside [min, max]")
        start = cdf(s, mu[i], inv[i], smin, smax)
        freq = cdf(s + 1, mu[i], inv[i], smin, smax) - start
        if freq <= 0 or freq > M:
            raise ValueError("ans_oracle: zero-frequency symbol")
        x_max = ((L >> SCALE_BITS) << 8) * freq
        while x >= x_max:
            emitted.append(x & 0xFF)
            x >>= 8
        x = ((x // freq) << SCALE_BITS) + (x % freq) + start
    return struct.pack(">I", x) + bytes(reversed(emitted))


def encode(symbols, mu, sigma, smin: int, smax: int, seg_len: int = 4096) -> bytes:
    """The whole container for int symbols + per-symbol float32 (mu, sigma)."""
    sym = np.asarray(symbols).reshape(-1)
    n = sym.shape[0]
    mu64, inv = _model(mu, sigma)
    assert mu64.shape[0] == n and inv.shape[0] == n
    n_seg = (n + seg_len - 1) // seg_len if n > 0 else 0
    segs = [encode_segment(sym[g * seg_len:(g + 1) * seg_len], mu64[g * seg_len:(g + 1) * seg_len], inv[g * seg_len:(g + 1) * seg_len],
                           smin, smax) for g in range(n_seg)]
    sizes = np.array([len(b) for b in segs], dtype="<u4").tobytes()
    return HEADER.pack(MAGIC, CHECKSUM, n, seg_len, smin, smax, n_seg) + sizes + b"".join(segs)


def decode_segment(buf: bytes, mu, inv, smin, smax):
    if len(buf) < 4:
        raise ValueError("ans_oracle: truncated segment")
    x = struct.unpack(">I", buf[:4])[0]
    at = 4
    out = np.empty(len(mu), dtype=np.int64)
    for i in range(len(mu)):
        slot = x & (M - 1)
        lo, hi = smin, smax                                  # largest s with C(s) <= slot (C(min) = 0, C(max + 1) = 2^20 > slot)
        while lo < hi:
            mid = (lo + hi + 1) >> 1
            if cdf(mid, mu[i], inv[i], smin, smax) <= slot:
                lo = mid
            else:
                hi = mid - 1
        start = cdf(lo, mu[i], inv[i], smin, smax)
        freq = cdf(lo + 1, mu[i], inv[i], smin, smax) - start
        if freq <= 0 or not (start <= slot < start + freq):
            raise ValueError("ans_oracle: corrupt stream (no symbol owns the slot)")
        out[i] = lo
        x = freq * (x >> SCALE_BITS) + slot - start
        while x < L:
            if at >= len(buf):
                raise ValueError("ans_oracle: truncated segment")
            x = (x << 8) | buf[at]
            at += 1
    if at != len(buf):
        raise ValueError("ans_oracle: segment longer than its symbols need")
    return out


def decode(stream: bytes, mu, sigma) -> np.ndarray:
    magic, crc, n, seg_len, smin, smax, n_seg = HEADER.unpack_from(stream, 0)
    if magic != MAGIC:
        raise ValueError("ans_oracle: not a GSA4 stream")
    if crc != CHECKSUM:
        raise ValueError("ans_oracle: the stream was coded with another Phi table")
    if seg_len <= 0 or n_seg != ((n + seg_len - 1) // seg_len if n > 0 else 0):
        raise ValueError("ans_oracle: malformed header")
    sizes = np.frombuffer(stream, dtype="<u4", count=n_seg, offset=HEADER.size)
    at = HEADER.size + 4 * n_seg
    if at + int(sizes.sum()) != len(stream):
        raise ValueError("ans_oracle: segment sizes do not add up to the stream")
    mu64, inv = _model(mu, sigma)
    assert mu64.shape[0] == n
    out = np.empty(n, dtype=np.int64)
    for g in range(n_seg):
        a, b = g * seg_len, min(n, (g + 1) * seg_len)
        out[a:b] = decode_segment(stream[at:at + int(sizes[g])], mu64[a:b], inv[a:b], smin, smax)
        at += int(sizes[g])
    return out
