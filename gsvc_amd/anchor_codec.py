"""Entropy coding of the anchor geometry.

The reference hands the 16-bit anchor grid to MPEG G-PCC through the external ``tmc3`` executable (reference
utils/encodings.py:714-826, ``encode_anchor`` / ``decode_anchor``: lossless octree geometry of the quantised anchors).  ``tmc3``
is not part of the reference tree and cannot be had here, so this is an own lossless coder of the same kind and for the same
data — the integer anchor grid of ``Quantize_anchor`` — with its own bitstream (there is no G-PCC bitstream to be compatible
with on this side of the boundary; round trips and sizes are what the tests pin):

* **occupancy octree**: the points are put in Morton order; level by level every occupied node emits the 8-bit mask of its
  occupied children (nodes and children in Morton order, which is the order the decoder regenerates them in).  A node that holds a
  single point costs log2(8) = 3 bits per remaining level under the level's own symbol statistics, i.e. its raw coordinates —
  no special case needed.
* **entropy stage**: a static model per level (the histogram of that level's masks, 12-bit frequencies) and an interleaved
  rANS coder (32-bit states, 16-bit renormalisation, one lane per 256 symbols, at most 16 384) written as NumPy array operations — a level of a
  million symbols is a thousand vector steps.  Every operation is integer: the streams are exactly reproducible.
* **lattice mode**: GSVC's anchors are voxel centres (``voxel_size`` = 0.001: reference scene/gaussian_model.py:748-752 and the
  anchor growing of :1316-1449 place them on multiples of the voxel size), which the 16-bit grid then quantises with a
  non-integer number of grid steps per voxel — an octree over the 16-bit grid cannot see that regularity.  When the caller
  passes the anchors' positions and the voxel size, the coder codes the LATTICE indices (round(position / voxel_size): 9-12 bits
  per axis instead of 16) and keeps, as exceptions, the grid values of the few anchors whose lattice point does not quantise
  back to their grid value.  Uniformly scattered anchors cost about log2(lattice cells / anchors) + 2 to 3 bits each (the counting bound is + 1.44): 11.7 bits per anchor for 245 k anchors in a 64-frame 1080p cube, where the raw grid takes 48 and the same octree over the 16-bit grid 31.

Stream layout (little endian): magic "GSAO1", mode, point / unique counts, per-axis bit width, lattice parameters, then per
level: symbol count, the frequency table, lane count, final rANS states, 16-bit words.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

MAGIC = b"GSAO1"
PROB_BITS = 12
PROB_SCALE = 1 << PROB_BITS
RANS_L = 1 << 16            # lower bound of the normalised state interval [2^16, 2^32)
MAX_LANES = 16384        # one lane per 256 symbols (32 state bits per lane): a level of 4 M symbols is 256 vector steps


# ------------------------------------------------------------------------------------------------ interleaved rANS (NumPy)
def _quantise_freqs(counts: np.ndarray) -> np.ndarray:
    """Frequencies summing to 2^12 with every present symbol >= 1 (largest-remainder, deterministic)."""
    total = int(counts.sum())
    present = counts > 0
    f = np.zeros_like(counts, dtype=np.int64)
    if total == 0:
        return f
    raw = counts.astype(np.float64) * (PROB_SCALE / total)
    f[present] = np.maximum(1, np.floor(raw[present]).astype(np.int64))
    diff = PROB_SCALE - int(f.sum())
    if diff > 0:
        order = np.argsort(-(raw - np.floor(raw)), kind="stable")
        order = order[present[order]]
        f[order[:diff % len(order)]] += 1
        f[order] += diff // len(order)
    while diff < 0:
        # take from the largest (never below 1)
        i = int(np.argmax(f))
        take = min(-diff, int(f[i]) - 1)
        if take <= 0:
            raise ValueError("anchor_codec: more than 4096 distinct symbols")
        f[i] -= take
        diff += take
    assert int(f.sum()) == PROB_SCALE and (f[present] >= 1).all()
    return f


def _rans_encode(symbols: np.ndarray, freq: np.ndarray):
    """Returns (lanes, final states uint32 [lanes], words uint16 [...]) for symbols under the static model ``freq`` (sum 2^12).
    Symbol i belongs to lane i % lanes; rows of ``lanes`` symbols are coded from the last to the first."""
    n = int(symbols.size)
    # every lane ends with a 32-bit state in the stream: one lane per 256 symbols keeps that under 1/8 bit per symbol
    lanes = int(min(MAX_LANES, max(1, n // 256)))
    cum = np.concatenate([[0], np.cumsum(freq)[:-1]]).astype(np.uint64)
    f_sym = freq[symbols].astype(np.uint64)
    c_sym = cum[symbols]
    x = np.full(lanes, RANS_L, dtype=np.uint64)
    rows = (n + lanes - 1) // lanes
    chunks = []
    for r in range(rows - 1, -1, -1):
        lo = r * lanes
        k = min(lanes, n - lo)
        f, c = f_sym[lo:lo + k], c_sym[lo:lo + k]
        xs = x[:k]
        x_max = ((RANS_L >> PROB_BITS) << 16) * f                     # state bound for this frequency
        m = xs >= x_max
        if m.any():
            chunks.append((xs[m] & np.uint64(0xFFFF)).astype(np.uint16))
            xs = np.where(m, xs >> np.uint64(16), xs)
        else:
            chunks.append(np.zeros(0, np.uint16))
        x[:k] = (xs // f) * np.uint64(PROB_SCALE) + (xs % f) + c
    words = np.concatenate(chunks[::-1]) if chunks else np.zeros(0, np.uint16)
    return lanes, x.astype(np.uint32), words


def _rans_decode(n: int, freq: np.ndarray, lanes: int, states: np.ndarray, words: np.ndarray) -> np.ndarray:
    cum = np.concatenate([[0], np.cumsum(freq)]).astype(np.int64)
    slot2sym = np.repeat(np.arange(freq.size, dtype=np.int64), freq)         # 4096 entries
    f64, c64 = freq.astype(np.uint64), cum[:-1].astype(np.uint64)
    x = states.astype(np.uint64).copy()
    out = np.empty(n, dtype=np.int64)
    ptr = 0
    rows = (n + lanes - 1) // lanes
    words = words.astype(np.uint64)
    for r in range(rows):
        lo = r * lanes
        k = min(lanes, n - lo)
        xs = x[:k]
        slot = (xs & np.uint64(PROB_SCALE - 1)).astype(np.int64)
        s = slot2sym[slot]
        out[lo:lo + k] = s
        xs = f64[s] * (xs >> np.uint64(PROB_BITS)) + slot.astype(np.uint64) - c64[s]
        m = xs < np.uint64(RANS_L)
        cnt = int(m.sum())
        if cnt:
            if ptr + cnt > words.size:
                raise ValueError("anchor_codec: truncated stream")
            xs[m] = (xs[m] << np.uint64(16)) | words[ptr:ptr + cnt]
            ptr += cnt
        x[:k] = xs
    if ptr != words.size or not (x == np.uint64(RANS_L)).all():
        raise ValueError("anchor_codec: corrupt stream (the coder did not return to its initial state)")
    return out


def _pack_level(symbols: np.ndarray) -> bytes:
    counts = np.bincount(symbols, minlength=256).astype(np.int64)
    freq = _quantise_freqs(counts)
    lanes, states, words = _rans_encode(symbols, freq)
    present = np.flatnonzero(freq)
    table = np.stack([present, freq[present]], axis=1).astype(np.uint16).tobytes()
    return (struct.pack("<IHHI", symbols.size, present.size, lanes, words.size) + table + states.astype("<u4").tobytes() +
            words.astype("<u2").tobytes())


def _unpack_level(buf: memoryview, at: int, expect: int):
    n, n_present, lanes, n_words = struct.unpack_from("<IHHI", buf, at)
    at += 12
    if n != expect or lanes < 1 or lanes > MAX_LANES or n_present < 1 or n_present > 256:
        raise ValueError("anchor_codec: corrupt level header")
    table = np.frombuffer(buf, dtype="<u2", count=2 * n_present, offset=at).reshape(n_present, 2).astype(np.int64)
    at += 4 * n_present
    freq = np.zeros(256, dtype=np.int64)
    freq[table[:, 0] & 0xFF] = table[:, 1]
    if int(freq.sum()) != PROB_SCALE:
        raise ValueError("anchor_codec: corrupt frequency table")
    states = np.frombuffer(buf, dtype="<u4", count=lanes, offset=at)
    at += 4 * lanes
    words = np.frombuffer(buf, dtype="<u2", count=n_words, offset=at)
    at += 2 * n_words
    return _rans_decode(n, freq, lanes, states, words), at


# ------------------------------------------------------------------------------------------------ octree over integer points
def _spread(v: np.ndarray) -> np.ndarray:
    """Bits b of a 16-bit value -> bit 3 b (the classic magic-number spread)."""
    v = v.astype(np.uint64) & np.uint64(0xFFFF)
    v = (v | (v << np.uint64(32))) & np.uint64(0x001F00000000FFFF)
    v = (v | (v << np.uint64(16))) & np.uint64(0x001F0000FF0000FF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v


def _compact(v: np.ndarray) -> np.ndarray:
    v = v & np.uint64(0x1249249249249249)
    v = (v | (v >> np.uint64(2))) & np.uint64(0x10C30C30C30C30C3)
    v = (v | (v >> np.uint64(4))) & np.uint64(0x100F00F00F00F00F)
    v = (v | (v >> np.uint64(8))) & np.uint64(0x001F0000FF0000FF)
    v = (v | (v >> np.uint64(16))) & np.uint64(0x001F00000000FFFF)
    v = (v | (v >> np.uint64(32))) & np.uint64(0xFFFF)
    return v


def _morton(pts: np.ndarray, bits: int) -> np.ndarray:
    """Per level (from the top) the bits (x, y, z): x at position 3 b + 2, y at 3 b + 1, z at 3 b."""
    return (_spread(pts[:, 0]) << np.uint64(2)) | (_spread(pts[:, 1]) << np.uint64(1)) | _spread(pts[:, 2])


def _demorton(key: np.ndarray, bits: int) -> np.ndarray:
    return np.stack([_compact(key >> np.uint64(2)), _compact(key >> np.uint64(1)), _compact(key)], axis=1).astype(np.int64)


def _sort_keys(keys: np.ndarray) -> np.ndarray:
    """np.sort of uint64 keys below 2^63, on the GPU when there is one and the array is large (4 M keys: 0.35 s on the host)."""
    if keys.shape[0] >= (1 << 16):
        try:
            import torch
            if torch.cuda.is_available() and int(keys.max()) < (1 << 63):
                return torch.sort(torch.from_numpy(keys.astype(np.int64)).cuda()).values.cpu().numpy().astype(np.uint64)
        except Exception:  # noqa: BLE001
            pass
    return np.sort(keys)


def _occupancy_levels_gpu(keys: np.ndarray, bits: int):
    """The occupancy bytes of every octree level from the sorted Morton keys, on the GPU when there is one and the set is large (the
    host loop below takes 0.3 s at 4 M points: one flatnonzero + reduceat per level): a node's occupancy = the OR (= the sum, over the
    level's DISTINCT cells) of 1 << child over its run of keys.  Returns a list of int64 arrays, or None (no device: the host loop)."""
    if keys.shape[0] < (1 << 16) or (keys.shape[0] and int(keys.max()) >= (1 << 63)):
        return None
    try:
        import torch
        if not torch.cuda.is_available():
            return None
        k = torch.from_numpy(keys.astype(np.int64)).cuda()
        out = []
        for level in range(bits):
            shift = 3 * (bits - 1 - level)
            cell = torch.unique_consecutive(k >> shift)            # the distinct (node, child) cells of this level, in key order
            child, node = cell & 7, cell >> 3
            _, inv = torch.unique_consecutive(node, return_inverse=True)
            occ = torch.zeros(int(inv[-1]) + 1, dtype=torch.int64, device=k.device).scatter_add_(0, inv, torch.ones_like(child) << child)
            out.append(occ)
        return [o.cpu().numpy() for o in out]
    except Exception:  # noqa: BLE001
        return None


def _encode_octree(pts: np.ndarray, bits: int) -> bytes:
    """pts: distinct non-negative integer points < 2^bits per axis."""
    keys = _sort_keys(_morton(pts, bits))
    occ_levels = _occupancy_levels_gpu(keys, bits)
    if occ_levels is not None:
        return b"".join(_pack_level(occ) for occ in occ_levels)
    parts = []
    for level in range(bits):
        shift = np.uint64(3 * (bits - 1 - level))
        child = ((keys >> shift) & np.uint64(7)).astype(np.int64)
        node = keys >> (shift + np.uint64(3))
        starts = np.concatenate([[0], np.flatnonzero(node[1:] != node[:-1]) + 1])
        occ = np.bitwise_or.reduceat(np.left_shift(1, child), starts)
        parts.append(_pack_level(occ.astype(np.int64)))
    return b"".join(parts)


_BIT = np.arange(8, dtype=np.int64)


_POP = np.array([bin(i).count("1") for i in range(256)], dtype=np.int64)
_CHILDREN = np.concatenate([[c for c in range(8) if (i >> c) & 1] for i in range(256)] + [[]]).astype(np.uint64)
_CHILD_AT = np.concatenate([[0], np.cumsum(_POP)]).astype(np.int64)        # where mask i's children start in _CHILDREN


def _decode_octree(buf: memoryview, at: int, n_points: int, bits: int):
    nodes = np.zeros(1, dtype=np.uint64)
    for level in range(bits):
        occ, at = _unpack_level(buf, at, nodes.size)
        if (occ < 1).any() or (occ > 255).any():
            raise ValueError("anchor_codec: corrupt occupancy symbol")
        cnt = _POP[occ]
        total = int(cnt.sum())
        if total > n_points:
            raise ValueError("anchor_codec: corrupt stream (more nodes than points)")
        # child j of node i: the j-th set bit of its mask (table look-up instead of an [n, 8] expansion)
        first = np.cumsum(cnt) - cnt
        within = np.arange(total, dtype=np.int64) - np.repeat(first, cnt)
        child = _CHILDREN[np.repeat(_CHILD_AT[occ], cnt) + within]
        nodes = (np.repeat(nodes, cnt) << np.uint64(3)) | child
    if nodes.size != n_points:
        raise ValueError("anchor_codec: corrupt stream (point count)")
    return _demorton(nodes, bits), at


# ------------------------------------------------------------------------------------------------ public
def _lex(p: np.ndarray) -> np.ndarray:
    """Rows sorted by (x, y, z); values below 2^20 go through one packed 64-bit key (a plain sort, no argsort + gather)."""
    if p.shape[0] == 0:
        return p
    if p.min() >= 0 and p.max() < (1 << 20):
        q = p.astype(np.uint64)
        key = np.sort((q[:, 0] << np.uint64(40)) | (q[:, 1] << np.uint64(20)) | q[:, 2])
        m = np.uint64((1 << 20) - 1)
        return np.stack([key >> np.uint64(40), (key >> np.uint64(20)) & m, key & m], axis=1).astype(p.dtype)
    return p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]


def _unique_rows(p: np.ndarray):
    """(distinct rows sorted by (x, y, z), their multiplicities) of non-negative integer rows below 2^20 per column — np.unique(p,
    axis=0, return_counts=True) through ONE packed 64-bit key per row: a plain 1-D sort instead of a lexicographic one over a
    structured view (4.1 M anchors: 1.4 s of the 4K model's 3.1 s stream encode, round 6), on the GPU when there is one."""
    if p.shape[0] == 0:
        return p.reshape(0, 3), np.zeros(0, np.int64)
    if p.min() < 0 or p.max() >= (1 << 20):
        return np.unique(p, axis=0, return_counts=True)
    q = p.astype(np.uint64)
    key = ((q[:, 0] << np.uint64(40)) | (q[:, 1] << np.uint64(20)) | q[:, 2]).astype(np.int64)      # < 2^60: order-preserving as int64
    u = c = None
    if key.shape[0] >= (1 << 16):
        try:
            import torch
            if torch.cuda.is_available():
                tu, tc = torch.unique(torch.from_numpy(key).cuda(), return_counts=True)              # sorted ascending
                u, c = tu.cpu().numpy(), tc.cpu().numpy()
        except Exception:  # noqa: BLE001  (no torch / no device: the host sort below)
            u = c = None
    if u is None:
        u, c = np.unique(key, return_counts=True)
    m = np.int64((1 << 20) - 1)
    return np.stack([u >> np.int64(40), (u >> np.int64(20)) & m, u & m], axis=1).astype(np.int64), c.astype(np.int64)


def _grid_of(lattice_idx: np.ndarray, voxel_size: float, interval: np.ndarray, a_min: np.ndarray) -> np.ndarray:
    """Quantize_anchor's grid value of the lattice point (float32 arithmetic, as gsvc_amd.encodings.Quantize_anchor._grid)."""
    a = (lattice_idx.astype(np.float64) * float(voxel_size)).astype(np.float32)
    q = np.floor((a - a_min.astype(np.float32)) / interval.astype(np.float32))
    return np.clip(q, 0, 65535).astype(np.int64)


def _lattice_test_gpu(positions: np.ndarray, q: np.ndarray, voxel_size: float, interval: np.ndarray, a_min: np.ndarray):
    """(lattice index of every anchor, "its grid value is its lattice point's") with the elementwise passes on the GPU — the same
    IEEE operations as the host lines it replaces (round half to even in float64, _grid_of's float32 arithmetic): 0.25 s of the
    4 M-anchor encode.  None without a device."""
    try:
        import torch
        if not torch.cuda.is_available():
            return None
        dev = torch.device("cuda")
        idx = torch.round(torch.from_numpy(positions).to(dev) / voxel_size).to(torch.int64)
        a = (idx.to(torch.float64) * voxel_size).to(torch.float32)
        g = torch.floor((a - torch.from_numpy(a_min).to(dev)) / torch.from_numpy(interval).to(dev)).clamp_(0, 65535).to(torch.int64)
        good = (g == torch.from_numpy(q).to(dev)).all(dim=1)
        return idx.cpu().numpy(), good.cpu().numpy()
    except Exception:  # noqa: BLE001
        return None


_torch_device = None          # tests: force encode_anchors' torch path onto this device ("cpu": compared with the numpy path byte for byte)


def _cuda_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False


def _spread_t(v):
    """_spread on an int64 torch tensor (values below 2^16)."""
    v = v & 0xFFFF
    v = (v | (v << 32)) & 0x001F00000000FFFF
    v = (v | (v << 16)) & 0x001F0000FF0000FF
    v = (v | (v << 8)) & 0x100F00F00F00F00F
    v = (v | (v << 4)) & 0x10C30C30C30C30C3
    v = (v | (v << 2)) & 0x1249249249249249
    return v


def _encode_lattice_torch(positions: np.ndarray, q: np.ndarray, voxel_size: float, interval: np.ndarray, a_min: np.ndarray, device):
    """encode_anchors' lattice mode with every per-anchor pass on ``device`` (round 6: the 4 M-anchor encode spent 0.7 s in numpy
    passes over [n, 3] int64 arrays): lattice index, grid-value test, origin / span, the distinct lattice points (packed 60-bit key,
    torch.unique), their Morton keys sorted, and the occupancy bytes of every octree level.  The same integers as the numpy lines
    (IEEE float64 / float32 elementwise arithmetic, round half to even; tests/test_anchor_codec_cpu.py compares the byte streams
    with ``device='cpu'``).  Returns None when the lattice mode does not apply (the caller then takes the grid mode), else
    (origin [3] int, bits, n_uniq, dup [k, 2] int64 numpy, exceptions [m, 3] int64 numpy sorted by (x, y, z), octree bytes)."""
    import torch
    dev = torch.device(device)
    tq = torch.from_numpy(q).to(dev)
    idx = torch.round(torch.from_numpy(positions).to(dev) / voxel_size).to(torch.int64)
    a = (idx.to(torch.float64) * voxel_size).to(torch.float32)
    g = torch.floor((a - torch.from_numpy(a_min).to(dev)) / torch.from_numpy(interval).to(dev)).clamp_(0, 65535).to(torch.int64)
    good = (g == tq).all(dim=1)
    n, n_good = int(q.shape[0]), int(good.sum())
    if n_good == 0 or n_good / n < 0.9:
        return None
    ig = idx[good]
    origin = ig.min(dim=0).values
    span = ig.max(dim=0).values - origin + 1
    if int(span.max()) > (1 << 15):
        return None
    rel = ig - origin
    bits = max(1, int(rel.max()).bit_length())
    key = (rel[:, 0] << 40) | (rel[:, 1] << 20) | rel[:, 2]
    ukey, counts = torch.unique(key, return_counts=True)                  # ascending = (x, y, z) order
    m = (1 << 20) - 1
    ux, uy, uz = ukey >> 40, (ukey >> 20) & m, ukey & m
    dup_i = (counts > 1).nonzero().squeeze(1)
    dup = torch.stack([dup_i, counts[dup_i]], dim=1).cpu().numpy().astype(np.int64)
    exc = _lex(tq[~good].cpu().numpy().astype(np.int64))
    keys = torch.sort((_spread_t(ux) << 2) | (_spread_t(uy) << 1) | _spread_t(uz)).values
    parts = []
    for level in range(bits):
        shift = 3 * (bits - 1 - level)
        cell = torch.unique_consecutive(keys >> shift)
        child, node = cell & 7, cell >> 3
        _, inv = torch.unique_consecutive(node, return_inverse=True)
        occ = torch.zeros(int(inv[-1]) + 1, dtype=torch.int64, device=dev).scatter_add_(0, inv, torch.ones_like(child) << child)
        parts.append(occ)
    body = b"".join(_pack_level(o.cpu().numpy()) for o in parts)
    return [int(v) for v in origin.cpu()], bits, int(ukey.shape[0]), dup, exc, body


def encode_anchors(anchors_q: np.ndarray, positions: np.ndarray | None = None, voxel_size: float | None = None,
                   interval: np.ndarray | None = None, a_min: np.ndarray | None = None) -> bytes:
    """Lossless code of the quantised anchors ``anchors_q`` (uint16-valued [n, 3]; any order, duplicates allowed).  With
    ``positions`` (the anchors' float positions, same rows), ``voxel_size`` and the quantiser's ``interval`` / ``a_min`` the
    lattice mode is tried.  ``decode_anchors`` returns the anchors sorted by (x, y, z)."""
    q = np.asarray(anchors_q).astype(np.int64).reshape(-1, 3)
    n = q.shape[0]
    if n and (q.min() < 0 or q.max() > 65535):
        raise ValueError("anchor_codec: grid values outside 0 .. 65535")
    head = [MAGIC]
    lattice = None
    if n and positions is not None and voxel_size and interval is not None and a_min is not None:
        interval, a_min = np.asarray(interval, np.float32).reshape(3), np.asarray(a_min, np.float32).reshape(3)
        tdev = _torch_device if _torch_device is not None else ("cuda" if n >= (1 << 16) and _cuda_available() else None)
        if tdev is not None:
            try:
                got = _encode_lattice_torch(np.asarray(positions, np.float64).reshape(-1, 3), q, float(voxel_size), interval, a_min, tdev)
            except RuntimeError:          # (out of device memory, ...: the host path answers)
                got = "host"
            if got is None:
                pass                      # lattice mode does not apply: grid mode below (its own GPU sorts)
            elif got != "host":
                origin, bits, n_uniq, dup, exc, body = got
                extra = zlib.compress(dup.astype("<i8").tobytes() + exc.astype("<u2").tobytes(), 9)
                head.append(struct.pack("<BQQBI", 1, n, n_uniq, bits, len(extra)))
                head.append(struct.pack("<QQ3qd3f3f", int(dup.shape[0]), int(exc.shape[0]), *origin, float(voxel_size),
                                        *[float(v) for v in interval], *[float(v) for v in a_min]))
                return b"".join(head) + extra + body
        on_dev = None if tdev is not None else _lattice_test_gpu(np.asarray(positions, np.float64).reshape(-1, 3), q, float(voxel_size), interval, a_min) if n >= (1 << 16) else None
        if tdev is not None and got is None:
            idx = good = None             # decided on the device: no lattice mode
        elif on_dev is not None:
            idx, good = on_dev
        else:
            idx = np.round(np.asarray(positions, np.float64).reshape(-1, 3) / float(voxel_size)).astype(np.int64)
            good = (_grid_of(idx, voxel_size, interval, a_min) == q).all(axis=1)
        # worth it when almost every anchor is its lattice point's grid value and the lattice is coarser than the grid
        if good is not None:
            span = idx[good].max(axis=0) - idx[good].min(axis=0) + 1 if good.any() else np.array([1 << 20] * 3)
            if good.mean() >= 0.9 and int(span.max()) <= (1 << 15):
                lattice = (idx, good)
    if lattice is None:
        uniq, counts = _unique_rows(q) if n else (q, np.zeros(0, np.int64))
        bits = 16
        body = _encode_octree(uniq, bits) if n else b""
        dup = np.flatnonzero(counts > 1)
        extra = zlib.compress(np.stack([dup, counts[dup]], axis=1).astype("<i8").tobytes(), 9)
        head.append(struct.pack("<BQQBI", 0, n, uniq.shape[0], bits, len(extra)))
        return b"".join(head) + extra + body
    idx, good = lattice
    origin = idx[good].min(axis=0)
    rel = idx[good] - origin
    bits = max(1, int(rel.max()).bit_length())
    uniq, counts = _unique_rows(rel)
    # anchors that share a lattice point share its grid value: their multiplicity is all there is to keep
    dup = np.flatnonzero(counts > 1)
    exc = _lex(q[~good])                                       # anchors kept by their grid value
    extra = zlib.compress(np.stack([dup, counts[dup]], axis=1).astype("<i8").tobytes() + exc.astype("<u2").tobytes(), 9)
    head.append(struct.pack("<BQQBI", 1, n, uniq.shape[0], bits, len(extra)))
    head.append(struct.pack("<QQ3qd3f3f", int(dup.size), int(exc.shape[0]), *[int(v) for v in origin], float(voxel_size),
                            *[float(v) for v in interval], *[float(v) for v in a_min]))
    return b"".join(head) + extra + _encode_octree(uniq, bits)


MAX_ANCHORS = 1 << 31      # sanity cap on the header's 64-bit counts (a stream is untrusted input: StreamPack.load)


def _read_extra(raw: bytes, mode: int, n: int, n_uniq: int, n_dup: int, n_exc: int):
    """The zlib section of an anchor stream — (index, multiplicity) pairs of the repeated points and, in lattice mode, the
    exception list — inflated with an output limit and validated BEFORE anything indexes with it: on the device an out-of-range
    index is an assert that poisons the context, and a huge multiplicity an enormous allocation.  Returns (dup [k, 2] int64,
    exc [n_exc, 3] int64)."""
    if n > MAX_ANCHORS or n_uniq > n or n_dup > n_uniq or n_exc > n:
        raise ValueError("anchor_codec: corrupt header (counts)")
    limit = 16 * n_dup + 6 * n_exc if mode == 1 else 16 * n_uniq
    z = zlib.decompressobj()
    try:
        extra = z.decompress(raw, limit + 1)
    except zlib.error as e:
        raise ValueError(f"anchor_codec: corrupt stream (multiplicity section: {e})") from None
    if len(extra) > limit or z.unconsumed_tail or not z.eof:
        raise ValueError("anchor_codec: corrupt stream (multiplicity section larger than its header allows)")
    if mode == 1:
        if len(extra) != limit:
            raise ValueError("anchor_codec: corrupt stream (multiplicity section size)")
        dup = np.frombuffer(extra, dtype="<i8", count=2 * n_dup).reshape(-1, 2)
        exc = np.frombuffer(extra, dtype="<u2", offset=16 * n_dup, count=3 * n_exc).reshape(-1, 3).astype(np.int64)
    else:
        if len(extra) % 16:
            raise ValueError("anchor_codec: corrupt stream (multiplicity section size)")
        dup = np.frombuffer(extra, dtype="<i8").reshape(-1, 2)
        exc = np.zeros((0, 3), np.int64)
    if dup.shape[0]:
        i, c = dup[:, 0], dup[:, 1]
        if i[0] < 0 or i[-1] >= n_uniq or (np.diff(i) <= 0).any() or (c < 2).any() or (c > n).any():
            raise ValueError("anchor_codec: corrupt stream (multiplicity entries)")
    if int((dup[:, 1] - 1).sum()) + n_uniq + exc.shape[0] != n:
        raise ValueError("anchor_codec: corrupt stream (anchor count)")
    return dup, exc


def decode_anchors(data: bytes) -> np.ndarray:
    """uint16 [n, 3], sorted by (x, y, z)."""
    buf = memoryview(data)
    if bytes(buf[:5]) != MAGIC:
        raise ValueError("anchor_codec: not an anchor stream")
    mode, n, n_uniq, bits, n_extra = struct.unpack_from("<BQQBI", buf, 5)
    at = 5 + struct.calcsize("<BQQBI")
    if mode not in (0, 1) or bits < 1 or bits > 16 or n_uniq > n:
        raise ValueError("anchor_codec: corrupt header")
    n_dup = n_exc = 0
    if mode == 1:
        n_dup, n_exc, ox, oy, oz, voxel, i0, i1, i2, m0, m1, m2 = struct.unpack_from("<QQ3qd3f3f", buf, at)
        at += struct.calcsize("<QQ3qd3f3f")
    dup, exc = _read_extra(bytes(buf[at:at + n_extra]), mode, int(n), int(n_uniq), int(n_dup), int(n_exc))
    at += n_extra
    if n == 0:
        return np.zeros((0, 3), np.uint16)
    pts, at = _decode_octree(buf, at, int(n_uniq), int(bits))
    if mode == 0:
        out = pts
        if dup.size:                                           # multiplicities are indexed in np.unique's (lexicographic) order
            rep = np.ones(pts.shape[0], dtype=np.int64)
            pts = _lex(pts)
            rep[dup[:, 0]] = dup[:, 1]
            out = np.repeat(pts, rep, axis=0)
    else:
        if n_dup:                                              # multiplicities are indexed in np.unique's (lexicographic) order
            pts = _lex(pts)
            rep = np.ones(pts.shape[0], dtype=np.int64)
            rep[dup[:, 0]] = dup[:, 1]
            pts = np.repeat(pts, rep, axis=0)
        idx = pts + np.array([ox, oy, oz], dtype=np.int64)
        out = np.concatenate([_grid_of(idx, voxel, np.array([i0, i1, i2], np.float32), np.array([m0, m1, m2], np.float32)), exc])
    if out.shape[0] != n:
        raise ValueError("anchor_codec: corrupt stream (anchor count)")
    return _lex(out).astype(np.uint16)


# ------------------------------------------------------------------------------------------------ decode on the GPU
def _level_headers(buf: memoryview, at: int, bits: int):
    """Walks the per-level records without decoding them: [(n, table [n_present, 2], lanes, states view, words view)], end offset."""
    levels = []
    for _ in range(bits):
        n, n_present, lanes, n_words = struct.unpack_from("<IHHI", buf, at)
        at += 12
        if lanes < 1 or lanes > MAX_LANES or n_present < 1 or n_present > 256:
            raise ValueError("anchor_codec: corrupt level header")
        table = np.frombuffer(buf, dtype="<u2", count=2 * n_present, offset=at).reshape(n_present, 2)
        at += 4 * n_present
        states = np.frombuffer(buf, dtype="<u4", count=lanes, offset=at)
        at += 4 * lanes
        words = np.frombuffer(buf, dtype="<u2", count=n_words, offset=at)
        at += 2 * n_words
        levels.append((int(n), table, int(lanes), states, words))
    return levels, at


def decode_anchors_gpu(data: bytes, device="cuda"):
    """``decode_anchors`` with the entropy decode and the octree expansion on the GPU (csrc/anchor.hip): returns an int32 CUDA
    tensor [n, 3] sorted by (x, y, z) — the same values as ``decode_anchors``.  The host only walks the level headers (the
    symbol counts of all levels are in the stream, so nothing is read back before the end) and stages the streams in one copy."""
    import ctypes as C
    import torch
    from . import _lib
    buf = memoryview(data)
    if bytes(buf[:5]) != MAGIC:
        raise ValueError("anchor_codec: not an anchor stream")
    mode, n, n_uniq, bits, n_extra = struct.unpack_from("<BQQBI", buf, 5)
    at = 5 + struct.calcsize("<BQQBI")
    if mode not in (0, 1) or bits < 1 or bits > 16 or n_uniq > n:
        raise ValueError("anchor_codec: corrupt header")
    n_dup = n_exc = 0
    if mode == 1:
        n_dup, n_exc, ox, oy, oz, voxel, i0, i1, i2, m0, m1, m2 = struct.unpack_from("<QQ3qd3f3f", buf, at)
        at += struct.calcsize("<QQ3qd3f3f")
    dup, exc = _read_extra(bytes(buf[at:at + n_extra]), mode, int(n), int(n_uniq), int(n_dup), int(n_exc))      # before the device
    at += n_extra
    dev = torch.device(device)
    if n == 0:
        return torch.zeros(0, 3, dtype=torch.int32, device=dev)
    levels, _ = _level_headers(buf, at, int(bits))
    expect = 1
    for lv in levels:                         # a level has as many symbols as the level above has children: checked on the device
        if lv[0] < 1 or lv[0] > n_uniq:
            raise ValueError("anchor_codec: corrupt level header")
    if levels[0][0] != 1:
        raise ValueError("anchor_codec: corrupt level header")
    # one staged blob: per level [freq 256 x u16][states][words], every section 16-byte aligned
    sections, size = [], 0
    def put(arr):
        nonlocal size
        off = size
        sections.append((off, arr))
        size = (off + arr.nbytes + 15) // 16 * 16
        return off
    offs = []
    for n_l, table, lanes, states, words in levels:
        freq = np.zeros(256, dtype=np.uint16)
        freq[table[:, 0] & 0xFF] = table[:, 1]
        if int(freq.astype(np.int64).sum()) != PROB_SCALE:
            raise ValueError("anchor_codec: corrupt frequency table")
        offs.append((put(freq), put(np.ascontiguousarray(states)), put(np.ascontiguousarray(words))))
    blob = np.zeros(max(size, 16), dtype=np.uint8)
    for off, arr in sections:
        blob[off:off + arr.nbytes] = arr.view(np.uint8).reshape(-1)
    st = _lib.current_stream(dev)
    L = _lib.lib()
    g = torch.from_numpy(blob).to(dev, non_blocking=False)
    total_syms = sum(lv[0] for lv in levels)
    syms = torch.empty(total_syms, dtype=torch.uint8, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    descs = (_lib.AnchorLevelC * len(levels))()
    sym_at = []
    pos = 0
    for d, (n_l, table, lanes, states, words), (o_f, o_s, o_w) in zip(descs, levels, offs):
        d.freq, d.states, d.words = g.data_ptr() + o_f, g.data_ptr() + o_s, g.data_ptr() + o_w
        d.out, d.n, d.n_words, d.lanes = syms.data_ptr() + pos, n_l, int(words.size), lanes
        sym_at.append(pos)
        pos += n_l
    _lib.check(L.gsvc_anchor_rans_decode(descs, len(levels), _lib.ptr(err), st), "gsvc_anchor_rans_decode")
    # expansion, level after level: nodes of level l + 1 = children of level l's nodes
    nodes = None
    for li, (n_l, *_rest) in enumerate(levels):
        occ = syms[sym_at[li]:sym_at[li] + n_l]
        nxt = levels[li + 1][0] if li + 1 < len(levels) else int(n_uniq)
        cnt = torch.empty(n_l, dtype=torch.int64, device=dev)
        _lib.check(L.gsvc_octree_popcount(_lib.ptr(occ), n_l, _lib.ptr(cnt), st), "gsvc_octree_popcount")
        incl = torch.cumsum(cnt, dim=0)
        out = torch.full((nxt,), -1, dtype=torch.int64, device=dev)
        _lib.check(L.gsvc_octree_expand(_lib.ptr(nodes), _lib.ptr(occ), _lib.ptr(incl), n_l, nxt, _lib.ptr(out), _lib.ptr(err), st),
                   "gsvc_octree_expand")
        # the children must fill the next level exactly
        err.bitwise_or_((incl[-1:] != nxt).to(torch.int32) * 16)
        nodes = out
    pts = torch.empty(int(n_uniq), 3, dtype=torch.int64, device=dev)
    _lib.check(L.gsvc_morton_decode(_lib.ptr(nodes), int(n_uniq), _lib.ptr(pts), st), "gsvc_morton_decode")
    code = int(err.item())                     # the one read-back
    if code:
        raise ValueError(f"anchor_codec: corrupt stream (device decode error {code})")

    def lex(p):
        key = torch.sort((p[:, 0] << 40) | (p[:, 1] << 20) | p[:, 2]).values
        m = (1 << 20) - 1
        return torch.stack([key >> 40, (key >> 20) & m, key & m], dim=1)

    def with_multiplicity(p, dup):
        if dup.size == 0:
            return p
        p = lex(p)                             # multiplicities are indexed in np.unique's (lexicographic) order
        rep = torch.ones(p.shape[0], dtype=torch.int64, device=dev)
        rep[torch.from_numpy(dup[:, 0].copy()).to(dev)] = torch.from_numpy(dup[:, 1].copy()).to(dev)
        return torch.repeat_interleave(p, rep, dim=0)

    if mode == 0:
        out = with_multiplicity(pts, dup)
    else:
        idx = with_multiplicity(pts, dup) + torch.tensor([ox, oy, oz], dtype=torch.int64, device=dev)
        # Quantize_anchor's grid value of the lattice point, in the float32 arithmetic of _grid_of
        a = (idx.to(torch.float64) * float(voxel)).to(torch.float32)
        a_min = torch.tensor([m0, m1, m2], dtype=torch.float32, device=dev)
        interval = torch.tensor([i0, i1, i2], dtype=torch.float32, device=dev)
        q = torch.floor((a - a_min) / interval).clamp_(0, 65535).to(torch.int64)
        out = torch.cat([q, torch.from_numpy(exc).to(dev)]) if n_exc else q
    if out.shape[0] != n:
        raise ValueError("anchor_codec: corrupt stream (anchor count)")
    return lex(out).to(torch.int32)
