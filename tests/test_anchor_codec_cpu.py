"""Host tests of gsvc_amd/anchor_codec.py (the stand-in for the reference's G-PCC anchor geometry coding, reference
utils/encodings.py:714-826): exact round trips in both modes, sizes against the counting bound, corrupt streams refused."""
import math

import numpy as np
import pytest

from gsvc_amd import anchor_codec as ac


def _bound_bits(cells: float, n: int) -> float:
    """log2 C(cells, n): what any lossless code of n distinct points among `cells` equally likely places needs on average."""
    return (math.lgamma(cells + 1) - math.lgamma(n + 1) - math.lgamma(cells - n + 1)) / math.log(2)


@pytest.mark.parametrize("n", [0, 1, 2, 777, 50_000])
def test_grid_mode_round_trip_with_duplicates(n):
    rng = np.random.default_rng(n)
    q = rng.integers(0, 65536, (n, 3)).astype(np.uint16)
    if n > 10:
        q[5] = q[3]; q[6] = q[3]; q[-1] = q[0]                      # duplicates keep their multiplicity
    out = ac.decode_anchors(ac.encode_anchors(q))
    want = q[np.lexsort((q[:, 2], q[:, 1], q[:, 0]))] if n else q.reshape(0, 3)
    assert out.dtype == np.uint16 and np.array_equal(out, want)


def test_grid_mode_size_is_near_the_counting_bound():
    """Uniform points in a thin z slab of the 16-bit grid (what a GSVC model looks like to a coder that does not know the voxel
    lattice): within 12 % of log2 C(cells, n) — a plain occupancy octree with one static model per level pays ~1.5 bits per point
    over the counting bound at the levels where nodes hold one or two points."""
    rng = np.random.default_rng(1)
    n = 120_000
    q = np.unique(np.stack([rng.integers(0, 65536, n), rng.integers(0, 65536, n), rng.integers(30000, 34096, n)], 1), axis=0)
    data = ac.encode_anchors(q.astype(np.uint16))
    assert np.array_equal(ac.decode_anchors(data), q[np.lexsort((q[:, 2], q[:, 1], q[:, 0]))].astype(np.uint16))
    bound = _bound_bits(65536.0 * 65536.0 * 4096.0, q.shape[0])
    assert 8 * len(data) <= 1.12 * bound, (8 * len(data) / q.shape[0], bound / q.shape[0])
    assert 8 * len(data) < 0.8 * 48 * q.shape[0]                      # raw storage is 48 bits per anchor


def test_lattice_mode_codes_voxel_centres_in_a_third_of_the_raw_size():
    """Anchors on the 0.001 voxel lattice of a UVG-shaped cube (x in +-1.1, y in +-0.62, 64 frames of z), quantised by
    Quantize_anchor's rule: the lattice mode finds them, round-trips the GRID values exactly (exceptions included) and needs
    about log2(lattice cells / anchors) + 2.6 bits per anchor (the counting bound is + 1.44)."""
    rng = np.random.default_rng(7)
    voxel = 0.001
    lo, hi = np.array([-1.1, -0.62, -0.0367]), np.array([1.1, 0.62, 0.0367])
    n = 200_000
    idx = np.unique(np.round(rng.uniform(lo, hi, (n, 3)) / voxel), axis=0)
    pos = (idx * voxel).astype(np.float32)
    a_min, a_max = pos.min(axis=0), pos.max(axis=0)
    interval = ((a_max - a_min) / 65536.0 + 1e-6).astype(np.float32)
    q = np.clip(np.floor((pos - a_min) / interval), 0, 65535).astype(np.uint16)
    # a few anchors that are NOT lattice points (they must come back exactly too)
    pos[:50] += np.float32(0.00037)
    q[:50] = np.clip(np.floor((pos[:50] - a_min) / interval), 0, 65535).astype(np.uint16)
    data = ac.encode_anchors(q, positions=pos, voxel_size=voxel, interval=interval, a_min=a_min)
    assert data[5] == 1                                               # lattice mode
    want = q[np.lexsort((q[:, 2], q[:, 1], q[:, 0]))]
    assert np.array_equal(ac.decode_anchors(data), want)
    cells = np.prod((hi - lo) / voxel)
    per_anchor = 8 * len(data) / q.shape[0]
    assert per_anchor <= math.log2(cells / q.shape[0]) + 3.0, per_anchor
    assert per_anchor < 20.0                                          # raw: 48; the 16-bit octree: ~31
    # without the positions the same anchors take the grid mode and cost about twice as much
    plain = ac.encode_anchors(q)
    assert plain[5] == 0 and np.array_equal(ac.decode_anchors(plain), want) and len(plain) > 1.6 * len(data)


def test_corrupt_streams_are_refused():
    rng = np.random.default_rng(3)
    q = rng.integers(0, 65536, (3000, 3)).astype(np.uint16)
    data = bytearray(ac.encode_anchors(q))
    with pytest.raises(ValueError):
        ac.decode_anchors(b"nope" + bytes(data))
    with pytest.raises(ValueError):
        ac.decode_anchors(bytes(data[:len(data) // 2]))
    bad = bytearray(data)
    bad[len(bad) // 2] ^= 0x55
    try:                                                              # a flipped bit either fails a check or changes the points
        out = ac.decode_anchors(bytes(bad))
        assert not np.array_equal(out, q[np.lexsort((q[:, 2], q[:, 1], q[:, 0]))])
    except ValueError:
        pass


def _with_extra(data: bytes, pairs: np.ndarray, n=None) -> bytes:
    """The grid-mode stream ``data`` with its multiplicity section replaced by ``pairs`` (and optionally the anchor count)."""
    import struct
    import zlib
    head = struct.calcsize("<BQQBI")
    mode, n0, n_uniq, bits, n_extra = struct.unpack_from("<BQQBI", data, 5)
    assert mode == 0
    extra = zlib.compress(np.asarray(pairs, dtype="<i8").tobytes(), 9)
    return data[:5] + struct.pack("<BQQBI", mode, n0 if n is None else n, n_uniq, bits, len(extra)) + extra + data[5 + head + n_extra:]


@pytest.mark.parametrize("decoder", ["host", "gpu"])
def test_hostile_multiplicity_sections_are_refused_before_they_index_anything(decoder):
    """the (index, count) pairs of an untrusted anchor.b went straight into rep[idx] = cnt / repeat_interleave.
    The GPU decoder must refuse them on the host (an out-of-range index on the device is an assert that poisons the context):
    this test runs it WITHOUT a GPU — every case raises before the first device call."""
    rng = np.random.default_rng(5)
    q = rng.integers(0, 4096, (2000, 3)).astype(np.uint16)
    q[100:110] = q[0]                                                 # one point eleven times
    data = ac.encode_anchors(q)
    dec = ac.decode_anchors if decoder == "host" else (lambda b: ac.decode_anchors_gpu(b, device="cpu"))
    n_uniq = np.unique(q, axis=0).shape[0]
    for pairs, n in (([[n_uniq, 11]], None),                          # index past the unique points
                     ([[-1, 11]], None),                              # negative index
                     ([[0, 1 << 40]], None),                          # a multiplicity that would allocate terabytes
                     ([[0, 1]], None),                                # multiplicity below 2
                     ([[3, 6], [3, 6]], None),                        # the same point listed twice
                     ([[0, 12]], None),                               # counts that do not add up to n
                     ([[0, 11]], 1 << 40)):                           # an absurd anchor count in the header
        with pytest.raises(ValueError, match="anchor_codec"):
            dec(_with_extra(data, np.array(pairs, dtype=np.int64), n))
    import struct
    import zlib
    bomb = zlib.compress(bytes(64 << 20), 9)                          # 64 MiB of zeros in 64 KiB: inflated only up to the header's limit
    head = struct.calcsize("<BQQBI")
    mode, n0, nu, bits, n_extra = struct.unpack_from("<BQQBI", data, 5)
    with pytest.raises(ValueError, match="larger than its header allows"):
        dec(data[:5] + struct.pack("<BQQBI", mode, n0, nu, bits, len(bomb)) + bomb + data[5 + head + n_extra:])
    if decoder == "host":
        want = q[np.lexsort((q[:, 2], q[:, 1], q[:, 0]))]
        assert np.array_equal(ac.decode_anchors(data), want)


def test_torch_lattice_path_writes_the_same_bytes_as_the_numpy_path():
    """encode_anchors' device path (round 6: lattice index, grid-value test, distinct lattice points, Morton keys and octree levels as
    torch operations — on the GPU in production) forced onto the CPU: byte for byte the numpy path's stream, with duplicates, with
    anchors off the lattice (exceptions) and when the lattice mode does not apply."""
    import numpy as np
    from gsvc_amd import anchor_codec as ac
    rng = np.random.default_rng(5)
    voxel = 0.001
    a_min = np.array([-1.05, -0.6, -0.12], np.float32)
    interval = ((np.array([1.05, 0.6, 0.12], np.float32) - a_min) / 65535.0).astype(np.float32)

    def streams(pos, snap=True):
        a = ((np.round(pos / voxel) * voxel) if snap else pos).astype(np.float32)
        q = np.clip(np.floor((a - a_min) / interval), 0, 65535).astype(np.uint16)
        host = ac.encode_anchors(q, positions=pos, voxel_size=voxel, interval=interval, a_min=a_min)
        ac._torch_device = "cpu"
        try:
            dev = ac.encode_anchors(q, positions=pos, voxel_size=voxel, interval=interval, a_min=a_min)
        finally:
            ac._torch_device = None
        return host, dev, q
    lattice = np.round(rng.uniform([-1, -0.55, -0.1], [1, 0.55, 0.1], (30000, 3)) / voxel) * voxel
    pos = np.concatenate([lattice, lattice[:500], lattice[:40]])                       # duplicates (multiplicity 2 and 3)
    host, dev, q = streams(pos)
    assert host == dev and host[5] == 1                                                 # lattice mode, the same bytes
    assert np.array_equal(ac.decode_anchors(dev), ac._lex(q.astype(np.int64)).astype(np.uint16))
    off = pos.copy()
    off[:900] += rng.uniform(-0.4, 0.4, (900, 3)) * voxel                               # 3 % off the lattice: exceptions
    host, dev, q = streams(off)
    assert host == dev
    host, dev, _ = streams(rng.uniform([-1, -0.55, -0.1], [1, 0.55, 0.1], (20000, 3)), snap=False)  # grid values of off-lattice positions: grid mode either way
    assert host == dev and host[5] == 0
