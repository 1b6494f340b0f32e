"""oracle/ans_oracle.py — the independent integer statement of the attribute coder's specification (gsvc_amd/csrc/ans.hip) — on
the CPU: the committed stream (tests/golden/ans_stream.npz), round trips, the Phi table's checksum against the library's, refusals.
The HIP kernels are held to the same bytes in tests/test_codec_gpu.py."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_committed_stream_is_what_the_specification_produces():
    from oracle import ans_oracle
    g = np.load(os.path.join(HERE, "golden", "ans_stream.npz"))
    stream = ans_oracle.encode(g["sym"], g["mu"], g["sigma"], int(g["smin"]), int(g["smax"]), seg_len=int(g["seg_len"]))
    assert stream == g["stream"].tobytes()
    assert np.array_equal(ans_oracle.decode(stream, g["mu"], g["sigma"]), g["sym"])
    # the generator's inputs are reproducible too
    from tests.golden.make_golden_ans import case
    sym, mu, sigma, smin, smax = case()
    assert np.array_equal(sym, g["sym"]) and np.array_equal(mu, g["mu"]) and np.array_equal(sigma, g["sigma"])


def test_phi_table_checksum_equals_the_librarys():
    import ctypes
    from gsvc_amd import _lib
    from oracle import ans_oracle
    L = _lib.lib()
    L.gsvc_ans_table_checksum.restype = ctypes.c_uint32
    assert int(L.gsvc_ans_table_checksum()) == ans_oracle.CHECKSUM      # no GPU needed: the table is built on the host
    tab = ans_oracle.phi_table()
    assert tab[0] == 0 and tab[-1] == 0xFFFFFFFF and tab[2048] == 1 << 31 and all(a <= b for a, b in zip(tab, tab[1:]))


@pytest.mark.parametrize("seed,n,seg_len,smin,smax", [(0, 1, 8, -3, 3), (1, 700, 64, -20, 20), (2, 513, 512, -15000, 15000), (3, 300, 4096, 0, 1)])
def test_round_trips_and_code_length(seed, n, seg_len, smin, smax):
    from oracle import ans_oracle
    rng = np.random.default_rng(seed)
    mu = rng.normal(0, min(4, smax), n).astype(np.float32)
    sigma = rng.uniform(0.2, 3.0, n).astype(np.float32)
    sym = np.clip(np.rint(mu + sigma * rng.normal(0, 1, n)), smin, smax).astype(np.int32)
    stream = ans_oracle.encode(sym, mu, sigma, smin, smax, seg_len=seg_len)
    assert np.array_equal(ans_oracle.decode(stream, mu, sigma), sym)
    # the code length is the model's: sum of -log2(freq / 2^20) within the per-segment constant (4 state bytes + a flush byte or two)
    m64, inv = ans_oracle._model(mu, sigma)
    bits = sum(-np.log2((ans_oracle.cdf(int(s) + 1, m64[i], inv[i], smin, smax) - ans_oracle.cdf(int(s), m64[i], inv[i], smin, smax))
                        / ans_oracle.M) for i, s in enumerate(sym))
    n_seg = (n + seg_len - 1) // seg_len
    payload = len(stream) - ans_oracle.HEADER.size - 4 * n_seg
    assert bits / 8 - 1 <= payload <= bits / 8 + 6 * n_seg + 1, (bits / 8, payload)


def test_refusals():
    from oracle import ans_oracle
    mu, sigma = np.zeros(10, np.float32), np.ones(10, np.float32)
    sym = np.zeros(10, np.int32)
    stream = ans_oracle.encode(sym, mu, sigma, -2, 2, seg_len=4)
    with pytest.raises(ValueError):
        ans_oracle.encode(np.full(10, 3, np.int32), mu, sigma, -2, 2)                       # symbol outside the range
    with pytest.raises(ValueError):
        ans_oracle.decode(b"GSA2" + stream[4:], mu, sigma)
    with pytest.raises(ValueError):
        ans_oracle.decode(stream[:-1], mu, sigma)
    bad = bytearray(stream)
    bad[4] ^= 1                                                                             # another Phi table
    with pytest.raises(ValueError):
        ans_oracle.decode(bytes(bad), mu, sigma)
