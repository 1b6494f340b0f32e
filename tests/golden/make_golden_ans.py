#!/usr/bin/env python3
"""A committed byte stream of the attribute coder, so that its format cannot drift silently: symbols, model and the stream
oracle/ans_oracle.py (the independent integer statement of gsvc_amd/csrc/ans.hip's specification) produces for them.  The
reference's coder (external gsvc_cuda_ans, reference README.md:51) is not in its tree: there is no reference stream to capture,
this fixture pins OUR format.  tests/test_ans_oracle_cpu.py checks the oracle against it, tests/test_codec_gpu.py the HIP encoder
(same bytes) and decoder (same symbols).  Usage:  python tests/golden/make_golden_ans.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ans_oracle  # noqa: E402


def case(seed=7, n=5000, smin=-15000, smax=15000):
    """GSVC-shaped: a feature-like part (sigma ~ 1 symbol), a scaling-like part (sigma of hundreds of symbols: ~9 bits each), a
    part at the context model's 1e-9 scale clamp (sigma = 1e-9 / Q: the symbol is certain, or a model miss costs 20 bits), and a
    few symbols at the ends of the range."""
    rng = np.random.default_rng(seed)
    mu = np.concatenate([rng.normal(0, 3, n // 2), rng.normal(800, 2000, n // 4), rng.normal(0, 40, n - n // 2 - n // 4)]).astype(np.float32)
    sigma = np.concatenate([rng.uniform(0.3, 2.0, n // 2), rng.uniform(100, 600, n // 4),
                            np.full(n - n // 2 - n // 4, 1e-9 / 0.001)]).astype(np.float32)
    sym = np.rint(mu.astype(np.float64) + sigma.astype(np.float64) * rng.normal(0, 1, n)).astype(np.int64)
    sym[-40:-20] += rng.integers(-3, 4, 20)                  # misses under the clamped model
    sym = np.clip(sym, smin, smax)
    sym[:4] = [smin, smax, smin + 1, smax - 1]
    perm = rng.permutation(n)
    return sym[perm].astype(np.int32), mu[perm], sigma[perm], smin, smax


if __name__ == "__main__":
    sym, mu, sigma, smin, smax = case()
    stream = ans_oracle.encode(sym, mu, sigma, smin, smax, seg_len=1024)
    assert np.array_equal(ans_oracle.decode(stream, mu, sigma), sym)
    path = os.path.join(HERE, "ans_stream.npz")
    np.savez_compressed(path, sym=sym, mu=mu, sigma=sigma, smin=np.int32(smin), smax=np.int32(smax), seg_len=np.int32(1024),
                        stream=np.frombuffer(stream, dtype=np.uint8))
    print(f"ans_stream.npz: {len(sym)} symbols -> {len(stream)} bytes ({8 * len(stream) / len(sym):.2f} bit/symbol), file {os.path.getsize(path)} bytes")
