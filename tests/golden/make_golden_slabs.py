#!/usr/bin/env python3
"""The stream codec's z-slab split: the reference's own ``reorder_and_split`` (utils/encodings.py:827-862: (z, x, y)
order, slabs of 0.01 whose boundaries are float32 accumulations) on seeded anchor sets — continuous coordinates and coordinates on
the 0.001 voxel lattice, where anchors sit exactly on slab boundaries and the accumulated boundaries decide their slab.  Inputs
are regenerated from seeds (tests/test_stream_slabs_cpu.py); the fixture holds the reference's permutation and slab ranges.
Build container only.  Usage: python tests/golden/make_golden_slabs.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import _ref_import  # noqa: E402
from tests.golden.make_golden_common import save  # noqa: E402

CASES = [(0, 0.033, False), (1, 0.033, True), (2, 0.3125, False), (5, 0.3125, True), (10, 0.0955, True), (12, 0.033, True)]      # (seed, z range, on the lattice)


def anchors(seed, zr, lattice, n=3000):
    g = torch.Generator().manual_seed(seed)
    a = torch.cat([torch.rand(n, 2, generator=g) * 2 - 1, (torch.rand(n, 1, generator=g) * 2 - 1) * zr], 1)
    return (a * 1000).round() / 1000 if lattice else a


def main():
    with _ref_import.install():
        import utils.encodings as E
        out = {"cases": np.array([[s, z, float(l)] for s, z, l in CASES])}
        for i, (seed, zr, lattice) in enumerate(CASES):
            a = anchors(seed, zr, lattice)
            sel, splits = E.reorder_and_split(a)
            sp = np.array([[int(x), int(y)] for x, y in splits], dtype=np.int64)
            assert sp[0, 0] == 0 and sp[-1, 1] == a.shape[0] and (sp[1:, 0] == sp[:-1, 1]).all(), "the reference dropped anchors: not a fixture"
            out[f"c{i}::selection"] = sel.numpy().astype(np.int32)
            out[f"c{i}::splits"] = sp
            print(i, seed, zr, lattice, len(sp), "slabs")
        save("stream_slabs", **out)


if __name__ == "__main__":
    main()
