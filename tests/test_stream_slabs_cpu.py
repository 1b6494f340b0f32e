"""gsvc_amd.stream_codec.reorder_and_split against the reference's (tests/golden/stream_slabs.npz, written by make_golden_slabs.py
from /root/reference/utils/encodings.py:827-862): the same (z, x, y) permutation and the same slab ranges, including anchors that
sit exactly on a slab boundary (the boundaries are float32 accumulations in the reference, and here)."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def test_slab_split_matches_the_reference():
    from tests.golden.make_golden_slabs import anchors
    from gsvc_amd.stream_codec import reorder_and_split
    g = np.load(os.path.join(HERE, "golden", "stream_slabs.npz"))
    for i, (seed, zr, lattice) in enumerate(g["cases"]):
        a = anchors(int(seed), float(zr), bool(lattice))
        sel, splits = reorder_and_split(a)
        assert np.array_equal(sel.numpy().astype(np.int32), g[f"c{i}::selection"]), i
        assert [list(s) for s in splits] == g[f"c{i}::splits"].tolist(), i


def test_slab_split_covers_anchors_the_reference_walk_leaves_out():
    """z = +0.05 exactly with |z| <= 0.05: the reference's accumulated upper bound stops at 0.05 (exclusive) and those anchors are in
    no slab; here they join the last one."""
    from gsvc_amd.stream_codec import reorder_and_split
    g = torch.Generator().manual_seed(3)
    a = torch.cat([torch.rand(3000, 2, generator=g) * 2 - 1, (torch.rand(3000, 1, generator=g) * 2 - 1) * 0.05], 1)
    a = (a * 1000).round() / 1000
    assert float(a[:, 2].max()) == float(torch.tensor(0.05))
    sel, splits = reorder_and_split(a)
    assert splits[0][0] == 0 and splits[-1][1] == 3000 and all(x[1] == y[0] for x, y in zip(splits, splits[1:]))
    assert torch.equal(torch.sort(sel).values, torch.arange(3000))
