#!/usr/bin/env python3
"""Model creation: the reference's own ``GaussianModel.create_from_pcd`` (scene/gaussian_model.py:748-800: voxel
down-sampling, per-anchor initial scaling log(sqrt(mean squared distance to the 3 nearest anchors)), identity rotations, opacity
inverse_sigmoid(0.1), zero offsets / features, unit masks) on PyTorch-CPU.  ``simple_knn._C.distCUDA2`` — an external CUDA extension
whose source ships as a zip beside the reference — is filled with a brute-force float64 statement of what it computes (the mean of the
three smallest squared distances to OTHER points; simple-knn.zip!simple_knn.cu:63-218), so the fixture also pins csrc/knn.hip through
the reference's call path.  Two cases: a fixed voxel size (the configuration's 0.001) and ``voxel_size <= 0`` (the median of the 3-NN
distances becomes the voxel size).  The fixture holds the input points and the created tensors.

Runs in the build container only (needs /root/reference).  Usage:  python tests/golden/make_golden_init.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import _ref_import  # noqa: E402
from tests.golden.make_golden_common import save  # noqa: E402


def mean_3nn_dist2(pts: torch.Tensor) -> torch.Tensor:
    p = pts.double()
    d = torch.cdist(p, p) ** 2
    d.fill_diagonal_(float("inf"))
    return d.topk(3, dim=1, largest=False).values.mean(dim=1).float()


def main():
    mode_ctx = _ref_import.install()
    sys.modules["simple_knn._C"].distCUDA2 = mean_3nn_dist2
    with mode_ctx:
        import arguments as A
        import scene.gaussian_model as GM
        GM.distCUDA2 = mean_3nn_dist2
        out = {}
        rng = np.random.default_rng(77)
        pts = rng.uniform([-1.0, -0.5625, -0.03], [1.0, 0.5625, 0.03], (3000, 3))
        pts[:200] = pts[200:400] + rng.normal(0, 2e-4, (200, 3))          # near-duplicates: some share a voxel and are merged
        out["points"] = pts.astype(np.float64)
        for tag, voxel in (("fixed", 0.001), ("auto", 0.0)):
            ref = GM.GaussianModel(A.ModelParams(), feat_dim=50, n_offsets=10, voxel_size=voxel, update_depth=3, update_init_factor=16,
                                   update_hierachy_factor=4, use_feat_bank=False, n_features_per_level=2, log2_hashmap_size=9,
                                   log2_hashmap_size_2D=9, resolutions_list=(18, 24), resolutions_list_2D=(130, 258))
            ref.update_anchor_bound(-1.0, -0.5625, -0.03125)
            np.random.seed(5)                                         # voxelize_sample shuffles in place before np.unique (order-free result)
            ref.create_from_pcd(SimpleNamespace(points=pts.copy()), spatial_lr_scale=2.0)
            pre = tag + "::"
            out[pre + "voxel_size"] = np.float64(ref.voxel_size)
            for nm in ("_anchor", "_offset", "_mask", "_anchor_feat", "_scaling", "_rotation", "_opacity"):
                out[pre + nm] = getattr(ref, nm)
            out[pre + "spatial_lr_scale"] = np.float64(ref.spatial_lr_scale)
            # (nn.Parameter(t.requires_grad_(False)) is a Parameter that DOES require grad: what the reference's lines produce)
            out[pre + "requires_grad"] = np.array([bool(getattr(ref, nm).requires_grad) for nm in
                                                   ("_anchor", "_offset", "_mask", "_anchor_feat", "_scaling", "_rotation", "_opacity")])
            print(tag, "voxel", ref.voxel_size, "anchors", ref._anchor.shape[0])
        save("model_init", **out)


if __name__ == "__main__":
    main()
