"""Golden vector for the bit accounting (reference scene/gaussian_model.py:1599-1725 `estimate_final_bits`), generated
by running the REFERENCE on PyTorch-CPU in the build container (stubs: _ref_import.py; the hash grid behind it is
oracle/grid_oracle.c in the `_gridencoder` slot).  The model is the one of tiny_model.npz (state dict committed there)
with the per-anchor parameters replaced by seeded draws that exercise every term (features far from their predicted
mean, masked offsets, pruned anchors).  Writes tests/golden/final_bits.npz.

Run: python tests/golden/make_golden_bits.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402
from make_golden_common import npy  # noqa: E402


def main():
    mode = _ref_import.install()
    g = np.load(os.path.join(HERE, "tiny_model.npz"))
    with mode:
        import arguments as A
        import scene.gaussian_model as GM
        mp = A.ModelParams()
        ref = GM.GaussianModel(mp, feat_dim=8, n_offsets=4, voxel_size=0.001, update_depth=3, update_init_factor=16,
                               update_hierachy_factor=4, use_feat_bank=False, n_features_per_level=2, log2_hashmap_size=9,
                               log2_hashmap_size_2D=11, resolutions_list=(18, 24, 33), resolutions_list_2D=(130, 258))
        sd = {k[4:]: torch.from_numpy(np.array(g[k])) for k in g.files if k.startswith("sd::")}
        An = sd["_anchor"].shape[0]
        rng = torch.Generator().manual_seed(911)
        K, F = 4, 8
        per_anchor = {
            "_anchor": sd["_anchor"].clone(),
            "_offset": torch.randn(An, K, 3, generator=rng) * 1.5,
            "_mask": torch.randn(An, K, 1, generator=rng) * 4.0,            # sigmoid > 0.01 for most, not all; some anchors fully masked
            "_anchor_feat": torch.randn(An, F, generator=rng) * 3.0,
            "_scaling": torch.randn(An, 6, generator=rng) * 0.5 - 3.0,
            "_rotation": sd["_rotation"].clone(),
            "_opacity": sd["_opacity"].clone(),
        }
        per_anchor["_mask"][::9] = -9.0                                       # every 9th anchor: all offsets masked -> pruned by get_mask_anchor
        for nm, v in per_anchor.items():
            setattr(ref, nm, nn.Parameter(v.clone(), requires_grad=nm not in ("_rotation", "_opacity")))
            sd[nm] = v.clone()
        ref.load_state_dict(sd, strict=True)
        ref.update_anchor_bound(float(g["x_lim"]), float(g["y_lim"]), float(g["z_lim"]))
        with torch.no_grad():
            log_info, bi = ref.estimate_final_bits()
        out = {"in::" + nm: npy(v) for nm, v in per_anchor.items()}
        for f in ("bit_anchor", "bit_anchor_gpcc", "bit_feat", "bit_scaling", "bit_offsets", "bit_hash", "bit_masks", "bit_mlp",
                  "bit_mlp_encoded"):
            out["bits::" + f] = np.float64(getattr(bi, f))
        out["log_info"] = np.array(log_info)
        print(log_info)
        print({k: float(v) for k, v in out.items() if k.startswith("bits::")})
    np.savez_compressed(os.path.join(HERE, "final_bits.npz"), **out)


if __name__ == "__main__":
    main()
