"""Time the REFERENCE's own Python hot path on PyTorch-CPU (this container only; /root/reference never travels).

BASELINE.md section 4: the reference's `generate_neural_gaussians` (anchor -> neural Gaussians: getters, hash grid
through our CPU grid oracle in the `_gridencoder` slot, entropy context, quantisation noise, rate, the generator /
deformation MLPs) forward + backward, followed by our CPU raster oracle forward + backward on the Gaussians it
produced, at BASELINE.json configs[0] size (~5k Gaussians) and at a 50k-anchor size.  3 warm-ups + N timed
iterations, median and min.  Writes profiles/r01/reference_cpu_timing.json; `--round3` times the round-3 shapes instead
(BASELINE.json configs[2]: 245 k anchors in a 64-frame 1080p cube, 16-frame slab; configs[3]: cfg_20240919.yaml as is, 100 k
anchors in a 600-frame cube, threshold .05) and writes profiles/r03/reference_cpu_timing.json.

Run: python tests/golden/time_reference_cpu.py
"""
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_import  # noqa: E402


def build(ref_mods, anchors, seed, zl=-0.3125):
    A, GM = ref_mods
    mp = A.ModelParams()
    mp.threshold = 0.05
    torch.manual_seed(seed)
    ref = GM.GaussianModel(mp, feat_dim=50, n_offsets=10, voxel_size=0.001, update_depth=3, update_init_factor=16,
                           update_hierachy_factor=4, use_feat_bank=False, n_features_per_level=2, log2_hashmap_size=13,
                           log2_hashmap_size_2D=15)
    g = torch.Generator().manual_seed(seed + 1)
    xl, yl = -1.0, -0.5625
    ref.update_anchor_bound(xl, yl, zl)
    lim = torch.tensor([[-xl, -yl, -zl]])
    import torch.nn as nn
    ref._anchor = nn.Parameter((torch.rand(anchors, 3, generator=g) * 2 - 1) * lim)
    ref._offset = nn.Parameter(torch.randn(anchors, 10, 3, generator=g) * 0.1)
    ref._mask = nn.Parameter(torch.ones(anchors, 10, 1))
    ref._anchor_feat = nn.Parameter(torch.randn(anchors, 50, generator=g) * 0.1)
    ref._scaling = nn.Parameter(torch.randn(anchors, 6, generator=g) * 0.3 - 5.0)
    rots = torch.zeros(anchors, 4)
    rots[:, 0] = 1
    ref._rotation = nn.Parameter(rots, requires_grad=False)
    ref._opacity = nn.Parameter(torch.zeros(anchors, 1), requires_grad=False)
    return ref


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    mode_ctx = _ref_import.install()
    import oracle
    oracle.build()
    with mode_ctx:
        import arguments as A
        import scene.gaussian_model as GM
        import ortho_gaussian_renderer.guassian as G
        results = {"host": {"cpus": os.cpu_count(), "torch_threads": torch.get_num_threads(), "torch": torch.__version__},
                   "what": "reference generate_neural_gaussians (TRAINING_ENTROPY) fwd+bwd on PyTorch-CPU with the CPU grid "
                           "oracle in the _gridencoder slot, then the CPU raster oracle fwd+bwd on its output",
                   "cases": []}
        round3 = "--round3" in sys.argv
        cases = ((("configs[0]-size: 256x256, ~5k Gaussians", 7000, (256, 256), 10, None, -0.3125),
                  ("50k anchors, 1080p", 50000, (1080, 1920), 3, None, -0.3125)) if not round3 else
                 (("configs[2] shape: 1080p, 245k anchors in a 64-frame cube, 16-frame slab", 245000, (1080, 1920), 2, 8.0 / 960.0, -64 / 2 / 960.0),
                  ("configs[3] shape (cfg_20240919.yaml as is): 1080p, 100k anchors in a 600-frame cube, threshold .05", 100000, (1080, 1920), 2,
                   0.05, -0.3125)))
        for label, anchors, (H, W), iters, thr_fixed, zl in cases:
            ref = build((A, GM), anchors, 7, zl)
            z_cam = 0.0
            frame = SimpleNamespace(cam_pos=torch.tensor([0.0, 0.0, z_cam]))
            thr = thr_fixed if thr_fixed is not None else (0.05 if anchors <= 7000 else 8.0 / 960.0 * 4)
            visible = (ref.get_anchor[:, 2] - z_cam).abs() < thr
            scale = max(H, W, 64) / 2
            view = np.eye(4, dtype=np.float32)
            view[2, 3] = -z_cam
            st = oracle.make_settings(H, W, -W / 2 / scale, -H / 2 / scale, scale, thr, view)
            t_gen, t_ras, P = [], [], 0
            for it in range(3 + iters):
                ref.zero_grad()
                t0 = time.perf_counter()
                gss = G.generate_neural_gaussians(frame, ref, visible, G.GenerateMode.TRAINING_ENTROPY)
                loss = gss.xyz.sum() + gss.color.sum() + gss.opacity.sum() + gss.scaling.sum() + gss.rot.sum() + gss.bit_per_param
                loss.backward()
                t1 = time.perf_counter()
                arrs = [t.detach().numpy().astype(np.float32) for t in (gss.xyz, gss.color, gss.opacity, gss.scaling, gss.rot)]
                fwd = oracle.raster_forward(st, *arrs, num_threads=os.cpu_count())
                oracle.raster_backward(st, *arrs, fwd, np.ones((3, H, W), np.float32))
                t2 = time.perf_counter()
                if it >= 3:
                    t_gen.append(t1 - t0)
                    t_ras.append(t2 - t1)
                P = int(gss.xyz.shape[0])
            results["cases"].append({
                "case": label, "anchors": anchors, "visible_anchors": int(visible.sum()), "gaussians": P, "iterations": iters,
                "reference_generate_fwd_bwd_s": {"median": float(np.median(t_gen)), "min": float(np.min(t_gen))},
                "oracle_raster_fwd_bwd_s": {"median": float(np.median(t_ras)), "min": float(np.min(t_ras))},
                "gaussians_per_s_generate_plus_raster": P / float(np.median(t_gen) + np.median(t_ras))})
            print(json.dumps(results["cases"][-1]))
    out = os.path.join(ROOT, "profiles", "r03" if "--round3" in sys.argv else "r01", "reference_cpu_timing.json")
    json.dump(results, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
