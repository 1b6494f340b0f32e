#!/usr/bin/env python3
"""Time the REFERENCE's own step body on PyTorch-CPU (build container only; /root/reference never travels): the slice of
/root/reference/pipeline/train.py that make_golden_step.py executes — four render() calls with the view swap, the flip-average, every
loss term, loss.backward() (:348-462) — read from the reference's file at run time, at BASELINE.json configs[2] shape (1080p, 64-frame
cube, 245 k anchors x K = 10, 16-frame slab, TRAINING_ENTROPY, lambda 0.004) and at configs[3]'s per-GPU shape (cfg_20240919.yaml
as is).  The native slots hold oracle/ (grid + rasterizer, OpenMP on every core).  This is the SAME work as bench.py's headline step
minus the optimizer: the CPU number the bench line carries as ``cpu_baseline_reference_python`` (host stated: it is this container,
not the GPU box).  Writes profiles/r05/reference_step_cpu_timing.json.

Run: python tests/golden/time_reference_step_cpu.py [--cfg3-too]
"""
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import _ref_import  # noqa: E402
from tests.golden.make_golden_step import reference_slice  # noqa: E402,F401


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    mode_ctx = _ref_import.install(rasterizer=True)
    with mode_ctx:
        import arguments as A
        import scene.gaussian_model as GM
        import ortho_gaussian_renderer as OGR
        from frame_cube.frame import Frame
        from utils.encodings import get_binary_vxl_size
        from utils.loss_utils import calc_optical_loss, l1_loss_func, ssim_func
        from utils.train_util import TrainingController
        from gsvc_amd.frame import SyntheticFrameCube
        step_code = reference_slice("render_mode = controller.render_mode", "loss.backward()")
        cases = [("configs[2] shape: 1080p, 64-frame cube, 245k anchors, 16-frame slab", 245_000, 64, 8.0)]
        if "--cfg3-too" in sys.argv:
            cases.append(("configs[3] per-GPU shape (cfg_20240919.yaml as is): 1080p, 600-frame cube, 100k anchors, threshold .05", 100_000, 600, None))
        results = {"host": {"cpus": os.cpu_count(), "torch_threads": torch.get_num_threads(), "torch": torch.__version__,
                            "where": "build container (not the GPU box: /root/reference does not travel)"},
                   "what": "the reference's step body (pipeline/train.py:348-462: 4 renders, flip-average, L1 + SSIM + regularisers + optical "
                           "+ lambda (rates + table bits) + mask term, loss.backward()) on PyTorch-CPU, oracle grid + rasterizer (OpenMP) in "
                           "the native slots; TRAINING_ENTROPY, lambda 0.004; no optimizer step",
                   "cases": []}
        H, W = 1080, 1920
        for label, anchors, T, slab in cases:
            cube = SyntheticFrameCube(H, W, T, seed=1234)
            mp = A.ModelParams()
            mp.threshold = (slab / cube.scale) if slab is not None else 0.05
            opt = A.OptimizationParams()
            opt.lmbda = 0.004
            torch.manual_seed(0)
            ref = GM.GaussianModel(mp, feat_dim=50, n_offsets=10, voxel_size=0.001, update_depth=3, update_init_factor=16,
                                   update_hierachy_factor=4, use_feat_bank=False, n_features_per_level=8, log2_hashmap_size=13,
                                   log2_hashmap_size_2D=15)
            ref.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
            g = torch.Generator().manual_seed(1)
            lim = torch.tensor([[-cube.x_min, -cube.y_min, -cube.z_min]]) * 1.1
            ref._anchor = nn.Parameter((torch.rand(anchors, 3, generator=g) * 2 - 1) * lim)
            ref._offset = nn.Parameter(torch.randn(anchors, 10, 3, generator=g) * 0.1)
            ref._mask = nn.Parameter(torch.ones(anchors, 10, 1))
            ref._anchor_feat = nn.Parameter(torch.randn(anchors, 50, generator=g) * 0.1)
            ref._scaling = nn.Parameter(torch.randn(anchors, 6, generator=g) * 0.3 - 6.0)
            rots = torch.zeros(anchors, 4)
            rots[:, 0] = 1
            ref._rotation = nn.Parameter(rots, requires_grad=False)
            ref._opacity = nn.Parameter(torch.zeros(anchors, 1), requires_grad=False)
            ref.spatial_lr_scale = 1.0
            ref.training_setup(opt)
            controller = TrainingController(opt)
            controller.current_iteration = 16001
            idx = T // 2

            def frame_of(i):
                f = cube[i]
                return Frame(image_id=i, plane="xy", image=f.image.contiguous(), x_min=f.x_min, y_min=f.y_min, z=f.z, image_width=W,
                             image_height=H, view_matrix=f.view_matrix.clone(), view_matrix_s=f.view_matrix_s.clone(), scale=f.scale,
                             cam_pos=f.cam_pos.clone())
            dataset = SimpleNamespace(x_min=cube.x_min, y_min=cube.y_min, scale=cube.scale, width=W, height=H)
            times, active = [], 0
            for it in range(3):
                for p in ref.parameters():
                    p.grad = None
                ns = dict(render=OGR.render, frame1=frame_of(idx), frame2=frame_of(idx + 1), gaussians=ref,
                          pipe=SimpleNamespace(debug=False, compute_cov3D_python=False, model_path=None), background=torch.tensor([0.0, 0.0, 0.0]),
                          controller=controller, opt=opt, iteration=16001, torch=torch, l1_loss_func=l1_loss_func, ssim_func=ssim_func,
                          calc_optical_loss=calc_optical_loss, get_binary_vxl_size=get_binary_vxl_size,
                          optical_flow=cube.get_optical_flow(idx), frame_cube=SimpleNamespace(dataset=dataset))
                t0 = time.perf_counter()
                exec(step_code, ns)
                dt = time.perf_counter() - t0
                rr = [ns[k] for k in ("render_results1_f", "render_results1_b", "render_results2_f", "render_results2_b")]
                active = sum(int(r.active_gaussains) for r in rr)
                print(f"{label}: step {it}: {dt:.2f} s, active Gaussians {active}, loss {float(ns['loss']):.4f}", flush=True)
                if it >= 1:
                    times.append(dt)
            results["cases"].append({"case": label, "anchors": anchors, "frames": T, "threshold": float(mp.threshold),
                                     "active_gaussians_per_step": active, "visible_anchors_per_render": [int(r.visible_mask.sum()) for r in rr],
                                     "seconds_per_step": {"median": float(np.median(times)), "min": float(np.min(times)), "timed_steps": len(times)},
                                     "gaussians_per_s": active / float(np.median(times))})
    out = os.path.join(ROOT, "profiles", "r05", "reference_step_cpu_timing.json")
    json.dump(results, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
