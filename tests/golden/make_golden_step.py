#!/usr/bin/env python3
"""One whole fitting step of the reference, phase by phase: the step body of /root/reference/pipeline/train.py — the
four ``render()`` calls with the view swap (:348-393), the f/b flip-average (:368-393), every loss term with its weight and
denominator (:407-460), ``loss.backward()`` (:462) and the four ``training_statis`` calls (:560-565) — is READ from the reference's
file at generation time, compiled and executed on PyTorch-CPU at production dimensions with oracle/ in the two native slots
(tests/golden/_ref_import.py).  Nothing of that text is stored: the fixture holds numbers only — the loss, each term, the two
averaged images (every other row), strided rows + float64 sums of every parameter gradient, and the densification accumulators.

One case per phase of the schedule (reference utils/train_util.py:20-41): FULL_PRECISION, QUANTIZED, TRAINING_ENTROPY, STE_ENTROPY,
at an iteration inside the phase under the reference's default OptimizationParams (lmbda = 0.004 as BASELINE.json's configs[2];
opacity_reg non-zero so that its weight is pinned).  Model, anchors, frames, flow and the random draws are regenerated from seeds on
both sides (tests/golden/seeded.py).

Runs in the build container only (needs /root/reference).  Usage:  python tests/golden/make_golden_step.py
"""
import os
import sys
import textwrap
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import _ref_import, seeded  # noqa: E402
from tests.golden.make_golden_common import grads_of, save  # noqa: E402



def reference_slice(first_marker: str, last_marker: str):
    """Code object of the reference's step body between two marker lines (inclusive), dedented — read from the reference's file
    here and now, never stored."""
    path = os.path.join(_ref_import.REF, "pipeline", "train.py")
    lines = open(path, encoding="utf-8").read().splitlines()
    start = next(i for i, ln in enumerate(lines) if ln.strip() == first_marker)
    end = next(i for i in range(start, len(lines)) if lines[i].strip() == last_marker)
    body = textwrap.dedent("\n".join(lines[start:end + 1]))
    print(f"reference pipeline/train.py:{start + 1}-{end + 1}")
    return compile(body, f"<reference pipeline/train.py:{start + 1}-{end + 1}>", "exec")


def main():
    mode_ctx = _ref_import.install(rasterizer=True)
    with mode_ctx:
        import arguments as A
        import scene.gaussian_model as GM
        import ortho_gaussian_renderer as OGR
        from ortho_gaussian_renderer import GenerateMode
        from frame_cube.frame import Frame
        from utils.encodings import get_binary_vxl_size
        from utils.loss_utils import calc_optical_loss, l1_loss_func, ssim_func
        from utils.train_util import TrainingController

        sc, P, ST = seeded.SCENE, seeded.PROD, seeded.STEP
        H, W, T, idx = sc["H"], sc["W"], sc["T"], sc["frame"]
        fn1, fn2 = seeded.frame_numbers(H, W, T, idx), seeded.frame_numbers(H, W, T, idx + 1)
        mp = A.ModelParams()
        mp.threshold = sc["threshold"]
        opt = A.OptimizationParams()
        opt.lmbda, opt.opacity_reg = ST["lmbda"], ST["opacity_reg"]
        torch.manual_seed(0)
        ref = GM.GaussianModel(mp, feat_dim=P["feat_dim"], n_offsets=P["n_offsets"], voxel_size=0.001, update_depth=3,
                               update_init_factor=16, update_hierachy_factor=4, use_feat_bank=False,
                               n_features_per_level=P["n_features_per_level"], log2_hashmap_size=P["log2_hashmap_size"],
                               log2_hashmap_size_2D=P["log2_hashmap_size_2D"], resolutions_list=P["resolutions_list"],
                               resolutions_list_2D=P["resolutions_list_2D"])
        ref.update_anchor_bound(fn1["x_min"], fn1["y_min"], fn1["z_min"])
        for name, t in seeded.anchors(sc["A"], fn1, sc["threshold"], sc["seed"]).items():
            setattr(ref, name, nn.Parameter(t, requires_grad=name not in ("_rotation", "_opacity")))
        seeded.fill_parameters(ref, sc["seed"])
        ref.spatial_lr_scale = 1.0

        def frame_of(fn, i):
            # images are stored transposed [3, W, H] and turned back at use (reference frame_cube/frame.py:145, pipeline/train.py:407)
            return Frame(image_id=i, plane="xy", image=seeded.gt_image(H, W, i, sc["seed"]).permute(0, 2, 1).contiguous(), x_min=fn["x_min"],
                         y_min=fn["y_min"], z=fn["z"], image_width=W, image_height=H, view_matrix=fn["view_matrix"].clone(),
                         view_matrix_s=fn["view_matrix_s"].clone(), scale=fn["scale"], cam_pos=fn["cam_pos"].clone())

        dataset = SimpleNamespace(x_min=fn1["x_min"], y_min=fn1["y_min"], scale=fn1["scale"], width=W, height=H)
        step_code = reference_slice("render_mode = controller.render_mode", "loss.backward()")
        statis_code = reference_slice("if controller.gaussian_statis:", "gaussians.training_statis(render_results2_b)")
        out = {"meta::A": np.int64(sc["A"]), "meta::lmbda": np.float64(opt.lmbda), "meta::opacity_reg": np.float64(opt.opacity_reg),
               "meta::weights": np.array([opt.lambda_dssim, opt.scaling_reg, opt.opacity_reg, opt.optical_lambda, opt.lmbda], dtype=np.float64)}

        # (mode, iteration, white background): the four phases on black, and the full-precision phase once more on a WHITE background
        # (reference pipeline/train.py:327 `bg_color = [1, 1, 1] if model_params.white_background`): out = C + T bg and the
        # background's share of dL/dalpha (the compositing backward's non-zero-background form) inside a whole step
        cases = [(m, it, False) for m, it in sorted(ST["iterations"].items())] + [(0, ST["iterations"][0], True)]
        for mode_value, iteration, white in cases:
            pre = f"m{mode_value}w::" if white else f"m{mode_value}::"
            ref.training_setup(opt)                                   # fresh optimiser + zeroed densification accumulators
            for p in ref.parameters():
                p.grad = None
            controller = TrainingController(opt)
            controller.current_iteration = iteration
            assert controller.render_mode == GenerateMode(mode_value), (controller.render_mode, mode_value)
            ns = dict(render=OGR.render, frame1=frame_of(fn1, idx), frame2=frame_of(fn2, idx + 1), gaussians=ref,
                      pipe=SimpleNamespace(debug=False, compute_cov3D_python=False, model_path=None),
                      background=torch.tensor([1.0, 1.0, 1.0] if white else [0.0, 0.0, 0.0]),
                      controller=controller, opt=opt, iteration=iteration, torch=torch, l1_loss_func=l1_loss_func, ssim_func=ssim_func,
                      calc_optical_loss=calc_optical_loss, get_binary_vxl_size=get_binary_vxl_size,
                      optical_flow=seeded.optical_flow(H, W, idx, sc["seed"]), frame_cube=SimpleNamespace(dataset=dataset))
            with seeded.SeededDraws(5000 + 100 * mode_value) as draws:
                exec(step_code, ns)
            out[pre + "n_draws"] = np.int64(draws.count)
            out[pre + "iteration"] = np.int64(iteration)
            rr = [ns[k] for k in ("render_results1_f", "render_results1_b", "render_results2_f", "render_results2_b")]
            out[pre + "loss"] = ns["loss"]
            for nm in ("Ll1", "ssim_loss", "scaling_reg", "opacity_reg", "optical_loss"):
                out[pre + nm] = ns[nm]
            if controller.entropy_constrained:
                for nm in ("bit_per_param", "bit_hash_grid"):
                    out[pre + nm] = ns[nm]
                out[pre + "denom"] = np.float64(ns["denom"])
                out[pre + "mask_reg"] = torch.mean(torch.sigmoid(ref._mask))
                out[pre + "rates"] = np.array([[float(getattr(r, nm)) for nm in ("bit_per_param", "bit_per_feat_param", "bit_per_scaling_param",
                                                                                  "bit_per_offsets_param")] for r in rr])
            out[pre + "image1"], out[pre + "image2"] = ns["image1"][:, ::2], ns["image2"][:, ::2]
            out[pre + "counts"] = np.array([[int(r.visible_mask.sum()), r.generated_gaussians.xyz.shape[0], int(r.active_gaussains),
                                             int(r.num_rendered)] for r in rr], dtype=np.int64)
            out[pre + "visible_masks"] = np.stack([np.packbits(r.visible_mask.numpy()) for r in rr])
            out[pre + "retain_grad"] = np.bool_(ns["retain_grad"])
            grads_of(ref.named_parameters(), out, pre, rows=mode_value in (0, 2) and not white)      # modes 1, 3 and the white case: sums only (fixture size)
            with torch.no_grad():
                exec(statis_code, ns)
            out[pre + "gaussian_statis"] = np.bool_(controller.gaussian_statis)
            for nm in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"):
                t = getattr(ref, nm)
                out[pre + "statis::" + nm] = t[::8]
                out[pre + "statis_sum::" + nm] = np.array([float(t.double().sum()), float(t.double().abs().sum())])
            print(pre, "loss", float(ns["loss"]), "L1", float(ns["Ll1"]), "ssim", float(ns["ssim_loss"]), "optical", float(ns["optical_loss"]),
                  "draws", draws.count, "statis", bool(controller.gaussian_statis), "counts", out[pre + "counts"].tolist())
        for name, t in sorted(ref.state_dict().items()):
            if t.is_floating_point() and t.numel():
                out["param_sum::" + name] = np.array([float(t.double().sum()), float(t.double().abs().sum())])
        save("step_fixture", **out)
    print("done")


if __name__ == "__main__":
    main()
