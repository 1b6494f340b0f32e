"""Does GSVC_DETERMINISTIC=1 compute the SAME gradients as the default mode (up to summation order)?  One step per phase from the same
parameters, seeds and plan in both modes: per parameter max |g_det - g_def| / max |g_def|.  Anything above ~1e-4 would be a different
quantity, not a different order.  Usage: python tools/ab/det_vs_default_grads.py [anchors=120000]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gsvc_amd  # noqa: E402,F401
from gsvc_amd import switches  # noqa: E402
from gsvc_amd.arguments import cfg_20240919  # noqa: E402
from gsvc_amd.frame import SyntheticFrameCube  # noqa: E402
from gsvc_amd.model import GaussianModel  # noqa: E402
from gsvc_amd.train import Trainer  # noqa: E402


def main():
    A = int(sys.argv[1]) if len(sys.argv) > 1 else 120_000
    dev = torch.device("cuda", 0)
    mp_, opt, pipe = cfg_20240919()
    cube = SyntheticFrameCube(1080, 1920, 64, seed=1234, device=dev).materialize()
    mp_.threshold = 8.0 / cube.scale
    opt.start_stat, opt.update_until, opt.pause_densification, opt.update_from = 0, 10 ** 9, 0, 10 ** 9
    B = 10 ** 9
    torch.manual_seed(0)
    np.random.seed(0)
    pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                       mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                       log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device=dev)
    lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
    pc.create_from_points(np.random.default_rng(0).uniform(lim, -lim, (A, 3)), spatial_lr_scale=1.0)
    pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
    (opt.full_precision_training_total, opt.quantized_training_total, opt.entropy_constrained_train_total,
     opt.ste_entropy_constrained_train_total) = 0, 0, B, 0
    pc.training_setup(opt)
    tr = Trainer(pc, cube, opt, pipe, mp_, seed=0)
    for it in range(1, 61):
        tr.step(it)
    torch.cuda.synchronize()
    captured = {}
    stats = {}

    def capture(*a, **k):
        if k.get("only") is not None:
            return None
        captured.clear()
        for n, p in pc.named_parameters():
            if p.grad is not None:
                captured[n] = p.grad.detach().clone()
    pc.optimizer.step = capture
    PHASES = {"FULL": (B, 0, 0, 0), "QUANT": (0, B, 0, 0), "ENTROPY": (0, 0, B, 0), "STE": (0, 0, 0, B)}
    worst = 0.0
    for planned in (False, True):
        for phase, totals in PHASES.items():
            (opt.full_precision_training_total, opt.quantized_training_total, opt.entropy_constrained_train_total,
             opt.ste_entropy_constrained_train_total) = totals
            res = {}
            for det in (False, True):
                if det:
                    os.environ["GSVC_DETERMINISTIC"] = "1"
                else:
                    os.environ.pop("GSVC_DETERMINISTIC", None)
                switches.reload()
                for n in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"):
                    getattr(pc, n).zero_()
                tr._plan = tr._plan_idx = None
                tr.rng.seed(7)
                torch.manual_seed(1234)
                tr.controller.current_iteration = 100
                tr.controller._entropy_constrained = False      # (sticky, as the reference's: the warm-up ran in the entropy phase)
                if planned:
                    from gsvc_amd.ortho_gaussian_renderer import plan_views
                    with torch.no_grad():
                        tr._plan_idx, tr._plan_mode = 20, tr.controller.render_mode
                        tr._plan = plan_views(tr._views(20), pc, pipe, tr.background, tr._plan_mode)
                out = tr.step(100, frame_idx=20)
                torch.cuda.synchronize()
                res[det] = ({k: v.clone() for k, v in captured.items()}, float(out.loss),
                            {n: getattr(pc, n).clone() for n in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom")})
            os.environ.pop("GSVC_DETERMINISTIC", None)
            switches.reload()
            g0, l0, s0 = res[False]
            g1, l1, s1 = res[True]
            rows = []
            for n in sorted(g0):
                if n not in g1:
                    rows.append((float("inf"), n + " MISSING in deterministic"))
                    continue
                scale = float(g0[n].abs().max())
                d = float((g0[n] - g1[n]).abs().max())
                rows.append((d / scale if scale > 0 else (0.0 if d == 0 else float("inf")), n))
            for n in g1:
                if n not in g0:
                    rows.append((float("inf"), n + " MISSING in default"))
            for n in s0:
                scale = float(s0[n].abs().max())
                d = float((s0[n].float() - s1[n].float()).abs().max())
                rows.append((d / scale if scale > 0 else (0.0 if d == 0 else float("inf")), "accumulator " + n))
            rows.sort(reverse=True)
            worst = max(worst, rows[0][0])
            print(f"{'planned' if planned else 'plan-less'} {phase:8s}: loss {l0:.9f} vs {l1:.9f}; {len(g0)} gradient tensors; largest relative "
                  f"differences: " + ", ".join(f"{n} {r:.2e}" for r, n in rows[:4]), flush=True)
    print(f"largest relative difference over all phases and tensors: {worst:.3e}")
    tr.close()


if __name__ == "__main__":
    main()
