"""Is one fitting step a deterministic function of (model, frame pair, seed)?  The same step — optimizer replaced by a gradient capture, the
random generator re-seeded — is run several times in one process and every parameter's gradient compared bit for bit, per phase and
under the step's structural switches (one raster stream, no small-work stream, no early plan, layer-by-layer MLPs ...), to find which
piece, if any, makes two identical runs differ.  Usage: python tools/ab/determinism_check.py [anchors] [repeats]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsvc_amd import switches
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
from gsvc_amd.train import Trainer

A = int(sys.argv[1]) if len(sys.argv) > 1 else 245_000
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
mp_, opt, pipe = cfg_20240919()
cube = SyntheticFrameCube(1080, 1920, 64, seed=1234, device=dev).materialize()
mp_.threshold = 8.0 / cube.scale
opt.start_stat, opt.update_until, opt.pause_densification, opt.update_from = 10 ** 9, 10 ** 9, 0, 10 ** 9
B = 10 ** 9
torch.manual_seed(0); np.random.seed(0)
pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                   mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                   log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device=dev)
lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
pc.create_from_points(np.random.default_rng(0).uniform(lim, -lim, (A, 3)), spatial_lr_scale=1.0)
pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
(opt.full_precision_training_total, opt.quantized_training_total, opt.entropy_constrained_train_total,
 opt.ste_entropy_constrained_train_total) = B, 0, 0, 0
pc.training_setup(opt)
tr = Trainer(pc, cube, opt, pipe, mp_, seed=0)
for it in range(1, 41):          # a model with non-degenerate opacities / scales
    tr.step(it)
torch.cuda.synchronize()
real_step = pc.optimizer.step
captured = {}


def capture(*a, **k):
    if k.get("only") is not None:          # the guarded early update: let it through as a no-op (its tensors keep their gradients)
        return None
    captured.clear()
    for n, p in pc.named_parameters():
        if p.grad is not None:
            captured[n] = p.grad.detach().clone()


pc.optimizer.step = capture
PHASES = {"FULL": (B, 0, 0, 0), "QUANT": (0, B, 0, 0), "ENTROPY": (0, 0, B, 0), "STE": (0, 0, 0, B)}
VARIANTS = [("default", {}), ("GSVC_DETERMINISTIC=1", {"GSVC_DETERMINISTIC": "1"}),
            ("default, planned step", {"_PLAN": "1"}), ("GSVC_DETERMINISTIC=1, planned step", {"GSVC_DETERMINISTIC": "1", "_PLAN": "1"}), ("one raster stream", {"GSVC_RASTER_STREAMS": "1"}), ("no small-work stream", {"GSVC_NO_RATE_OVERLAP": "1"}),
            ("no early plan", {"GSVC_NO_EARLY_PLAN": "1"}), ("no prefetch", {"GSVC_NO_PREFETCH": "1"}),
            ("one stream, no overlap, no prefetch", {"GSVC_RASTER_STREAMS": "1", "GSVC_NO_RATE_OVERLAP": "1", "GSVC_NO_PREFETCH": "1"}),
            ("layer-by-layer MLPs", {"GSVC_NO_MLP_CHAIN": "1"}), ("no multi-product launches", {"GSVC_NO_SHARED_INPUT": "1", "GSVC_NO_ACCUM_MANY": "1"})]
only = os.environ.get("DET_PHASES")
if os.environ.get("DET_VARIANTS"):          # e.g. DET_VARIANTS=default,GSVC_DETERMINISTIC=1
    VARIANTS = [v for v in VARIANTS if any(v[0].startswith(w) for w in os.environ["DET_VARIANTS"].split(","))]
for phase, totals in PHASES.items():
    if only and phase not in only.split(","):
        continue
    (opt.full_precision_training_total, opt.quantized_training_total, opt.entropy_constrained_train_total,
     opt.ste_entropy_constrained_train_total) = totals
    for tag, env in VARIANTS:
        env = dict(env)
        planned = bool(env.pop("_PLAN", None))
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        switches.reload()
        try:
            tr.prefetch = not switches.NO_PREFETCH
            runs = []
            for rep in range(REP + 1):          # the first run warms allocator and plan up and is not compared
                tr._plan = tr._plan_idx = None
                tr.rng.seed(7)
                torch.manual_seed(1234)
                tr.controller.current_iteration = 100
                if planned:          # the production form: the step runs from a plan built ahead (ranked gathers, the plan's rate sample)
                    from gsvc_amd.ortho_gaussian_renderer import plan_views
                    with torch.no_grad():
                        tr._plan_idx, tr._plan_mode = 20, tr.controller.render_mode
                        tr._plan = plan_views(tr._views(20), pc, pipe, tr.background, tr._plan_mode)
                tr.step(100, frame_idx=20)
                torch.cuda.synchronize()
                if rep:
                    runs.append({k: v.clone() for k, v in captured.items()})
            bad = {}
            for k in runs[0]:
                d = max(float((r[k] - runs[0][k]).abs().max()) for r in runs[1:])
                if d != 0.0:
                    n = max(int((r[k] != runs[0][k]).sum()) for r in runs[1:])
                    bad[k] = (d / max(float(runs[0][k].abs().max()), 1e-30), n, runs[0][k].numel())
            worst = sorted(bad.items(), key=lambda kv: -kv[1][0])[:6]
            print(f"{phase:8s} {tag:40s} tensors {len(runs[0])} differing {len(bad)}" +
                  ("  worst: " + "; ".join(f"{k} {v[0]:.1e} of scale in {v[1]}/{v[2]}" for k, v in worst) if bad else "  bit-identical"), flush=True)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            switches.reload()

# what the fixed order costs: the entropy-constrained step, default against GSVC_DETERMINISTIC=1, same process
import time
pc.optimizer.step = real_step
(opt.full_precision_training_total, opt.quantized_training_total, opt.entropy_constrained_train_total,
 opt.ste_entropy_constrained_train_total) = PHASES["ENTROPY"]
tr._plan = tr._plan_idx = None
for tag, val in (("default", None), ("GSVC_DETERMINISTIC=1", "1"), ("default", None), ("GSVC_DETERMINISTIC=1", "1")):
    os.environ.pop("GSVC_DETERMINISTIC", None) if val is None else os.environ.__setitem__("GSVC_DETERMINISTIC", val)
    switches.reload()
    for it in range(200, 206):
        tr.step(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(206, 236):
        tr.step(it)
    torch.cuda.synchronize()
    print(f"COST {tag:24s} {1e3 * (time.perf_counter() - t0) / 30:.3f} ms per step (TRAINING_ENTROPY, {A} anchors)", flush=True)
os.environ.pop("GSVC_DETERMINISTIC", None)
switches.reload()
