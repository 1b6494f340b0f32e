"""WHICH piece of a fitting step makes two identical runs differ?  Every autograd function of gsvc_amd gets its forward and backward
wrapped: an exact integer checksum (sum of the float32 bit patterns as int64: order-independent) of every tensor going in and coming
out is logged per call.  The same step is run REP times; the first call whose INPUTS agree across the runs and whose OUTPUTS do not
is the source (later differences are consequences).  Usage: python tools/ab/determinism_probe.py [anchors] [repeats]
Environment: DET_PHASES=FULL,QUANT,ENTROPY,STE (default all); any GSVC_* switch applies (GSVC_DETERMINISTIC=1: expect no source)."""
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

LOG = []
ON = [False]


def _sum(t):
    if not isinstance(t, torch.Tensor) or t.numel() == 0:
        return None
    t = t.detach()
    if t.dtype == torch.float32:
        return int(t.contiguous().view(torch.int32).to(torch.int64).sum())
    if t.dtype in (torch.bool, torch.uint8, torch.int32, torch.int64):
        return int(t.to(torch.int64).sum())
    return int(t.float().contiguous().view(torch.int32).to(torch.int64).sum())


def _sums(xs):
    return tuple(_sum(x) for x in xs)


def wrap(cls):
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *a, **k):
        out = fwd(ctx, *a, **k)
        if ON[0]:
            torch.cuda.synchronize()
            LOG.append((cls.__name__ + ".fwd", _sums(a), _sums(out if isinstance(out, tuple) else (out,))))
        return out

    def backward(ctx, *g):
        out = bwd(ctx, *g)
        if ON[0]:
            torch.cuda.synchronize()
            LOG.append((cls.__name__ + ".bwd", _sums(g), _sums(out if isinstance(out, tuple) else (out,))))
        return out
    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)


def wrap_all():
    import importlib
    n = 0
    for name in ("generate", "mlp", "rasterizer", "encodings", "entropy_models", "loss_utils", "model", "train", "time_util",
                 "ortho_gaussian_renderer.renderer", "ortho_gaussian_renderer.preprocess", "gridencoder_backend"):
        try:
            m = importlib.import_module("gsvc_amd." + name)
        except Exception:  # noqa: BLE001
            continue
        for v in list(vars(m).values()):
            if isinstance(v, type) and issubclass(v, torch.autograd.Function) and v is not torch.autograd.Function and not getattr(v, "_probed", False):
                wrap(v)
                v._probed = True
                n += 1
    return n


def main():
    A = int(sys.argv[1]) if len(sys.argv) > 1 else 245_000
    REP = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda", 0)
    mp_, opt, pipe = cfg_20240919()
    cube = SyntheticFrameCube(1080, 1920, 64, seed=1234, device=dev).materialize()
    mp_.threshold = 8.0 / cube.scale
    opt.start_stat, opt.update_until, opt.pause_densification, opt.update_from = 10 ** 9, 10 ** 9, 0, 10 ** 9
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
     opt.ste_entropy_constrained_train_total) = B, 0, 0, 0
    pc.training_setup(opt)
    tr = Trainer(pc, cube, opt, pipe, mp_, seed=0)
    for it in range(1, 41):
        tr.step(it)
    torch.cuda.synchronize()
    print("wrapped autograd functions:", wrap_all(), flush=True)
    captured = {}

    def capture(*a, **k):
        if k.get("only") is not None:
            return None
        captured.clear()
        for n, p in pc.named_parameters():
            if p.grad is not None:
                captured[n] = _sum(p.grad)
    pc.optimizer.step = capture
    PHASES = {"FULL": (B, 0, 0, 0), "QUANT": (0, B, 0, 0), "ENTROPY": (0, 0, B, 0), "STE": (0, 0, 0, B)}
    only = os.environ.get("DET_PHASES")
    for phase, totals in PHASES.items():
        if only and phase not in only.split(","):
            continue
        (opt.full_precision_training_total, opt.quantized_training_total, opt.entropy_constrained_train_total,
         opt.ste_entropy_constrained_train_total) = totals
        logs, grads = [], []
        for rep in range(REP + 1):
            tr._plan = tr._plan_idx = None
            tr.rng.seed(7)
            torch.manual_seed(1234)
            tr.controller.current_iteration = 100
            LOG.clear()
            ON[0] = rep > 0
            if os.environ.get("DET_PLAN"):          # the production form: a plan built ahead (ranked gathers, the plan's rate sample)
                from gsvc_amd.ortho_gaussian_renderer import plan_views
                with torch.no_grad():
                    tr._plan_idx, tr._plan_mode = 20, tr.controller.render_mode
                    tr._plan = plan_views(tr._views(20), pc, pipe, tr.background, tr._plan_mode)
            tr.step(100, frame_idx=20)
            torch.cuda.synchronize()
            ON[0] = False
            if rep:
                logs.append(list(LOG))
                grads.append(dict(captured))
        bad_params = sorted(k for k in grads[0] if any(g.get(k) != grads[0][k] for g in grads[1:]))
        print(f"{phase}: {len(logs[0])} calls logged; parameters whose gradient checksum differs: {len(bad_params)} {bad_params[:8]}", flush=True)
        if any(len(l) != len(logs[0]) or [c[0] for c in l] != [c[0] for c in logs[0]] for l in logs[1:]):
            print(f"{phase}: the call sequences differ between runs ({[len(l) for l in logs]})", flush=True)
            continue
        sources = consequences = 0
        for i, call in enumerate(logs[0]):
            ins_eq = all(l[i][1] == call[1] for l in logs[1:])
            outs_eq = all(l[i][2] == call[2] for l in logs[1:])
            if not outs_eq:
                which = [j for j in range(len(call[2])) if any(l[i][2][j] != call[2][j] for l in logs[1:])]
                if ins_eq:
                    sources += 1
                    print(f"  SOURCE   #{i:3d} {call[0]:28s} outputs {which} differ with identical inputs", flush=True)
                else:
                    consequences += 1
                    if consequences <= 6:
                        print(f"  follows  #{i:3d} {call[0]:28s} outputs {which} (inputs differ too)", flush=True)
        print(f"{phase}: {sources} source call(s), {consequences} downstream", flush=True)
    tr.close()


if __name__ == "__main__":
    main()
