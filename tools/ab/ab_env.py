"""A/B of an environment switch inside ONE process (same model trajectory): alternating timed segments of fitting steps.
usage: python tools/ab/ab_env.py VAR valueA valueB"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
from gsvc_amd.train import Trainer
var, va, vb = sys.argv[1:4]
if os.environ.get("GSVC_AB_BIND_EARLY"):
    torch.cuda.set_device(0)
    from gsvc_amd.hostbind import bind_to_device
    print("bound early:", bind_to_device(0), len(os.sched_getaffinity(0)))
dev = torch.device("cuda:0")
mp_, opt, pipe = cfg_20240919()
CFG3 = "cfg3" in sys.argv[1:] or bool(os.environ.get("GSVC_AB_CFG3"))      # BASELINE configs[3] per-GPU shape: yaml as is
cube = SyntheticFrameCube(1080, 1920, 600 if CFG3 else 64, seed=1234, device=dev).materialize()
if not CFG3:
    mp_.threshold = 8.0 / cube.scale
opt.full_precision_training_total, opt.quantized_training_total = 0, 0
opt.entropy_constrained_train_total = 10 ** 9
opt.start_stat, opt.update_until, opt.pause_densification = 0, 10 ** 9, 0
torch.manual_seed(0); np.random.seed(0)
pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                   mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                   log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device=dev)
rng = np.random.default_rng(0)
lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
pc.create_from_points(rng.uniform(lim, -lim, (100_000 if CFG3 else 245_000, 3)), spatial_lr_scale=1.0)
pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
pc.training_setup(opt)
tr = Trainer(pc, cube, opt, pipe, mp_, seed=0)
it = 0
if os.environ.get('GSVC_AB_PRE_OFF'):
    os.environ[var] = os.environ['GSVC_AB_PRE_OFF']
    from gsvc_amd import switches as _sw
    _sw.reload()
for _ in range(int(os.environ.get('GSVC_AB_PRE', '150'))):
    it += 1; tr.step(it)
res = {va: [], vb: []}
for rep in range(int(os.environ.get('GSVC_AB_REPS', '5'))):
    for v in (va, vb):
        if v == "unset":
            os.environ.pop(var, None)
        else:
            os.environ[var] = v
        from gsvc_amd import switches
        switches.reload()
        for _ in range(3):
            it += 1; tr.step(it)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        act = torch.zeros((), device=dev, dtype=torch.float64)
        for _ in range(int(os.environ.get('GSVC_AB_SEG', '25'))):
            it += 1; keep = tr.step(it) if os.environ.get("GSVC_AB_KEEP") else None; act += (keep if keep is not None else tr.step(it)).active_gaussians
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / int(os.environ.get('GSVC_AB_SEG', '25'))
        res[v].append((ms, float(act) / (4 * int(os.environ.get('GSVC_AB_SEG', '25')))))
for v, r in res.items():
    ms = np.array([x[0] for x in r]); a = np.array([x[1] for x in r])
    print(f"{var}={v}: {ms.mean():.3f} ms/step (min {ms.min():.3f}, max {ms.max():.3f}), active per render {a.mean():.0f}, us per 1000 active {1e3 * ms.mean() / (4 * a.mean() / 1e3):.2f}")
