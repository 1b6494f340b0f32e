"""Is the fitting step host-bound?  Per step: host time until Trainer.step returns, time until the device is idle, and the
synchronising calls inside the step (torch sync debug mode)."""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
from gsvc_amd.train import Trainer
dev = torch.device("cuda:0")
mp_, opt, pipe = cfg_20240919()
CFG3 = "cfg3" in sys.argv[1:] or bool(os.environ.get("GSVC_AB_CFG3"))      # BASELINE configs[3] per-GPU shape: yaml as is
cube = SyntheticFrameCube(1080, 1920, 600 if CFG3 else 64, seed=1234, device=dev).materialize()
if not CFG3:
    mp_.threshold = 8.0 / cube.scale
opt.full_precision_training_total, opt.quantized_training_total = 0, 0
opt.entropy_constrained_train_total = 10 ** 9
opt.start_stat, opt.update_until = 0, 10 ** 9
opt.pause_densification = 0
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
for i in range(1, 41):
    tr.step(i)
torch.cuda.synchronize()
host, total = [], []
for i in range(41, 71):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0); total.append(t2 - t0)
print(f"synced every step: host returns after {1e3*np.median(host):.2f} ms, device idle after {1e3*np.median(total):.2f} ms")
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(71, 121):
    tr.step(i)
torch.cuda.synchronize()
print(f"free running: {1e3*(time.perf_counter()-t0)/50:.2f} ms/step")
if os.environ.get("SYNC_DEBUG"):
    torch.cuda.set_sync_debug_mode("warn")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        tr.step(121)
    torch.cuda.set_sync_debug_mode("default")
    import traceback
    for x in w:
        print("SYNC:", x.filename, x.lineno, str(x.message)[:80])
import cProfile, pstats
if os.environ.get("CPROF"):
    pr = cProfile.Profile(); pr.enable()
    for i in range(122, 142):
        tr.step(i)
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(35)
