"""Host wall-clock per region of the fitting step (GSVC_HOST_TIMES=1: gsvc_amd.generate.region), free-running steps.
usage: GSVC_HOST_TIMES=1 python tools/ab/host_regions.py [cfg3]"""
import os, sys, time
os.environ["GSVC_HOST_TIMES"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsvc_amd import generate as G
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
from gsvc_amd.train import Trainer
dev = torch.device("cuda:0")
mp_, opt, pipe = cfg_20240919()
CFG3 = "cfg3" in sys.argv[1:]
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
for i in range(1, 61):
    tr.step(i)
torch.cuda.synchronize()
if os.environ.get("NOGC"):
    import gc
    gc.collect(); gc.freeze(); gc.disable()
G.HOST_TIMES.clear()
N = 100
t0 = time.perf_counter()
th = 0.0
for i in range(61, 61 + N):
    a = time.perf_counter()
    tr.step(i)
    th += time.perf_counter() - a
torch.cuda.synchronize()
print(f"free running: {1e3 * (time.perf_counter() - t0) / N:.3f} ms/step; host inside Trainer.step {1e3 * th / N:.3f} ms/step")
for k, (c, s) in sorted(G.HOST_TIMES.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:28s} {1e3 * s / N:7.3f} ms/step  ({c / N:.1f} calls/step)")
