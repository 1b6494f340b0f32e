"""Fitting-step time per phase of the schedule (FULL_PRECISION, QUANTIZED, TRAINING_ENTROPY, STE_ENTROPY) at the headline shape,
one process, same model.  usage: python tools/ab/mode_times.py [cfg3]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
opt.start_stat, opt.update_until, opt.pause_densification = 0, 10 ** 9, 0
opt.update_from = 10 ** 9
# phases of 200 iterations each, entropy first 150 as pre-training so that the opacity masks look like a fitted model's
opt.full_precision_training_total, opt.quantized_training_total = 10 ** 9, 0
opt.entropy_constrained_train_total, opt.ste_entropy_constrained_train_total = 0, 0
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
for _ in range(150):
    it += 1; tr.step(it)
def phase(name, fp, q, e, s):
    global it
    opt.full_precision_training_total, opt.quantized_training_total = fp, q
    opt.entropy_constrained_train_total, opt.ste_entropy_constrained_train_total = e, s
    for _ in range(8):
        it += 1; tr.step(it)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    act = torch.zeros((), device=dev, dtype=torch.float64)
    n = 40
    for _ in range(n):
        it += 1; act += tr.step(it).active_gaussians
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print(f"{name:18s} {ms:7.3f} ms/step   active per render {float(act) / n / 4:9.0f}", flush=True)
B = 10 ** 9
PH = {"FULL_PRECISION": (B, 0, 0, 0), "QUANTIZED": (0, B, 0, 0), "TRAINING_ENTROPY": (0, 0, B, 0), "STE_ENTROPY": (0, 0, 0, B)}
only = [a for a in sys.argv[1:] if a in PH]
for name in (only or ["FULL_PRECISION", "QUANTIZED", "TRAINING_ENTROPY", "STE_ENTROPY", "TRAINING_ENTROPY"]):
    phase(name, *PH[name])
