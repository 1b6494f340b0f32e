"""cProfile of the fitting step on BOTH host threads: the main thread (forward, optimizer) and the autograd engine's thread (the
custom backward functions), by cumulative and by own time.  usage: python tools/ab/host_prof2.py [cfg3]"""
import cProfile, os, pstats, sys, time
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
for i in range(1, 41):
    tr.step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(41, 91):
    tr.step(i)
torch.cuda.synchronize()
print(f"free running: {1e3 * (time.perf_counter() - t0) / 50:.2f} ms/step")
prm, prb = cProfile.Profile(), cProfile.Profile()
orig = torch.Tensor.backward
def bw(self, *a, **k):
    self.register_hook(lambda g: (prb.enable(), None)[1])      # runs first on the engine's thread
    return orig(self, *a, **k)
torch.Tensor.backward = bw
N = 30
prm.enable()
for i in range(91, 91 + N):
    tr.step(i)
prm.disable(); torch.cuda.synchronize()
torch.Tensor.backward = orig
for name, pr in (("MAIN THREAD", prm), ("AUTOGRAD THREAD", prb)):
    for key in ("cumulative", "tottime"):
        print(f"==== {name} by {key} (ms per step = seconds * {1e3 / N:.1f})")
        pstats.Stats(pr).sort_stats(key).print_stats(28 if key == "cumulative" else 22)
