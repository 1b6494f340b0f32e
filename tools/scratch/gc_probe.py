"""Does a fitting step leave reference cycles behind (garbage only the cyclic collector frees)?"""
import gc, os, sys, time
os.environ["GSVC_KEEP_GC"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
from gsvc_amd.train import Trainer
dev = torch.device("cuda:0")
mp_, opt, pipe = cfg_20240919()
cube = SyntheticFrameCube(1080, 1920, 600, seed=1234, device=dev).materialize()
opt.full_precision_training_total, opt.quantized_training_total = 0, 0
opt.entropy_constrained_train_total = 10 ** 9
opt.start_stat, opt.update_until, opt.pause_densification = 0, 10 ** 9, 0
torch.manual_seed(0); np.random.seed(0)
pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                   mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                   log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device=dev)
rng = np.random.default_rng(0)
lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
pc.create_from_points(rng.uniform(lim, -lim, (100_000, 3)), spatial_lr_scale=1.0)
pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
pc.training_setup(opt)
tr = Trainer(pc, cube, opt, pipe, mp_, seed=0)
for i in range(1, 41):
    tr.step(i)
torch.cuda.synchronize()
it = 40
res = {True: [], False: []}
for rep in range(6):
    for on in (True, False):
        if on:
            gc.unfreeze(); gc.enable()
        else:
            gc.collect(); gc.freeze(); gc.disable()
        for _ in range(5):
            it += 1; tr.step(it)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            it += 1; tr.step(it)
        torch.cuda.synchronize()
        res[on].append(1e3 * (time.perf_counter() - t0) / 40)
for on, v in res.items():
    print("gc enabled" if on else "gc frozen+disabled", " ".join(f"{x:.3f}" for x in v), f"mean {np.mean(v):.3f}")
