"""cProfile of one adjust_anchor call after N fitting steps (host time; the call synchronises often)."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
from gsvc_amd.train import Trainer
dev = torch.device("cuda:0")
mp_, opt, pipe = cfg_20240919()
cube = SyntheticFrameCube(1080, 1920, 64, seed=1234, device=dev).materialize()
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
pc.create_from_points(rng.uniform(lim, -lim, (245_000, 3)), spatial_lr_scale=1.0)
pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
pc.training_setup(opt)
tr = Trainer(pc, cube, opt, pipe, mp_, seed=0)
for i in range(1, 121):
    tr.step(i)
torch.cuda.synchronize()
kw = dict(check_interval=int(os.environ.get("CHECK", "5")), success_threshold=0.8, grad_threshold=float(os.environ.get("GRAD", "0.0002")), min_opacity=0.005)
for rep in range(3):
    a0 = pc._anchor.shape[0]
    pr = cProfile.Profile()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr.enable(); pc.adjust_anchor(**kw); torch.cuda.synchronize(); pr.disable()
    print(f"adjust_anchor #{rep}: {1e3 * (time.perf_counter() - t0):.1f} ms, anchors {a0} -> {pc._anchor.shape[0]}")
    if rep == 1:
        pstats.Stats(pr).sort_stats("tottime").print_stats(18)
    for i in range(20):
        tr.step(200 + 20 * rep + i)
    torch.cuda.synchronize()
