"""Sanity: 400 fitting steps of the bench model at 1080p through every phase boundary incl. densification; loss finite and falling."""
import os, sys, time
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
opt.full_precision_training_total, opt.quantized_training_total = 100, 50
opt.entropy_constrained_train_total, opt.ste_entropy_constrained_train_total = 200, 50
opt.start_stat, opt.update_from, opt.update_interval, opt.update_until, opt.pause_densification = 20, 60, 40, 260, 0
torch.manual_seed(0); np.random.seed(0)
pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                   mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                   log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device=dev)
rng = np.random.default_rng(0)
lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
pc.create_from_points(rng.uniform(lim, -lim, (120_000, 3)), spatial_lr_scale=1.0)
pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
pc.training_setup(opt)
tr = Trainer(pc, cube, opt, pipe, mp_, seed=0)
losses, t0 = [], time.perf_counter()
for it in range(1, 401):
    out = tr.step(it)
    if it % 20 == 0:
        l = float(out.loss)
        losses.append(l)
        print(f"it {it:4d} mode {tr.controller.render_mode} anchors {pc._anchor.shape[0]} loss {l:.5f} active {int(out.active_gaussians)}", flush=True)
        assert np.isfinite(l)
torch.cuda.synchronize()
print(f"400 steps in {time.perf_counter() - t0:.1f} s; loss {losses[0]:.4f} -> {losses[-1]:.4f}; repeated steps {getattr(tr, 'repeated_steps', 0)}")
for n, p in pc.named_parameters():
    assert torch.isfinite(p).all(), n
print("all parameters finite")
