"""Do the two opposite views of a frame see the same anchors?  (visible masks of the step plan)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
from gsvc_amd.train import Trainer
from gsvc_amd.generate import GenerateMode
from gsvc_amd.ortho_gaussian_renderer.renderer import plan_views
dev = torch.device("cuda:0")
mp_, opt, pipe = cfg_20240919()
cube = SyntheticFrameCube(1080, 1920, 64, seed=1234, device=dev).materialize()
mp_.threshold = 8.0 / cube.scale
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
for idx in (5, 30, 50):
    plan = plan_views(tr._views(idx), pc, pipe, tr.background, GenerateMode.TRAINING_FULL_PRECISION)
    m = plan.visible_masks
    plan.resolve()
    print(idx, "pairs flag", plan._pairs, "equal", plan.pairs_equal, "counts", [int(x.sum()) for x in m],
          "diff01", int((m[0] != m[1]).sum()), "diff23", int((m[2] != m[3]).sum()))
    d = (m[0] != m[1]).nonzero().squeeze(1)[:5]
    if d.numel():
        a = pc.get_anchor[d]
        print("  anchors", a.tolist(), "scales", pc.get_scaling[d, :3].tolist())
