"""Decoder render loop (render_frames: generation + two-view pair pass) on a FITTED model, tight against loose binning, one process:
a scaled 3 000-step fit of the synthetic 1080p video, then 3 x 48 frames per setting with the library's per-kernel events.
    python tools/ab/decode_fps_ab.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsvc_amd import _lib, switches
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.generate import GenerateMode
from gsvc_amd.model import GaussianModel
from gsvc_amd.ortho_gaussian_renderer import render_frames
from gsvc_amd.train import Trainer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda", 0)
mp_, opt, pipe = cfg_20240919()
cube = SyntheticFrameCube(1080, 1920, 64, seed=1234, device=dev).materialize()
mp_.threshold = 8.0 / cube.scale
s = N / 40_000.0
opt.iterations = N
opt.full_precision_training_total, opt.quantized_training_total = int(10_000 * s), int(5_000 * s)
opt.entropy_constrained_train_total = int(20_000 * s)
opt.ste_entropy_constrained_train_total = N - int(35_000 * s)
opt.start_stat, opt.update_from, opt.update_until = 10 ** 9, 10 ** 9, 10 ** 9
for name in dir(opt):
    if name.endswith("_max_steps"):
        setattr(opt, name, N)
torch.manual_seed(0); np.random.seed(0)
pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                   mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                   log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device=dev)
lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
pc.create_from_points(np.random.default_rng(0).uniform(lim, -lim, (100_000, 3)), spatial_lr_scale=1.0)
pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
pc.training_setup(opt)
with Trainer(pc, cube, opt, pipe, mp_, seed=0) as tr:
    for it in range(1, N + 1):
        tr.step(it)
torch.cuda.synchronize()
frames = [cube[i] for i in range(8, 56)]
bg = torch.zeros(3)
for rep in range(2):
    for loose in (0, 1):
        if loose:
            os.environ["GSVC_RASTER_LOOSE_BINNING"] = "1"
        else:
            os.environ.pop("GSVC_RASTER_LOOSE_BINNING", None)
        switches.reload()
        with torch.no_grad():
            for _ in render_frames(frames[:16], pc, pipe, bg, mode=GenerateMode.TRAINING_FULL_PRECISION):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            for _ in range(3):
                for img in render_frames(frames, pc, pipe, bg, mode=GenerateMode.TRAINING_FULL_PRECISION):
                    n += 1
            torch.cuda.synchronize()
            fps = n / (time.perf_counter() - t0)
            _lib.profile_enable(True)
            for img in render_frames(frames, pc, pipe, bg, mode=GenerateMode.TRAINING_FULL_PRECISION):
                pass
            torch.cuda.synchronize()
            prof = _lib.profile_collect()
            _lib.profile_enable(False)
        per = sorted(((1e3 * ms / len(frames), k) for k, (c, ms) in prof.items()), reverse=True)[:7]
        print(f"{'loose' if loose else 'tight'} binning: {fps:7.1f} two-view frames/s; per frame: " + "; ".join(f"{k} {us:.0f} us" for us, k in per), flush=True)
