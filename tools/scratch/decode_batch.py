import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
from gsvc_amd.ortho_gaussian_renderer import render_frames
dev = torch.device("cuda")
mp_, opt, pipe = cfg_20240919()
cube = SyntheticFrameCube(1080, 1920, 128, device=dev).materialize()
mp_.threshold = 8.0 / cube.scale
pc = GaussianModel(mp_, 50, 10, 0.001, 3, 16, 4, False, n_features_per_level=8, log2_hashmap_size=13, log2_hashmap_size_2D=15, device=dev)
rng = np.random.default_rng(0)
lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
pc.create_from_points(rng.uniform(lim, -lim, (245000, 3)), 1.0)
pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
bg = torch.zeros(3)
frames = [cube.get_dummy_frame(i) for i in range(8, 104)]
for b in (4, 8, 12, 16, 24):
    for _ in render_frames(frames[:2 * b], pc, pipe, bg, batch=b):
        pass
    torch.cuda.synchronize()
    best = 0
    for rep in range(3):
        t0 = time.perf_counter()
        n = sum(1 for _ in render_frames(frames, pc, pipe, bg, batch=b))
        torch.cuda.synchronize()
        best = max(best, n / (time.perf_counter() - t0))
    print(f"batch {b}: {best:.0f} fps", flush=True)
