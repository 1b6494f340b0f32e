"""torch.profiler view of the decoder loop (render_frames): GPU time by op (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import ProfilerActivity, profile
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
from gsvc_amd.ortho_gaussian_renderer import render_frames
dev = torch.device("cuda")
mp_, opt, pipe = cfg_20240919()
cube = SyntheticFrameCube(1080, 1920, 64, device=dev).materialize()
mp_.threshold = 8.0 / cube.scale
pc = GaussianModel(mp_, 50, 10, 0.001, 3, 16, 4, False, n_features_per_level=8, log2_hashmap_size=13, log2_hashmap_size_2D=15, device=dev)
rng = np.random.default_rng(0)
lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
pc.create_from_points(rng.uniform(lim, -lim, (220000, 3)), 1.0)
pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
bg = torch.zeros(3)
frames = [cube.get_dummy_frame(i) for i in range(8, 56)]
for _ in render_frames(frames[:16], pc, pipe, bg):
    pass
torch.cuda.synchronize()
t0 = time.perf_counter()
n = sum(1 for _ in render_frames(frames, pc, pipe, bg))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{n} frames in {dt * 1e3:.1f} ms = {n / dt:.0f} fps")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in render_frames(frames[:16], pc, pipe, bg):
        pass
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=32, max_name_column_width=60))
