"""Diagnostic: ATen ops dispatched per Python source line during one fitting step (TorchDispatchMode census)."""
import os, sys, traceback
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
from gsvc_amd.train import Trainer
dev = torch.device("cuda")
mp_, opt, pipe = cfg_20240919()
cube = SyntheticFrameCube(1080, 1920, 64, device=dev).materialize()
mp_.threshold = 8.0 / cube.scale
opt.full_precision_training_total = opt.quantized_training_total = 0
opt.entropy_constrained_train_total = 10 ** 9
opt.start_stat, opt.update_until, opt.pause_densification = 0, 10 ** 9, 0
pc = GaussianModel(mp_, 50, 10, 0.001, 3, 16, 4, False, n_features_per_level=8, log2_hashmap_size=13, log2_hashmap_size_2D=15, device=dev)
rng = np.random.default_rng(0)
lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
pc.create_from_points(rng.uniform(lim, -lim, (220000, 3)), 1.0)
pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
pc.training_setup(opt)
tr = Trainer(pc, cube, opt, pipe, mp_)
for i in range(3):
    tr.step(i + 1, frame_idx=30)
cnt = Counter()
from torch.utils._python_dispatch import TorchDispatchMode


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        fr = [f for f in traceback.extract_stack() if "/gsvc_amd/" in f.filename][-1:]
        where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr)
        cnt[(where, name)] += 1
        return func(*args, **(kwargs or {}))


with Census():
    tr.step(10, frame_idx=30)
by_line = Counter()
for (w, n), c in cnt.items():
    by_line[w] += c
print("total dispatched ops in forward+python side:", sum(cnt.values()))
for w, c in by_line.most_common(40):
    ops = sorted(((n, k) for (ww, n), k in cnt.items() if ww == w), key=lambda t: -t[1])[:6]
    print(f"{c:5d} {w:28s} " + ", ".join(f"{n}x{k}" for n, k in ops))
