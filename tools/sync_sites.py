"""Diagnostic: list the Python call sites of one fitting step that force a host sync (boolean-mask indexing,
nonzero, item) — counted through a TorchFunctionMode."""
import os, sys, traceback
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.overrides import TorchFunctionMode
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
from gsvc_amd.train import Trainer
dev = torch.device("cuda")
mp_, opt, pipe = cfg_20240919()
cube = SyntheticFrameCube(1080, 1920, 64, device=dev)
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

def has_bool(idx):
    if isinstance(idx, torch.Tensor):
        return idx.dtype == torch.bool
    if isinstance(idx, (tuple, list)):
        return any(has_bool(i) for i in idx)
    return False

def has_long(idx):
    if isinstance(idx, torch.Tensor):
        return idx.dtype == torch.long and idx.dim() > 0
    if isinstance(idx, (tuple, list)):
        return any(has_long(i) for i in idx)
    return False


class Spy(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", str(func))
        hit = None
        if name in ("nonzero", "item", "masked_select", "tolist", "__bool__", "__int__", "__float__"):
            hit = name
        elif name in ("__getitem__", "__setitem__") and len(args) > 1 and has_bool(args[1]):
            hit = name + "[bool]"
        elif name in ("__getitem__", "__setitem__") and len(args) > 1 and has_long(args[1]):
            hit = name + "[long] grad=%s shape=%s" % (getattr(args[0], "requires_grad", None), tuple(args[0].shape))
        elif name in ("index_put", "index_put_", "index_add_", "index_add"):
            hit = name + " shape=%s" % (tuple(args[0].shape),)
        if hit:
            fr = [f for f in traceback.extract_stack() if "/gsvc_amd/" in f.filename][-2:]
            cnt[(hit, " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr)))] += 1
        return func(*args, **(kwargs or {}))

with Spy():
    tr.step(10, frame_idx=30)
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(v, k[0], k[1])
