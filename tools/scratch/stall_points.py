"""Where is the fitting step host-bound?  A busy-wait of STALL_US on the host at one point of the step; if the free-running step
time does not move, the GPU had that much queued work at that point (alternating 25-step segments, one process)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
import gsvc_amd.train as T
STALL_US = float(os.environ.get("STALL_US", "300"))
point = [None]
def stall(name):
    if point[0] == name:
        t = time.perf_counter()
        while (time.perf_counter() - t) * 1e6 < STALL_US:
            pass
def wrap(mod, fn, name, after=False):
    orig = getattr(mod, fn)
    def w(*a, **k):
        if not after: stall(name)
        r = orig(*a, **k)
        if after: stall(name)
        return r
    setattr(mod, fn, w)
wrap(T, "render_many", "before_render")
wrap(T, "render_many", "after_render", after=True)
wrap(T, "ssim_l1_pair", "before_ssim")
wrap(T, "render_regs", "before_regs")
wrap(T, "calc_optical_loss", "before_optical")
wrap(T, "hash_grid_bits", "before_hashbits")
_bw = torch.Tensor.backward
def bw(self, *a, **k):
    stall("before_backward")
    return _bw(self, *a, **k)
torch.Tensor.backward = bw
wrap(T, "resolve_deferred", "before_resolve")
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
tr = T.Trainer(pc, cube, opt, pipe, mp_, seed=0)
it = 0
for _ in range(150):
    it += 1; tr.step(it)
points = [None, "before_render", "after_render", "before_ssim", "before_regs", "before_optical", "before_hashbits", "before_backward", "before_resolve"]
res = {p: [] for p in points}
for rep in range(3):
    for p in points:
        point[0] = p
        for _ in range(3):
            it += 1; tr.step(it)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            it += 1; tr.step(it)
        torch.cuda.synchronize()
        res[p].append(1e3 * (time.perf_counter() - t0) / 20)
base = np.mean(res[None])
for p in points:
    print(f"stall {STALL_US:.0f} us at {str(p):18s}: {np.mean(res[p]):.3f} ms/step ({np.mean(res[p]) - base:+.3f})  runs {' '.join('%.2f' % x for x in res[p])}")
