"""Coarse wall-clock breakdown of one fitting step (synchronising between sections; diagnostic only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
from gsvc_amd.train import Trainer
from gsvc_amd.generate import generate_neural_gaussians
from gsvc_amd.ortho_gaussian_renderer import prefilter_voxel, render
from gsvc_amd import loss_utils as LU

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
    tr.step(i + 1)

def timed(name, fn, n=3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    print(f"{name:34s} {1e3 * (time.perf_counter() - t0) / n:8.2f} ms")
    return r

fr = cube[30]
bg = torch.zeros(3)
mode = tr.controller.render_mode
vis = timed("prefilter_voxel", lambda: prefilter_voxel(fr, pc, pipe, bg))
timed("get_anchor", lambda: pc.get_anchor)
timed("get_mask", lambda: pc.get_mask)
gss = timed("generate_neural_gaussians", lambda: generate_neural_gaussians(fr, pc, vis, mode))
timed("calc_entropy_context", lambda: pc.calc_entropy_context(pc.get_anchor[vis]))
res = timed("render (fwd)", lambda: render(fr, pc, pipe, bg, retain_grad=True, mode=mode))
def fb():
    r = render(fr, pc, pipe, bg, retain_grad=True, mode=mode)
    (r.rendered_image.mean() + r.bit_per_param).backward()
    pc.optimizer.zero_grad(set_to_none=True)
    return r
timed("render fwd+bwd", fb)
img = res.rendered_image.detach()
gt = fr.image.to(dev).permute(0, 2, 1)
timed("ssim+l1", lambda: LU.ssim_func(img, gt) + LU.l1_loss_func(img, gt))
fr2 = cube[31]
res2 = render(fr2, pc, pipe, bg, retain_grad=True, mode=mode)
flow = cube.get_optical_flow(30)
timed("get_optical_flow", lambda: cube.get_optical_flow(30))
timed("optical loss (one dir)", lambda: LU.calc_optical_loss_one_frame(res, res2, flow, cube.x_min, cube.y_min, cube.scale, cube.width, cube.height, 10))
res.rendered_image.sum().backward()
timed("training_statis", lambda: pc.training_statis(res))
timed("dataset[idx]", lambda: cube[32])
def opt_step():
    for p in pc.parameters():
        if p.requires_grad and p.grad is None:
            p.grad = torch.zeros_like(p)
    pc.optimizer.step()
timed("optimizer.step", opt_step)
timed("full step", lambda: tr.step(10))
