"""Which Gaussians differ most between the polynomial and the literal replay of k_blend_bwd_tile."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gsvc_amd import synthetic, _lib
from tests.test_raster_gpu import _rasterizer, _to_dev, _run_backward
smin, smax = float(sys.argv[1]), float(sys.argv[2])
sc = synthetic.raster_scene(30_000, seed=31, sigma_px=(smin, smax))
s = sc["settings"]
d = {k: v.requires_grad_(True) for k, v in _to_dev(sc).items()}
r = _rasterizer(s, bg=(0., 0., 0.))
dL = torch.randn(3, s["H"], s["W"], device="cuda", generator=torch.Generator("cuda").manual_seed(3))
_, m2 = _run_backward(r, d, dL)
poly = {k: v.grad.clone() for k, v in d.items()}
_lib.profile_enable(2)
_, m2 = _run_backward(r, d, dL)
_lib.profile_enable(0)
for k in poly:
    a, b = poly[k].double(), d[k].grad.double()
    err = (a - b).abs()
    print(k, "max abs", float(err.max()), "scale", float(b.abs().max()), "rel", float(err.max() / b.abs().max()))
a, b = poly["means3D"].double(), d["means3D"].grad.double()
err = (a - b).abs().amax(dim=1)
top = torch.topk(err, 8).indices
sig = torch.exp(d["scales"].detach()) if False else d["scales"].detach()
for i in top.tolist():
    print(i, "err", float(err[i]), "grad", b[i].tolist(), "scales", sig[i].tolist(), "op", float(d["opacities"][i]), "mean", d["means3D"][i].tolist())
