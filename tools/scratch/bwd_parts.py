"""Where k_blend_bwd_tile's time goes: the kernel's timing experiments (GSVC_BWD_DEBUG bits, diagnostic instantiation) on a
fitting-like scene.  usage: GSVC_BWD_DEBUG=<bits|256> python tools/scratch/bwd_parts.py   (256 = diagnostic instantiation, nothing off)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gsvc_amd import _lib, rasterizer, synthetic
dev = torch.device("cuda")
for name, P, sigma in (("cfg2", 200_000, (0.5, 4.0)), ("fitting-like", 180_000, (2.0, 12.0))):
    sc = synthetic.raster_scene(P, H=1080, W=1920, T=600, seed=2026, window_frames=16, frame_id=300, sigma_px=sigma)
    s = sc["settings"]
    rs = rasterizer.GaussianRasterizationSettings(
        image_height=s["H"], image_width=s["W"], x_min=s["x_min"], y_min=s["y_min"], scale=s["scale"], threshold=s["threshold"],
        bg=torch.zeros(3), scale_modifier=1.0, viewmatrix=torch.tensor(s["viewmatrix"]), sh_degree=0,
        campos=torch.tensor([0.0, 0.0, s["z_cam"]]), prefiltered=False, debug=False)
    d = {k: torch.tensor(sc[k], device=dev, requires_grad=True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    r = rasterizer.GaussianRasterizer(raster_settings=rs)
    g = None
    for it in range(14):
        if it == 4:
            _lib.profile_enable(1); _lib.profile_collect()
        m2 = torch.zeros_like(d["means3D"], requires_grad=True)
        img, radii, n_inst = r(means3D=d["means3D"], means2D=m2, shs=None, colors_precomp=d["colors"], opacities=d["opacities"],
                               scales=d["scales"], rotations=d["rotations"], cov3D_precomp=None)
        if g is None:
            g = torch.randn_like(img)
        img.backward(g)
    torch.cuda.synchronize()
    pr = _lib.profile_collect(); _lib.profile_enable(0)
    n, ms = pr["k_blend_bwd"]
    print(f"dbg={os.environ.get('GSVC_BWD_DEBUG', '0'):>4s} {name:13s} {n_inst} instances  k_blend_bwd {1e3 * ms / n:7.1f} us")
