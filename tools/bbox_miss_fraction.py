"""Diagnostic: fraction of (tile, Gaussian) instances whose alpha bounding box does not touch the tile at all."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gsvc_amd import rasterizer, synthetic
for P, sig in ((200_000, (0.5, 4.0)), (180_000, (2.0, 14.0))):
    sc = synthetic.raster_scene(P, seed=2026, sigma_px=sig)
    s = sc["settings"]
    rs = rasterizer.GaussianRasterizationSettings(image_height=s["H"], image_width=s["W"], x_min=s["x_min"], y_min=s["y_min"], scale=s["scale"],
        threshold=s["threshold"], bg=torch.zeros(3), scale_modifier=1.0, viewmatrix=torch.tensor(s["viewmatrix"]))
    cs = rasterizer.settings_to_c(rs)
    d = {k: torch.tensor(sc[k], device="cuda") for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    _, radii, st = rasterizer.raster_forward(cs, d["means3D"], d["colors"], d["opacities"].view(-1).contiguous(), d["scales"], d["rotations"])
    off, pl = st.tile_lists()
    n = pl.numel()
    gx = (s["W"] + 15) // 16
    tile = torch.repeat_interleave(torch.arange(off.numel() - 1, device="cuda"), (off[1:] - off[:-1]).long())
    tx0, ty0 = (tile % gx) * 16, (tile // gx) * 16
    g = st.geom[:64 * P].view(torch.int32).view(P, 16)
    bx, by = g[:, 9][pl.long()], g[:, 10][pl.long()]
    lo = lambda v: ((v & 0xffff) ^ 0x8000) - 0x8000
    hi = lambda v: v >> 16
    miss = (lo(bx) > tx0 + 15) | (hi(bx) < tx0) | (lo(by) > ty0 + 15) | (hi(by) < ty0)
    print(P, sig, "instances", n, "bbox misses the tile:", round(float(miss.float().mean()), 4))
