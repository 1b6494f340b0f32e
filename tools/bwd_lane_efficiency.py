"""Lane efficiency of k_blend_bwd_tile's per-pixel replay (VERDICT round 2, item 3 i): of the 64 lanes that execute a replay of
one list entry over one 8x8 quadrant, how many hold a pixel the entry actually contributes to (alpha >= 1/255, not behind the
pixel's last contributor)?  GSVC_BWD_DEBUG=128 makes the kernel count, per launch, the replays, the valid lanes and the replayed
entries into spare words of the counters block (csrc/raster_bwd.hip); no SQ counters needed.

    python tools/bwd_lane_efficiency.py            # BASELINE configs[1] raster set (sigma 0.5 .. 4 px) and a fitting-like set (2 .. 12 px)
"""
import os
import sys

os.environ["GSVC_BWD_DEBUG"] = "128"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gsvc_amd import rasterizer, synthetic


def probe(name, P, sigma):
    dev = torch.device("cuda")
    sc = synthetic.raster_scene(P, H=1080, W=1920, T=600, seed=2026, window_frames=16, frame_id=300, sigma_px=sigma)
    s = sc["settings"]
    rs = rasterizer.GaussianRasterizationSettings(
        image_height=s["H"], image_width=s["W"], x_min=s["x_min"], y_min=s["y_min"], scale=s["scale"], threshold=s["threshold"],
        bg=torch.zeros(3), scale_modifier=1.0, viewmatrix=torch.tensor(s["viewmatrix"]), sh_degree=0,
        campos=torch.tensor([0.0, 0.0, s["z_cam"]]), prefiltered=False, debug=False)
    d = {k: torch.tensor(sc[k], device=dev, requires_grad=True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    m2 = torch.zeros_like(d["means3D"], requires_grad=True)
    r = rasterizer.GaussianRasterizer(raster_settings=rs)
    img, radii, n_inst = r(means3D=d["means3D"], means2D=m2, shs=None, colors_precomp=d["colors"], opacities=d["opacities"],
                           scales=d["scales"], rotations=d["rotations"], cov3D_precomp=None)
    st = r.last_state
    st.binning[64:88].zero_()
    img.backward(torch.randn_like(img))
    torch.cuda.synchronize()
    c = st.binning[64:88].view(torch.int64).cpu().tolist()
    replays, lanes, entries = c
    print(f"{name}: {P} Gaussians sigma {sigma} px, {n_inst} instances: {entries} replayed entries, {replays} (entry, quadrant) replays "
          f"= {replays / max(entries, 1):.2f} per entry, valid lanes {lanes / max(replays, 1):.1f} of 64 = {100.0 * lanes / max(64 * replays, 1):.1f} %")


if __name__ == "__main__":
    probe("configs[1] raster set", 200_000, (0.5, 4.0))
    probe("fitting-like footprints", 180_000, (2.0, 12.0))
