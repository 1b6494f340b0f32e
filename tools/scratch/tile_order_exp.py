"""Experiment: k_blend_bwd_tile with the workgroups taking the tiles in descending order of list length (longest first) against
the row-major order.  Needs the library built with -DGSVC_EXP_TILE_ORDER (GSVC_LIB_PATH)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsvc_amd import _lib, rasterizer, synthetic
dev = torch.device("cuda:0")
L = _lib.lib()
L.gsvc_exp_set_tile_order.argtypes = [C.c_void_p]
for P, sig in ((200_000, (0.5, 4.0)), (520_000, (0.5, 4.0)), (180_000, (2.0, 12.0))):
    sc = synthetic.raster_scene(P, H=1080, W=1920, T=600, seed=2026, window_frames=16, frame_id=300, sigma_px=sig)
    s = sc["settings"]
    rs = rasterizer.GaussianRasterizationSettings(image_height=1080, image_width=1920, x_min=s["x_min"], y_min=s["y_min"], scale=s["scale"],
        threshold=s["threshold"], bg=torch.zeros(3), scale_modifier=1.0, viewmatrix=torch.tensor(s["viewmatrix"]), sh_degree=0,
        campos=torch.tensor([0.0, 0.0, s["z_cam"]]), prefiltered=False, debug=False)
    cs = rasterizer.settings_to_c(rs)
    d = {k: torch.tensor(sc[k], device=dev) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    d["opacities"] = d["opacities"].view(-1).contiguous()
    dL = torch.randn(3, 1080, 1920, device=dev)
    _, radii, st = rasterizer.raster_forward(cs, d["means3D"], d["colors"], d["opacities"], d["scales"], d["rotations"])
    n_inst = st.counters()[0]
    off, _ = st.tile_lists()
    lens = torch.diff(off.to(torch.int64))
    order = torch.argsort(lens, descending=True).to(torch.int32).contiguous()
    grads = [torch.empty(P, 3, device=dev), torch.empty(P, 3, device=dev), torch.empty(P, 3, device=dev), torch.empty(P, device=dev),
             torch.empty(P, 3, device=dev), torch.empty(P, 4, device=dev)]
    scratch = torch.empty(rasterizer.backward_scratch_floats(P, st.max_instances), device=dev)
    def bwd():
        _lib.check(L.gsvc_raster_backward(C.byref(cs), P, st.max_instances, _lib.ptr(d["means3D"]), _lib.ptr(d["colors"]), _lib.ptr(d["opacities"]),
                                          _lib.ptr(d["scales"]), _lib.ptr(d["rotations"]), _lib.ptr(radii), _lib.ptr(st.geom), _lib.ptr(st.binning),
                                          _lib.ptr(st.image_state), _lib.ptr(dL), *[_lib.ptr(g) for g in grads], _lib.ptr(scratch),
                                          _lib.current_stream(dev)), "bwd")
    res = {}
    for name, ptr in (("row-major", None), ("longest first", order.data_ptr()), ("row-major again", None)):
        assert L.gsvc_exp_set_tile_order(ptr) == 0
        for _ in range(5): bwd()
        torch.cuda.synchronize()
        ref = [g.clone() for g in grads] if name == "row-major" else ref
        _lib.profile_enable(True)
        for _ in range(30): bwd()
        torch.cuda.synchronize()
        pr = _lib.profile_collect(); _lib.profile_enable(False)
        res[name] = 1e3 * pr["k_blend_bwd"][1] / pr["k_blend_bwd"][0]
        assert all(torch.equal(a, b) for a, b in zip(ref, grads))
    print(f"P={P} sigma={sig} instances={n_inst} mean list {float(lens.float().mean()):.0f} max {int(lens.max())}: " +
          ", ".join(f"{k} {v:.1f} us" for k, v in res.items()))
