#!/usr/bin/env python3
"""The Gaussians the REFERENCE generates, whole: one TRAINING_FULL_PRECISION ``render()`` of the production-
dimension model of make_golden_prod.py (reference ortho_gaussian_renderer/renderer.py:14-119, guassian.py:134-310 on PyTorch-CPU), of
which this fixture keeps EVERY row of what renderer.py:85-98 hands to the rasterizer — xyz, colour, opacity, scaling, rotation of the
~19 k compacted Gaussians — and what the rasterizer slot (oracle/raster_oracle.c) returned for them, forward view and opposite view:
radii, num_rendered, the per-tile ranges and the sorted point list, the image.  tests/test_prod_fixture_gpu.py feeds exactly these
rows to the HIP rasterizer: with identical inputs every integer must come out bit for bit (prod_render.npz compares through the MLPs,
whose CPU and GPU evaluations differ in the last bits upstream of the integer decisions).

Runs in the build container only (needs /root/reference).  Usage:  python tests/golden/make_golden_prod_gaussians.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import _ref_import, seeded  # noqa: E402
from tests.golden.make_golden_common import save  # noqa: E402

mode_ctx = _ref_import.install(rasterizer=True)

with mode_ctx:
    import arguments as A
    import scene.gaussian_model as GM
    import ortho_gaussian_renderer as OGR
    from ortho_gaussian_renderer import GenerateMode
    from frame_cube.frame import Frame

    sc = seeded.SCENE
    fn = seeded.frame_numbers(sc["H"], sc["W"], sc["T"], sc["frame"])
    mp = A.ModelParams()
    mp.threshold = sc["threshold"]
    P = seeded.PROD
    torch.manual_seed(0)
    ref = GM.GaussianModel(mp, feat_dim=P["feat_dim"], n_offsets=P["n_offsets"], voxel_size=0.001, update_depth=3,
                           update_init_factor=16, update_hierachy_factor=4, use_feat_bank=False,
                           n_features_per_level=P["n_features_per_level"], log2_hashmap_size=P["log2_hashmap_size"],
                           log2_hashmap_size_2D=P["log2_hashmap_size_2D"], resolutions_list=P["resolutions_list"],
                           resolutions_list_2D=P["resolutions_list_2D"])
    ref.update_anchor_bound(fn["x_min"], fn["y_min"], fn["z_min"])
    for name, t in seeded.anchors(sc["A"], fn, sc["threshold"], sc["seed"]).items():
        setattr(ref, name, nn.Parameter(t, requires_grad=name not in ("_rotation", "_opacity")))
    seeded.fill_parameters(ref, sc["seed"])

    def frame_for(view: str) -> Frame:
        vm, vms = (fn["view_matrix"], fn["view_matrix_s"]) if view == "f" else (fn["view_matrix_s"], fn["view_matrix"])
        return Frame(image_id=sc["frame"], plane="xy", image=None, x_min=fn["x_min"], y_min=fn["y_min"], z=fn["z"],
                     image_width=sc["W"], image_height=sc["H"], view_matrix=vm.clone(), view_matrix_s=vms.clone(),
                     scale=fn["scale"], cam_pos=fn["cam_pos"].clone())

    pipe = SimpleNamespace(debug=False, compute_cov3D_python=False)
    bg = torch.tensor([0.0, 0.0, 0.0])
    Ras = sys.modules["diff_gaussian_rasterization.cuda_ortho_gaussian_rasterizer"].GaussianRasterizer
    out = {}
    for view in ("f", "b"):
        frame = frame_for(view)
        with torch.no_grad(), seeded.SeededDraws(17) as draws:
            res = OGR.render(frame, ref, pipe, bg, retain_grad=False, mode=GenerateMode.TRAINING_FULL_PRECISION)
        assert draws.count == 0
        gs = res.generated_gaussians
        fwd = Ras.last["forward"]
        if view == "f":
            for nm in ("xyz", "color", "opacity", "scaling", "rot"):
                out["in::" + nm] = getattr(gs, nm).float()
        else:      # the generation does not depend on the view direction (guassian.py:225-273): the same rows
            assert all(torch.equal(getattr(gs, nm).float(), torch.from_numpy(out["in::" + nm]) if isinstance(out["in::" + nm], np.ndarray)
                                   else out["in::" + nm]) for nm in ("xyz", "color", "opacity", "scaling", "rot"))
        pre = view + "::"
        assert np.abs(res.radii.numpy()).max() < 2 ** 15
        out[pre + "radii"] = res.radii.numpy().astype(np.int16)
        out[pre + "num_rendered"] = np.int64(res.num_rendered)
        out[pre + "tile_ranges"] = fwd.tile_ranges.astype(np.int32)
        out[pre + "point_list"] = fwd.point_list[:fwd.num_rendered].astype(np.int32)
        out[pre + "image"] = res.rendered_image
        out[pre + "borderline"] = np.packbits(fwd.borderline.astype(bool))
        out[pre + "viewmatrix"] = frame.view_matrix.permute(1, 0).contiguous()       # what renderer.py:77 passes
        print(view, "Gaussians", gs.xyz.shape[0], "active", int(res.active_gaussains), "instances", res.num_rendered,
              "borderline px", int(fwd.borderline.sum()))
    out["meta::frame"] = np.array([sc["H"], sc["W"], fn["x_min"], fn["y_min"], fn["scale"], sc["threshold"], float(fn["cam_pos"][2])], np.float64)
    save("prod_gaussians", **out)
print("done")
