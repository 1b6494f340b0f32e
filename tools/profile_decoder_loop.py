"""Where the decoder loop's time goes (reference utils/report_utils.py:293-319: per output frame the visibility
test, the anchor -> Gaussian generation and the two-view frame).  On the fitted headline model (245 k anchors, 1080p, 16-frame slab,
200 fitting steps): frames per second of `render_frames` (batched, pipelined) and of a `render_pair` loop; per frame the HOST time to
issue it (loop time before the final synchronise) against the wall time, and every kernel's time per frame from the library's own
launch events (gsvc_profile_enable: events on the launch stream, so co-running kernels overlap in the sum).

    python tools/profile_decoder_loop.py [anchors=245000] [fit_steps=200] [frames=48]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsvc_amd import _lib  # noqa: E402
from gsvc_amd.arguments import cfg_20240919  # noqa: E402
from gsvc_amd.frame import SyntheticFrameCube  # noqa: E402
from gsvc_amd.generate import GenerateMode  # noqa: E402
from gsvc_amd.model import GaussianModel  # noqa: E402
from gsvc_amd.ortho_gaussian_renderer import render_frames, render_pair  # noqa: E402
from gsvc_amd.train import Trainer  # noqa: E402


def main():
    anchors = int(sys.argv[1]) if len(sys.argv) > 1 else 245_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    n_frames = int(sys.argv[3]) if len(sys.argv) > 3 else 48
    dev = torch.device("cuda", 0)
    mp_, opt, pipe = cfg_20240919()
    cube = SyntheticFrameCube(1080, 1920, 64, seed=1234, device=dev).materialize()
    mp_.threshold = 8.0 / cube.scale
    opt.full_precision_training_total, opt.quantized_training_total = 0, 0
    opt.entropy_constrained_train_total = 10 ** 9
    opt.start_stat, opt.update_until, opt.pause_densification = 0, 10 ** 9, 0
    torch.manual_seed(0)
    np.random.seed(0)
    pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                       mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                       log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device=dev)
    lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
    pc.create_from_points(np.random.default_rng(0).uniform(lim, -lim, (anchors, 3)), spatial_lr_scale=1.0)
    pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
    pc.training_setup(opt)
    tr = Trainer(pc, cube, opt, pipe, mp_, seed=0)
    for it in range(1, steps + 1):
        tr.step(it)
    torch.cuda.synchronize()
    bg = tr.background
    tr.close()
    frames = [cube.get_dummy_frame(i) for i in range(8, 8 + n_frames)]

    def loop_frames():
        for _ in render_frames(frames, pc, pipe, bg):
            pass

    def loop_pair():
        for fr in frames:
            render_pair(fr, pc, pipe, bg, mode=GenerateMode.DECODING_AS_IS)

    # the rasterizer's kernels ALONE on the chip, on one frame's Gaussians: single view against the two-view pass
    from gsvc_amd.generate import generate_neural_gaussians_many
    from gsvc_amd.ortho_gaussian_renderer.preprocess import prefilter_geometry, prefilter_voxels_many, raster_settings_for
    from gsvc_amd.rasterizer import raster_forward, settings_to_c
    with torch.no_grad():
        geometry = prefilter_geometry(pc)
        fr = frames[len(frames) // 2]
        vis = prefilter_voxels_many([fr], pc, pipe, bg, geometry=geometry)
        gss = generate_neural_gaussians_many([fr], pc, vis, GenerateMode.DECODING_AS_IS, dense=True, anchors=geometry[0])[0]
        args = tuple(t.contiguous() for t in (gss.xyz, gss.color, gss.opacity, gss.scaling, gss.rot))
        cs = settings_to_c(raster_settings_for(fr, pc, pipe, bg, 1.0))
        for pair in (False, True):
            for _ in range(3):
                _, radii, st = raster_forward(cs, *args, pair=pair)
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            for _ in range(10):
                raster_forward(cs, *args, pair=pair, sync=False)
            torch.cuda.synchronize()
            prof = _lib.profile_collect()
            _lib.profile_enable(False)
            print(f"rasterizer alone, {'two-view pass' if pair else 'single view'}: {args[0].shape[0]} Gaussians, {int((radii > 0).sum())} active, "
                  f"{st.listed_instances()} listed instances: " + "  ".join(f"{k} {1e3 * ms / max(c, 1):.1f} us" for k, (c, ms) in sorted(prof.items())) +
                  f"  (sum {sum(1e3 * ms / max(c, 1) for c, ms in prof.values()):.1f} us)", flush=True)

    for name, fn in (("render_frames (batches of 8, pipelined)", loop_frames), ("render_pair loop (one frame per call)", loop_pair)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        _lib.profile_enable(True)
        fn()
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        _lib.profile_enable(False)
        ksum = sum(ms for _, ms in prof.values())
        print(f"\n{name}: {n_frames / t_all:.0f} fps = {1e3 * t_all / n_frames:.3f} ms per frame; host issue {1e3 * t_host / n_frames:.3f} ms per frame "
              f"({'host-bound' if t_host > 0.9 * t_all else 'GPU-bound'}); library kernels {1e3 * ksum / n_frames:.1f} us per frame (sum over streams)")
        for k, (c, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
            print(f"    {k:28s} {c / n_frames:6.2f} launches/frame  {1e3 * ms / n_frames:8.1f} us/frame  ({1e3 * ms / max(c, 1):7.1f} us each)")


if __name__ == "__main__":
    main()
