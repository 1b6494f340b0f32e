"""Host profile of the render_pair loop (one frame per call) on the fitted headline model: cProfile over 48 frames, top functions by
own time.  Run it on a copy of the tree WITHOUT the compiled host modules (python setup_host.py clean_host) to see inside them.
    python tools/ab/pair_host_profile.py [anchors=245000] [fit_steps=100]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsvc_amd.arguments import cfg_20240919  # noqa: E402
from gsvc_amd.frame import SyntheticFrameCube  # noqa: E402
from gsvc_amd.generate import GenerateMode  # noqa: E402
from gsvc_amd.model import GaussianModel  # noqa: E402
from gsvc_amd.ortho_gaussian_renderer import render_pair  # noqa: E402
from gsvc_amd.train import Trainer  # noqa: E402


def main():
    anchors = int(sys.argv[1]) if len(sys.argv) > 1 else 245_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
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
    frames = [cube.get_dummy_frame(i) for i in range(8, 56)]

    def loop():
        for fr in frames:
            render_pair(fr, pc, pipe, bg, mode=GenerateMode.DECODING_AS_IS)

    with torch.no_grad():
        loop()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"render_pair loop: {len(frames) / t_all:.0f} fps, host issue {t_issue / len(frames) * 1e3:.3f} ms per frame, wall "
              f"{t_all / len(frames) * 1e3:.3f} ms per frame", flush=True)
        pr = cProfile.Profile()
        pr.enable()
        loop()
        pr.disable()
        torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats(22)


if __name__ == "__main__":
    main()
