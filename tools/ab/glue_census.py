"""WHERE do the PyTorch kernels of a fitting step come from?  One production step (plan built ahead, two raster streams) under
torch.profiler with shapes: every aten operator that launched a kernel is listed with its input shapes and the chain of its callers
(the autograd node or the Python-side operator it sits under).  Usage: python tools/ab/glue_census.py [anchors] [phase]
(phase: FULL | QUANT | ENTROPY | STE, default ENTROPY = the headline's)."""
import os
import sys
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gsvc_amd  # noqa: E402,F401
from gsvc_amd.arguments import cfg_20240919  # noqa: E402
from gsvc_amd.frame import SyntheticFrameCube  # noqa: E402
from gsvc_amd.model import GaussianModel  # noqa: E402
from gsvc_amd.train import Trainer  # noqa: E402


def main():
    A = int(sys.argv[1]) if len(sys.argv) > 1 else 245_000
    phase = sys.argv[2] if len(sys.argv) > 2 else "ENTROPY"
    dev = torch.device("cuda", 0)
    mp_, opt, pipe = cfg_20240919()
    cube = SyntheticFrameCube(1080, 1920, 64, seed=1234, device=dev).materialize()
    mp_.threshold = 8.0 / cube.scale
    opt.start_stat, opt.update_until, opt.pause_densification, opt.update_from = 10 ** 9, 10 ** 9, 0, 10 ** 9
    B = 10 ** 9
    torch.manual_seed(0)
    np.random.seed(0)
    pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                       mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                       log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device=dev)
    lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
    pc.create_from_points(np.random.default_rng(0).uniform(lim, -lim, (A, 3)), spatial_lr_scale=1.0)
    pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
    PH = {"FULL": (B, 0, 0, 0), "QUANT": (0, B, 0, 0), "ENTROPY": (0, 0, B, 0), "STE": (0, 0, 0, B)}
    (opt.full_precision_training_total, opt.quantized_training_total, opt.entropy_constrained_train_total,
     opt.ste_entropy_constrained_train_total) = PH[phase]
    pc.training_setup(opt)
    tr = Trainer(pc, cube, opt, pipe, mp_, seed=0)
    for it in range(1, 61):
        tr.step(it)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for it in range(61, 64):
            tr.step(it)
        torch.cuda.synchronize()
    rows = defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if not e.name.startswith("aten::"):
            continue
        own = sum(k.duration for k in e.kernels)
        if not e.kernels:
            continue
        chain, p = [], e.cpu_parent
        while p is not None and len(chain) < 4:
            if not p.name.startswith("aten::"):
                chain.append(p.name[:60])
            p = p.cpu_parent
        shapes = str([s for s in (e.input_shapes or []) if s])[:70]
        key = (e.name, shapes, " < ".join(chain) or "(top level)", ",".join(sorted({k.name[:40] for k in e.kernels})))
        rows[key][0] += 1
        rows[key][1] += own
    print(f"phase {phase}, {A} anchors, 3 steps profiled; per step: launches / device us / operator / shapes / under / kernel")
    tot_n = tot_us = 0
    for (name, shapes, chain, kern), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f"{n / 3:5.1f} {us / 3:8.1f}  {name:22s} {shapes:70s} {chain[:110]:110s} {kern}")
        tot_n += n
        tot_us += us
    print(f"total {tot_n / 3:.1f} launches, {tot_us / 3:.1f} us per step")
    tr.close()


if __name__ == "__main__":
    main()
