"""Where does a step's time go LATE in a fit?  The synthetic 1080p video is fitted through a scaled schedule up to the middle of the
entropy-constrained phase (default 6 000 of 10 000 steps), then 40 steps are run with the library's per-kernel events on:
instances / active Gaussians per render and the kernels' time per step, beside the same numbers 200 steps into the fit.
    python tools/ab/late_stage_profile.py [total_steps] [stop_at]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsvc_amd import _lib
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
from gsvc_amd.train import Trainer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
STOP = int(sys.argv[2]) if len(sys.argv) > 2 else int(0.6 * N)
dev = torch.device("cuda", 0)
mp_, opt, pipe = cfg_20240919()
cube = SyntheticFrameCube(1080, 1920, 64, seed=1234, device=dev).materialize()
mp_.threshold = 8.0 / cube.scale
s = N / 40_000.0
opt.iterations = N
opt.full_precision_training_total, opt.quantized_training_total = int(10_000 * s), int(5_000 * s)
opt.entropy_constrained_train_total = int(20_000 * s)
opt.ste_entropy_constrained_train_total = N - int(35_000 * s)
opt.start_stat, opt.update_from, opt.update_until = int(500 * s), int(1_500 * s), int(25_000 * s)
opt.update_interval, opt.pause_densification = max(20, int(100 * s)), int(1_000 * s)
for name in dir(opt):
    if name.endswith("_max_steps"):
        setattr(opt, name, N)
torch.manual_seed(0); np.random.seed(0)
pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                   mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                   log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device=dev)
lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
pc.create_from_points(np.random.default_rng(0).uniform(lim, -lim, (100_000, 3)), spatial_lr_scale=1.0)
pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
pc.training_setup(opt)
tr = Trainer(pc, cube, opt, pipe, mp_, seed=0)


def probe(tag, it0):
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    events = not os.environ.get("LATE_NO_EVENTS")      # (the per-kernel events serialise the step's streams: off for a kernel trace)
    _lib.profile_enable(events)
    inst = act = sub = 0
    n = 40
    for k in range(n):
        out = tr.step(it0 + k)
        inst += sum(r.num_rendered for r in out.renders) / 4
        act += float(out.active_gaussians) / 4
        sub += sum(int(r.radii.shape[0]) for r in out.renders) / 4
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    prof = _lib.profile_collect()
    _lib.profile_enable(False)
    per = sorted(((1e3 * ms / n, k, c / n) for k, (c, ms) in prof.items()), reverse=True)
    print(f"{tag}: iteration {it0}, mode {tr.controller.render_mode.name}, anchors {int(pc._anchor.shape[0])}: {wall:.2f} ms/step (events on), per render: "
          f"submitted {sub / n:.0f}, active {act / n:.0f}, instances {inst / n:.0f} ({inst / max(act, 1):.1f} tiles per active Gaussian)")
    print("   " + "; ".join(f"{k} {us:.0f} us x{c:.1f}" for us, k, c in per[:14]), flush=True)
    # how many (tile, Gaussian) instances of the 3-sigma rectangles can reach alpha >= 1/255 in their tile at all?  (the alpha box knows
    # the opacity; an instance whose box misses its tile is binned, sorted and walked, and owns a zero row of the backward's buffer)
    st = out.renders[0].raster_state
    off, pl = st.tile_lists()
    P_, W_ = st.P, st.cs.image_width
    gx = (W_ + 15) // 16
    tile = torch.repeat_interleave(torch.arange(off.numel() - 1, device=dev), (off[1:] - off[:-1]).long())
    tx0, ty0 = (tile % gx) * 16, (tile // gx) * 16
    g = st.geom[:64 * P_].view(torch.int32).view(P_, 16)
    bx, by = g[:, 9][pl.long()], g[:, 10][pl.long()]
    lo = lambda v: ((v & 0xffff) ^ 0x8000) - 0x8000
    hi = lambda v: v >> 16
    miss = (lo(bx) > tx0 + 15) | (hi(bx) < tx0) | (lo(by) > ty0 + 15) | (hi(by) < ty0)
    op = g[:, 5].view(torch.float32)[pl.long()]
    print(f"   instances of render 0: {pl.numel()}, alpha box misses the tile: {float(miss.float().mean()):.3f}; opacity of the listed Gaussians: "
          f"median {float(op.median()):.3f}, share below 0.05: {float((op < 0.05).float().mean()):.3f}", flush=True)
    # the compositing backward's own counters (its diagnostic instantiation): (entry, quadrant) replays, lanes of those replays that
    # held a contributing pixel, entries replayed
    pr = [0, 0, 0, 0]
    _lib.profile_enable(2)
    try:
        for k in range(2):
            o2 = tr.step(it0 + n + k)
            torch.cuda.synchronize()
            for r in o2.renders:
                c = r.raster_state.binning[64:88].view(torch.int64).tolist()
                for i in range(3):
                    pr[i] += c[i]
                pr[3] += r.raster_state.listed_instances()
    finally:
        _lib.profile_enable(False)
    print(f"   compositing backward: {pr[2] / max(pr[3], 1):.3f} of the listed instances are replayed, {pr[0] / max(pr[2], 1):.2f} quadrant replays per replayed "
          f"entry, {pr[1] / max(64 * pr[0], 1):.3f} of the replayed lanes hold a contributing pixel", flush=True)
    return it0 + n + 2


it = 1
while it <= 200:
    tr.step(it); it += 1
it = probe("early", it)
while it <= STOP:
    tr.step(it); it += 1
it = probe("late", it)
