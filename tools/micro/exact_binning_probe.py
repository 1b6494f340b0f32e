"""Second formulation (ii) of VERDICT round 5 next-2, MEASURED: what would the compositing backward gain if the binning listed a
Gaussian only in the tiles its alpha ellipse really reaches (an exact ellipse-against-tile test at binning time, instead of the
alpha BOX the tight binning of round 5 uses)?

The binning change itself (a per-Gaussian tile mask through k_preprocess / k_scatter / the row index) is NOT built; its effect on
the backward is: on one render of the fitted headline model the tile lists are taken from the production forward, the instances
whose ellipse reaches none of their tile's four 8x8 quadrants (k_blend_bwd_tile's own test: bbox_hits_b && ellipse_hits_quad,
raster_common.h) are removed from the lists ON THE DEVICE (lists compacted per tile in order, n_contrib renumbered, the rows of the
removed instances pre-zeroed — what an exact binning would hand the backward), and gsvc_raster_backward is timed on both sets of
lists, alone on the chip (HIP events around every launch: gsvc_profile_enable).  The gradients of the two runs must agree bit for
bit (the removed instances can contribute nothing).  Run it under `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES
SQ_BUSY_CYCLES` for the instruction counts of the two forms.

    python tools/micro/exact_binning_probe.py [anchors=245000] [fit_steps=200]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsvc_amd import _lib  # noqa: E402
from gsvc_amd.arguments import cfg_20240919  # noqa: E402
from gsvc_amd.frame import SyntheticFrameCube  # noqa: E402
from gsvc_amd.model import GaussianModel  # noqa: E402
from gsvc_amd.train import Trainer  # noqa: E402


def fitted_render(anchors, steps, dev):
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
    out = None
    for it in range(1, steps + 1):
        out = tr.step(it)
    torch.cuda.synchronize()
    r = out.renders[0]
    gs = r.generated_gaussians
    args = [t.detach().contiguous().clone() for t in (gs.xyz, gs.color, gs.opacity, gs.scaling, gs.rot)]
    cs = r.raster_state.cs
    tr.close()
    return cs, args


def main():
    anchors = int(sys.argv[1]) if len(sys.argv) > 1 else 245_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda", 0)
    cs, (xyz, color, opacity, scaling, rot) = fitted_render(anchors, steps, dev)
    from gsvc_amd.rasterizer import raster_forward
    L = _lib.lib()
    image, radii, st = raster_forward(cs, xyz, color, opacity, scaling, rot)
    P, m = st.P, st.max_instances
    H, W = cs.image_height, cs.image_width
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    a, b = C.c_uint64(), C.c_uint64()
    _lib.check(L.gsvc_raster_binning_layout(C.byref(cs), P, m, C.byref(a), C.byref(b)), "layout")
    al = lambda v: (v + 255) // 256 * 256  # noqa: E731
    off_pl = b.value
    off_bb = off_pl + al(4 * m)
    off_gs = off_bb + al(8 * m)
    off, pl = st.tile_lists()
    n = int(off[T])
    blob = st.binning
    bbox = blob[off_bb:off_bb + 8 * n].view(torch.int32).view(n, 2)
    gslot = blob[off_gs:off_gs + 4 * n].view(torch.int32)
    print(f"render: {P} Gaussians submitted, {int((radii > 0).sum())} active, num_rendered {st.counters()[0]}, listed instances {n} "
          f"(tight binning: flags {cs.flags}), {T} tiles", flush=True)

    # ---- which instances can the backward ever replay?  its own tests, per quadrant (raster_common.h ellipse_hits_quad) ----
    geom = st.geom[:64 * P].view(torch.float32).view(P, 16)
    ids = pl.long()
    tile_of = torch.repeat_interleave(torch.arange(T, device=dev), (off[1:] - off[:-1]).long(), output_size=n)
    tx0 = (tile_of % gx).float() * 16.0
    ty0 = (tile_of // gx).float() * 16.0
    u, v, A, B = geom[ids, 0], geom[ids, 1], geom[ids, 2], geom[ids, 3]
    Cc, o = geom[ids, 4], geom[ids, 5]

    def s16(x):
        return ((x & 0xffff) ^ 0x8000) - 0x8000
    bx, by = bbox[:, 0], bbox[:, 1]
    bx0, bx1, by0, by1 = s16(bx), s16(bx >> 16), s16(by), s16(by >> 16)
    live = torch.zeros(n, dtype=torch.bool, device=dev)
    live_box = torch.zeros(n, dtype=torch.bool, device=dev)
    tau2 = 2.0 * (torch.log(255.0 * o) + 1e-3)
    mbc, mba = -B / Cc, -B / A
    for q in range(4):
        x0 = tx0 + 8.0 * (q & 1)
        y0 = ty0 + 8.0 * (q >> 1)
        hit_box = ~((bx0 > x0 + 7) | (bx1 < x0) | (by0 > y0 + 7) | (by1 < y0))
        lx, hx, ly, hy = x0 - u, x0 + 7.0 - u, y0 - v, y0 + 7.0 - v
        inside = (lx <= 0) & (hx >= 0) & (ly <= 0) & (hy >= 0)

        def qf(dx, dy):
            return A * dx * dx + 2.0 * B * dx * dy + Cc * dy * dy
        q1 = qf(lx, torch.minimum(torch.maximum(mbc * lx, ly), hy))
        q2 = qf(hx, torch.minimum(torch.maximum(mbc * hx, ly), hy))
        q3 = qf(torch.minimum(torch.maximum(mba * ly, lx), hx), ly)
        q4 = qf(torch.minimum(torch.maximum(mba * hy, lx), hx), hy)
        qmin = torch.minimum(torch.minimum(q1, q2), torch.minimum(q3, q4))
        hit_ell = inside | ~(qmin > tau2 * 1.0005 + 2e-3)
        live |= hit_box & hit_ell
        live_box |= hit_box
    n_live = int(live.sum())
    print(f"instances whose ellipse reaches a quadrant of their tile: {n_live} of {n} ({100.0 * n_live / n:.1f} %); the alpha BOX alone keeps "
          f"{int(live_box.sum())} ({100.0 * int(live_box.sum()) / n:.1f} %): an exact binning would drop {n - n_live} instances "
          f"({100.0 * (n - n_live) / n:.1f} %)", flush=True)

    # ---- formulation (i), its statistics: pixels an entry can reach (alpha >= 1/255: geometry only, no occlusion) against the lanes
    # replayed for it at 8 x 8 granularity (today: the quadrants the ellipse test keeps) and at 4 x 4 granularity (ideal culling: the
    # blocks that hold at least one reaching pixel) ------------------------------------------------------------------------------
    keep_i = live.nonzero().squeeze(1)
    px = torch.arange(16, device=dev, dtype=torch.float32)
    reach_px = quads8 = blocks4 = 0
    for c0 in range(0, int(keep_i.shape[0]), 1 << 17):
        k_ = keep_i[c0:c0 + (1 << 17)]
        dx = (tx0[k_].view(-1, 1, 1) + px.view(1, 1, 16)) - u[k_].view(-1, 1, 1)          # [n, 1, 16]
        dy = (ty0[k_].view(-1, 1, 1) + px.view(1, 16, 1)) - v[k_].view(-1, 1, 1)          # [n, 16, 1]
        power = -0.5 * (A[k_].view(-1, 1, 1) * dx * dx + Cc[k_].view(-1, 1, 1) * dy * dy) - B[k_].view(-1, 1, 1) * dx * dy
        ok = (power <= 0) & (o[k_].view(-1, 1, 1) * torch.exp(power) >= 1.0 / 255.0)         # [n, 16, 16]
        reach_px += int(ok.sum())
        quads8 += int(ok.view(-1, 2, 8, 2, 8).any(dim=4).any(dim=2).sum())
        blocks4 += int(ok.view(-1, 4, 4, 4, 4).any(dim=4).any(dim=2).sum())
    f8, f4 = reach_px / max(64.0 * quads8, 1), reach_px / max(16.0 * blocks4, 1)
    # instructions per reaching pixel-entry: replay 33 per group of 64 lanes; reduction 74 cross-lane per 4 entries today (one per entry),
    # 36 DPP adds per 4 (entry, block) pairs at 16-lane granularity (one per pair)
    today = 33.0 / (64.0 * f8) + (74.0 / 4.0) / (reach_px / max(n_live, 1))
    blocks = 33.0 / (64.0 * f4) + 36.0 / (64.0 * f4)
    print(f"reach statistics (geometry only): {reach_px / max(n_live, 1):.1f} reaching pixels per live instance; 8 x 8 quadrants with a reaching pixel "
          f"{quads8 / max(n_live, 1):.2f} per instance, lane share {f8:.3f}; 4 x 4 blocks {blocks4 / max(n_live, 1):.2f} per instance, lane share {f4:.3f}", flush=True)
    print(f"vector instructions per reaching pixel-entry: today {today:.3f} (replay {33.0 / (64.0 * f8):.3f} + reduction "
          f"{(74.0 / 4.0) / (reach_px / max(n_live, 1)):.3f}); per-block queues {blocks:.3f} (replay {33.0 / (64.0 * f4):.3f} + 16-lane reduction "
          f"{36.0 / (64.0 * f4):.3f}, before the per-entry sum across blocks): formulation (i) {'loses' if blocks >= today else 'wins'} by "
          f"{100.0 * (blocks / today - 1.0):+.0f} %", flush=True)

    # ---- the backward on the production lists ------------------------------------------------------------------------------
    g = torch.Generator(device=dev).manual_seed(5)
    dL = torch.randn(3, H, W, device=dev, generator=g)
    grads = lambda: [torch.empty(P, k, device=dev) for k in (3, 3, 3, 1, 3, 4)]  # noqa: E731
    scratch = torch.zeros(int(L.gsvc_raster_backward_scratch_bytes(P, m)) // 4, device=dev)

    def backward(binning, image_state, out):
        _lib.check(L.gsvc_raster_backward(C.byref(cs), P, m, _lib.ptr(xyz), _lib.ptr(color), _lib.ptr(opacity), _lib.ptr(scaling), _lib.ptr(rot),
                                          _lib.ptr(radii), _lib.ptr(st.geom), _lib.ptr(binning), _lib.ptr(image_state), _lib.ptr(dL),
                                          *[_lib.ptr(t) for t in out], _lib.ptr(scratch), _lib.current_stream(dev)), "gsvc_raster_backward")

    def timed(binning, image_state, tag, reps=20):
        out = grads()
        for _ in range(3):
            backward(binning, image_state, out)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(reps):
            backward(binning, image_state, out)
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        _lib.profile_enable(False)
        res = {k: 1e3 * ms / max(c, 1) for k, (c, ms) in prof.items()}
        print(f"{tag:44s} " + "  ".join(f"{k} {v:7.1f} us" for k, v in sorted(res.items())), flush=True)
        return out, res
    base_out, base_t = timed(st.binning, st.image_state, "production lists (tight binning, alpha box)")

    # ---- the lists an exact ellipse-against-tile binning would have produced ------------------------------------------------
    blob2 = st.binning.clone()
    img2 = st.image_state.clone()
    c = torch.cumsum(live.to(torch.int32), 0)                                  # live entries up to and including position k
    c0 = torch.cat([c.new_zeros(1), c])                                        # ... before position k
    new_off = c0[off.long()].to(torch.int32)                                   # tile t starts behind the live entries before it
    blob2[a.value:a.value + 4 * (T + 1)].view(torch.int32).copy_(new_off)
    keep = live.nonzero().squeeze(1)
    blob2[off_pl:off_pl + 4 * n_live].view(torch.int32).copy_(pl[keep])
    blob2[off_bb:off_bb + 8 * n_live].view(torch.int32).view(n_live, 2).copy_(bbox[keep])
    blob2[off_gs:off_gs + 4 * n_live].view(torch.int32).copy_(gslot[keep])
    # n_contrib: 1-based position of a pixel's last contributor inside its tile's list -> its position among the live entries
    ia, ib = C.c_uint64(), C.c_uint64()
    _lib.check(L.gsvc_raster_image_layout(C.byref(cs), C.byref(ia), C.byref(ib)), "image layout")
    nc = img2[ib.value:ib.value + 4 * T * 256].view(torch.int32)
    tile_px = torch.arange(T * 256, device=dev) // 256
    beg = off[:-1].long()[tile_px]
    last = nc.long()
    pos = beg + last                                                            # entries [beg, beg + last) = up to the last contributor
    new_last = (c0[pos] - c0[beg]).to(torch.int32)
    assert bool(((last == 0) | live[(pos - 1).clamp_min(0)]).all()), "a pixel's last contributor was classified dead"
    nc.copy_(torch.where(last > 0, new_last, nc))
    scratch.zero_()                                                             # the removed instances' rows: zeros (never written now)
    new_out, new_t = timed(blob2, img2, "exact lists (ellipse reaches the tile)")
    for nm, x, y in zip(("means3D", "means2D", "colors", "opacities", "scales", "rotations"), base_out, new_out):
        same = torch.equal(x, y)
        print(f"  d{nm}: {'bit-identical' if same else 'DIFFERS max ' + format(float((x - y).abs().max()), '.3e')}", flush=True)
    kb, kn = base_t.get("k_blend_bwd", 0.0), new_t.get("k_blend_bwd", 0.0)
    gb, gn = base_t.get("k_gaussian_bwd", 0.0), new_t.get("k_gaussian_bwd", 0.0)
    print(f"k_blend_bwd_tile {kb:.1f} -> {kn:.1f} us ({100.0 * (kn - kb) / kb:+.1f} %), k_gaussian_bwd {gb:.1f} -> {gn:.1f} us (it reads the same rows: "
          f"the row index keeps the rectangle), instances {n} -> {n_live} ({100.0 * (n_live - n) / n:+.1f} %)", flush=True)


if __name__ == "__main__":
    main()
