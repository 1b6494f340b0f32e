"""Losses of the fitting step: L1, SSIM (11x11 Gaussian window, sigma 1.5, zero padding), optical-flow
consistency of Gaussian motion between adjacent frames.

Same values as reference utils/loss_utils.py:20-72 (l1_loss_func, ssim_func/_ssim) and :76-153
(calc_optical_loss_one_frame / calc_optical_loss).  SSIM uses the separability of the window (two 1-D
passes instead of one 11x11 depthwise conv per moment) and filters the five moments in one batched call.
"""
from __future__ import annotations

from functools import lru_cache
from math import exp

import torch
import torch.nn.functional as F


class _SsimL1(torch.autograd.Function):
    """(mean SSIM, mean |a-b|) of two [C,H,W] CUDA images through csrc/ssim.hip; gradient w.r.t. the first."""

    @staticmethod
    def forward(ctx, img1, img2):
        from . import _lib
        a, b = img1.contiguous().float(), img2.contiguous().float()
        C_, H, W = a.shape
        need = img1.requires_grad
        sums = torch.empty(2, device=a.device, dtype=torch.float32)
        work = torch.empty(2048, device=a.device, dtype=torch.float32)
        maps = [torch.empty_like(a) for _ in range(3)] if need else [None, None, None]
        _lib.check(_lib.lib().gsvc_ssim_l1_forward(_lib.ptr(a), _lib.ptr(b), C_, H, W, _lib.ptr(sums), _lib.ptr(work), _lib.ptr(maps[0]),
                                                   _lib.ptr(maps[1]), _lib.ptr(maps[2]), _lib.current_stream(a.device)),
                   "gsvc_ssim_l1_forward")
        if need:
            ctx.save_for_backward(a, b, *maps)
        out = sums / float(C_ * H * W)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_ssim, g_l1):
        from . import _lib
        a, b, m0, m1, m2 = ctx.saved_tensors
        C_, H, W = a.shape
        grads = torch.stack([g_ssim, g_l1]).float().contiguous()
        d = torch.empty_like(a)
        _lib.check(_lib.lib().gsvc_ssim_l1_backward(_lib.ptr(a), _lib.ptr(b), C_, H, W, _lib.ptr(grads), _lib.ptr(m0),
                                                    _lib.ptr(m1), _lib.ptr(m2), _lib.ptr(d), _lib.current_stream(a.device)),
                   "gsvc_ssim_l1_backward")
        return d, None


class _SsimL1Pair(torch.autograd.Function):
    """(mean SSIM, mean |a-b|, a) with a = (img_f + flip_W(img_b)) / 2 formed inside the kernels (the two-view frame of
    reference pipeline/train.py:368-375): no flip / add / div launches forward, no div / flip backward."""

    @staticmethod
    def forward(ctx, img_f, img_b, img2):
        from . import _lib
        f, b, g = img_f.contiguous().float(), img_b.contiguous().float(), img2.contiguous().float()
        C_, H, W = f.shape
        need = img_f.requires_grad or img_b.requires_grad
        sums = torch.empty(2, device=f.device, dtype=torch.float32)
        work = torch.empty(2048, device=f.device, dtype=torch.float32)
        avg = torch.empty_like(f)
        maps = [torch.empty_like(f) for _ in range(3)] if need else [None, None, None]
        _lib.check(_lib.lib().gsvc_ssim_l1_pair_forward(_lib.ptr(f), _lib.ptr(b), _lib.ptr(g), C_, H, W, _lib.ptr(sums), _lib.ptr(work),
                                                        _lib.ptr(maps[0]), _lib.ptr(maps[1]), _lib.ptr(maps[2]), _lib.ptr(avg),
                                                        _lib.current_stream(f.device)), "gsvc_ssim_l1_pair_forward")
        if need:
            ctx.save_for_backward(f, b, g, *maps)
        ctx.mark_non_differentiable(avg)
        ctx.set_materialize_grads(False)      # no zero image for the averaged frame's (absent) gradient
        out = sums / float(C_ * H * W)
        return out[0], out[1], avg

    @staticmethod
    def backward(ctx, g_ssim, g_l1, _g_avg):
        from . import _lib
        f, b, g, m0, m1, m2 = ctx.saved_tensors
        C_, H, W = f.shape
        zero = None
        if g_ssim is None or g_l1 is None:
            zero = torch.zeros((), device=f.device)
        grads = torch.stack([zero if g_ssim is None else g_ssim, zero if g_l1 is None else g_l1]).float().contiguous()
        df, db = torch.empty_like(f), torch.empty_like(b)
        _lib.check(_lib.lib().gsvc_ssim_l1_pair_backward(_lib.ptr(f), _lib.ptr(b), _lib.ptr(g), C_, H, W, _lib.ptr(grads), _lib.ptr(m0),
                                                         _lib.ptr(m1), _lib.ptr(m2), _lib.ptr(df), _lib.ptr(db),
                                                         _lib.current_stream(f.device)), "gsvc_ssim_l1_pair_backward")
        return df, db, None


def ssim_l1_pair(img_f, img_b, img2):
    """(ssim_func(a, img2), l1_loss_func(a, img2), a.detach()) for a = (img_f + flip(img_b, W)) / 2, [3,H,W] CUDA images."""
    return _SsimL1Pair.apply(img_f, img_b, img2)


def ssim_l1(img1, img2):
    """Fused (ssim_func(img1, img2), l1_loss_func(img1, img2)) for [3,H,W] CUDA images (one kernel each way)."""
    return _SsimL1.apply(img1, img2)


def l1_loss_func(network_output, gt):
    return (network_output - gt).abs().mean()


def l2_loss_func(network_output, gt):
    return ((network_output - gt) ** 2).mean()


@lru_cache(maxsize=4)
def _window_1d(window_size: int, sigma: float):
    g = torch.tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


def _blur(x, w1d, pad):
    """Depthwise separable Gaussian blur of [B,C,H,W] with zero padding."""
    C = x.shape[1]
    kh = w1d.view(1, 1, -1, 1).expand(C, 1, -1, 1)
    kw = w1d.view(1, 1, 1, -1).expand(C, 1, 1, -1)
    x = F.conv2d(x, kh, padding=(pad, 0), groups=C)
    return F.conv2d(x, kw, padding=(0, pad), groups=C)


def ssim_func(img1, img2, window_size=11, size_average=True):
    if img1.is_cuda and window_size == 11:
        # GPU tensors always take the fused HIP kernel (per image when a batch is given)
        if img1.dim() == 3:
            return _SsimL1.apply(img1, img2)[0]
        per = torch.stack([_SsimL1.apply(a, b)[0] for a, b in zip(img1, img2)])
        return per.mean() if size_average else per
    squeeze = img1.dim() == 3
    a = img1.unsqueeze(0) if squeeze else img1
    b = img2.unsqueeze(0) if squeeze else img2
    w = _window_1d(window_size, 1.5).to(device=a.device, dtype=a.dtype)
    pad = window_size // 2
    B = a.shape[0]
    m = _blur(torch.cat([a, b, a * a, b * b, a * b], dim=0), w, pad)
    mu1, mu2, e11, e22, e12 = m[:B], m[B:2 * B], m[2 * B:3 * B], m[3 * B:4 * B], m[4 * B:]
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1, s2, s12 = e11 - mu1_sq, e22 - mu2_sq, e12 - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    if size_average:
        return ssim_map.mean()
    return ssim_map.mean(1).mean(1).mean(1)


def _alive_slots(res, n_offsets):
    """Flat (anchor*K + slot) indices of the Gaussians a render generated (opacity > 0), in generation order."""
    vis = res.visible_mask
    idx = torch.arange(vis.shape[0] * n_offsets, device=vis.device).view(-1, n_offsets)
    return idx[vis].reshape(-1)[res.generated_gaussians.mask]


def _world_xy(res, keep):
    rows = res.generated_gaussians.concatenated_all.index_select(0, keep.nonzero(as_tuple=False).squeeze(1))
    scaling, anchor, offsets = rows[:, 0:6], rows[:, 6:9], rows[:, 19:22]
    return (anchor + offsets * scaling[:, :3])[:, :2]


def calc_optical_loss_one_frame(render_results1, render_results2, optical_flow, x_min, y_min, scale,
                                x_pix_max: int, y_pix_max: int, n_offsets=10):
    """Gaussians alive in both renders (same anchor, same offset slot): their world-xy displacement should
    equal the optical flow sampled at the frame-1 pixel (reference loss_utils.py:76-135)."""
    dev = render_results1.visible_mask.device
    total = render_results1.visible_mask.shape[0] * n_offsets
    alive1 = torch.zeros(total, dtype=torch.bool, device=dev)
    alive2 = torch.zeros(total, dtype=torch.bool, device=dev)
    alive1[_alive_slots(render_results1, n_offsets)] = True
    alive2[_alive_slots(render_results2, n_offsets)] = True
    common = (alive1 & alive2).view(-1, n_offsets)
    keep1 = common[render_results1.visible_mask].reshape(-1) & render_results1.generated_gaussians.mask
    keep2 = common[render_results2.visible_mask].reshape(-1) & render_results2.generated_gaussians.mask
    xy1 = _world_xy(render_results1, keep1)
    xy2 = _world_xy(render_results2, keep2)
    from .generate import host_values
    pix = ((xy1 - host_values([[x_min, y_min]], dev, xy1.dtype)) * scale).round().long()
    ok = (pix[:, 0] >= 0) & (pix[:, 1] >= 0) & (pix[:, 0] < x_pix_max) & (pix[:, 1] < y_pix_max)
    oki = ok.nonzero(as_tuple=False).squeeze(1)
    pix = pix.index_select(0, oki)
    flow = optical_flow.permute(2, 1, 0).to(dev)
    uv = flow[pix[:, 0], pix[:, 1], ...] / scale
    d = xy2.index_select(0, oki) - xy1.index_select(0, oki)
    return (d - uv).abs().mean(), pix, d * scale


class _OpticalPair(torch.autograd.Function):
    """Optical-flow consistency of one adjacent-frame pair over un-compacted renders (csrc/losses.hip)."""

    @staticmethod
    def forward(ctx, world1, world2, mask1, mask2, vis1, vis2, flow, K, anchors, x_min, y_min, scale, x_pix_max, y_pix_max):
        from . import _lib
        dev = world1.device
        world1, world2 = world1.contiguous(), world2.contiguous()
        m1, m2 = mask1.contiguous().view(torch.uint8), mask2.contiguous().view(torch.uint8)
        flow = flow.contiguous()
        n1, n2 = world1.shape[0], world2.shape[0]
        table = torch.empty(anchors * K, dtype=torch.int32, device=dev)
        partner = torch.empty(max(n1, 1), dtype=torch.int32, device=dev)
        sums = torch.empty(2, dtype=torch.float32, device=dev)
        partial = torch.empty(2 * max((n1 + 255) // 256, 1), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().gsvc_optical_forward(
            _lib.ptr(world1), _lib.ptr(m1), _lib.ptr(vis1.contiguous()), n1, _lib.ptr(world2), _lib.ptr(m2), _lib.ptr(vis2.contiguous()),
            n2, K, anchors, _lib.ptr(flow), flow.shape[1], flow.shape[2], float(x_min), float(y_min), float(scale), int(x_pix_max),
            int(y_pix_max), _lib.ptr(table), _lib.ptr(partner), _lib.ptr(sums), _lib.ptr(partial), _lib.current_stream(dev)),
            "gsvc_optical_forward")
        ctx.save_for_backward(partner, sums)
        ctx.n = (n1, n2)
        return sums[0] / (2.0 * sums[1])

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        partner, sums = ctx.saved_tensors
        n1, n2 = ctx.n
        dev = partner.device
        g1 = torch.empty(n1, 3, dtype=torch.float32, device=dev)
        g2 = torch.empty(n2, 3, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().gsvc_optical_backward(_lib.ptr(partner), n1, n2, _lib.ptr(sums), _lib.ptr(g.contiguous().float()),
                                                    _lib.ptr(g1), _lib.ptr(g2), _lib.current_stream(dev)), "gsvc_optical_backward")
        return (g1, g2) + (None,) * 12


def _optical_loss_dense(r1, r2, optical_flow, x_min, y_min, scale, x_pix_max: int, y_pix_max: int, n_offsets=10):
    """calc_optical_loss_one_frame for un-compacted results (render_many(dense=True)): the "alive in both renders"
    intersection and the pairing of the two renders' Gaussians go through an [anchors*K] slot table inside the fused
    kernels of csrc/losses.hip — the same mean over the same pairs, with no compaction and no host synchronisation."""
    g1, g2 = r1.generated_gaussians, r2.generated_gaussians
    return _OpticalPair.apply(g1.world_xyz, g2.world_xyz, g1.mask, g2.mask, r1.visible_index, r2.visible_index,
                              optical_flow.to(g1.world_xyz.device), n_offsets, r1.visible_mask.shape[0], x_min, y_min, scale,
                              x_pix_max, y_pix_max)


class _OpticalMany(torch.autograd.Function):
    """calc_optical_loss over renders that are row ranges of one set of tensors (render_many(dense=True)): every pair in four
    launches forward and one backward, the gradient in the concatenated layout (csrc/losses.hip k_optical_*_many)."""

    @staticmethod
    def forward(ctx, world, mask, vis, goff, pairs, flow, K, anchors, x_min, y_min, scale, x_pix_max, y_pix_max):
        import ctypes as C
        from . import _lib
        dev = world.device
        world, vis, flow = world.contiguous(), vis.contiguous(), flow.contiguous()
        m = mask.contiguous().view(torch.uint8)
        R, P = len(goff) - 1, len(pairs)
        off = (C.c_int64 * (R + 1))(*[int(v) for v in goff])
        src, dst = (C.c_int32 * P)(*[p[0] for p in pairs]), (C.c_int32 * P)(*[p[1] for p in pairs])
        L = _lib.lib()
        table = torch.empty(R * anchors, dtype=torch.int32, device=dev)
        partner = torch.empty(max(int(goff[-1]), 1), dtype=torch.int32, device=dev)
        res = torch.empty(2 * P + 1 + int(L.gsvc_optical_many_partial_floats(off, R, src, dst, P, K)), dtype=torch.float32, device=dev)
        sums, loss, partial = res[:2 * P], res[2 * P:2 * P + 1], res[2 * P + 1:]
        _lib.check(L.gsvc_optical_many_forward(_lib.ptr(world), _lib.ptr(m), _lib.ptr(vis), off, R, src, dst, P, K, anchors, _lib.ptr(flow),
                                               flow.shape[1], flow.shape[2], float(x_min), float(y_min), float(scale), int(x_pix_max),
                                               int(y_pix_max), _lib.ptr(table), _lib.ptr(partner), _lib.ptr(sums), C.c_void_p(partial.data_ptr()),
                                               C.c_void_p(loss.data_ptr()), _lib.current_stream(dev)), "gsvc_optical_many_forward")
        ctx.save_for_backward(partner, vis, table, sums)
        ctx.meta = (off, R, src, dst, P, K, anchors, world.shape)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        partner, vis, table, sums = ctx.saved_tensors
        off, R, src, dst, P, K, anchors, shape = ctx.meta
        gw = torch.empty(shape, dtype=torch.float32, device=partner.device)
        _lib.check(_lib.lib().gsvc_optical_many_backward(_lib.ptr(partner), _lib.ptr(vis), off, R, src, dst, P, K, anchors, _lib.ptr(table),
                                                         _lib.ptr(sums), _lib.ptr(g.contiguous().float()), _lib.ptr(gw),
                                                         _lib.current_stream(partner.device)), "gsvc_optical_many_backward")
        return (gw,) + (None,) * 12


class _RenderRegs(torch.autograd.Function):
    """(sum_r mean over opacity>0 of prod(scaling), sum_r mean(1 - neural_opacity)) over the concatenated
    un-compacted Gaussians of R renders (csrc/losses.hip)."""

    @staticmethod
    def forward(ctx, scaling, neural_opacity, mask, seg_offsets):
        import ctypes as C
        from . import _lib
        dev = scaling.device
        scaling, op = scaling.contiguous(), neural_opacity.contiguous()
        m = mask.contiguous().view(torch.uint8)
        R = len(seg_offsets) - 1
        seg = (C.c_int64 * (R + 1))(*[int(v) for v in seg_offsets])
        L = _lib.lib()
        sums = torch.empty(3 * R, dtype=torch.float32, device=dev)
        partial = torch.empty(int(L.gsvc_regs_partial_floats(seg, R)), dtype=torch.float32, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        _lib.check(L.gsvc_regs_forward(_lib.ptr(scaling), _lib.ptr(op), _lib.ptr(m), seg, R, _lib.ptr(sums), _lib.ptr(partial),
                                       _lib.ptr(out), _lib.current_stream(dev)), "gsvc_regs_forward")
        ctx.save_for_backward(scaling, m, sums)
        ctx.seg, ctx.R, ctx.op_shape = seg, R, neural_opacity.shape
        ctx.set_materialize_grads(False)
        return out[0], out[1]      # two scalars (indexing a returned 2-vector costs a zero-fill + copy + add per element in the backward)

    @staticmethod
    def backward(ctx, g0, g1):
        from . import _lib
        scaling, m, sums = ctx.saved_tensors
        dev = scaling.device
        if g0 is None and g1 is None:
            return None, None, None, None
        zero = torch.zeros((), device=dev) if (g0 is None or g1 is None) else None
        g = torch.stack([zero if g0 is None else g0, zero if g1 is None else g1])
        gs = torch.empty_like(scaling)
        go = torch.empty(scaling.shape[0], dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().gsvc_regs_backward(_lib.ptr(scaling), _lib.ptr(m), ctx.seg, ctx.R, _lib.ptr(sums),
                                                 _lib.ptr(g.contiguous().float()), _lib.ptr(gs), _lib.ptr(go),
                                                 _lib.current_stream(dev)), "gsvc_regs_backward")
        return gs, go.view(ctx.op_shape), None, None


def render_regs(scaling, neural_opacity, mask, seg_offsets):
    """Returns the pair (scaling regulariser, opacity regulariser) summed over the renders delimited by seg_offsets."""
    return _RenderRegs.apply(scaling, neural_opacity, mask, seg_offsets)


def calc_optical_loss(render_results1_f, render_results1_b, render_results2_f, render_results2_b, optical_flow,
                      x_min, y_min, scale, x_pix_max: int, y_pix_max: int, n_offsets=10):
    if render_results1_f.dense:
        rs = (render_results1_f, render_results1_b, render_results2_f, render_results2_b)
        batch = getattr(rs[0].generated_gaussians, "batch", None)
        if (batch is not None and getattr(batch, "world", None) is not None and batch.world.is_cuda and len(batch.seg_offsets) == 5
                and all(getattr(r.generated_gaussians, "batch", None) is batch for r in rs)):
            # the four renders are row ranges of the batch's tensors, in this order: both pairs in one pass
            return _OpticalMany.apply(batch.world, batch.mask, batch.vis, batch.seg_offsets, ((0, 2), (1, 3)),
                                      optical_flow.to(batch.world.device), n_offsets, rs[0].visible_mask.shape[0], x_min, y_min, scale,
                                      x_pix_max, y_pix_max)
        args = (optical_flow, x_min, y_min, scale, x_pix_max, y_pix_max, n_offsets)
        return (_optical_loss_dense(render_results1_f, render_results2_f, *args)
                + _optical_loss_dense(render_results1_b, render_results2_b, *args))
    lf, _, _ = calc_optical_loss_one_frame(render_results1_f, render_results2_f, optical_flow, x_min, y_min, scale,
                                           x_pix_max, y_pix_max, n_offsets)
    lb, _, _ = calc_optical_loss_one_frame(render_results1_b, render_results2_b, optical_flow, x_min, y_min, scale,
                                           x_pix_max, y_pix_max, n_offsets)
    return lf + lb


def psnr_func(img1, img2, data_range=1):
    """reference utils/metric_utils.py:10-13"""
    return 10 * torch.log10((data_range ** 2) / torch.mean((img1 - img2) ** 2))
