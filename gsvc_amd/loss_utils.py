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
    pix = ((xy1 - torch.tensor([[x_min, y_min]], dtype=xy1.dtype, device=dev)) * scale).round().long()
    ok = (pix[:, 0] >= 0) & (pix[:, 1] >= 0) & (pix[:, 0] < x_pix_max) & (pix[:, 1] < y_pix_max)
    oki = ok.nonzero(as_tuple=False).squeeze(1)
    pix = pix.index_select(0, oki)
    flow = optical_flow.permute(2, 1, 0).to(dev)
    uv = flow[pix[:, 0], pix[:, 1], ...] / scale
    d = xy2.index_select(0, oki) - xy1.index_select(0, oki)
    return (d - uv).abs().mean(), pix, d * scale


def _optical_loss_dense(r1, r2, optical_flow, x_min, y_min, scale, x_pix_max: int, y_pix_max: int, n_offsets=10):
    """calc_optical_loss_one_frame for un-compacted results (render_many(dense=True)): the "alive in both renders"
    intersection and the pairing of the two renders' Gaussians go through [anchors, K] tables indexed by anchor
    row, and membership becomes a 0/1 weight — the same mean over the same pairs, with no compaction and therefore
    no host synchronisation."""
    K = n_offsets
    dev = r1.visible_mask.device
    A = r1.visible_mask.shape[0]
    v1, v2 = r1.visible_index, r2.visible_index
    m1 = r1.generated_gaussians.mask.view(-1, K)
    m2 = r2.generated_gaussians.mask.view(-1, K)
    alive2 = torch.zeros(A, K, dtype=torch.bool, device=dev).index_put_((v2,), m2)
    xy2_table = torch.zeros(A, K, 2, dtype=r2.generated_gaussians.world_xyz.dtype, device=dev)
    xy2_table = xy2_table.index_put((v2,), r2.generated_gaussians.world_xyz.view(-1, K, 3)[:, :, :2])
    keep = (m1 & alive2.index_select(0, v1)).view(-1)                      # per Gaussian of render 1
    xy1 = r1.generated_gaussians.world_xyz[:, :2]
    xy2 = xy2_table.index_select(0, v1).view(-1, 2)
    pix = ((xy1 - torch.tensor([[x_min, y_min]], dtype=xy1.dtype, device=dev)) * scale).round().long()
    ok = (pix[:, 0] >= 0) & (pix[:, 1] >= 0) & (pix[:, 0] < x_pix_max) & (pix[:, 1] < y_pix_max)
    w = (keep & ok).to(xy1.dtype).unsqueeze(1)
    flow = optical_flow.permute(2, 1, 0).to(dev)
    uv = flow[pix[:, 0].clamp(0, x_pix_max - 1), pix[:, 1].clamp(0, y_pix_max - 1), ...] / scale
    d = xy2 - xy1
    return ((d - uv).abs() * w).sum() / (2.0 * w.sum())


def calc_optical_loss(render_results1_f, render_results1_b, render_results2_f, render_results2_b, optical_flow,
                      x_min, y_min, scale, x_pix_max: int, y_pix_max: int, n_offsets=10):
    if render_results1_f.dense:
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
