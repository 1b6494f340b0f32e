"""Evaluation loop and checkpoints (harness around the hot path).

``evaluate`` reports what reference utils/report_utils.py:268-390 logs for a fitted model — mean L1, PSNR, SSIM and MS-SSIM
of the two-view frames against the ground truth, and the frame rate — but renders through the decoder loop
(``render_frames``: batched generation, one two-view pass per frame) instead of two ``render`` calls, a flip and an
average per frame.  LPIPS (``lpips_fn``: gsvc_amd.lpips.LPIPS built from a weights file — the pretrained numbers are not in this
image) is reported when handed in, as the reference calls it (``lpips_fn(image, gt, normalize=True)``, report_utils.py:154).  ``save_checkpoint`` / ``load_checkpoint`` keep a model and
its optimizer in one ``torch.save`` file (the reference scatters this over ply / pkl files with third-party readers).
"""
from __future__ import annotations

import time

import torch

from .loss_utils import l1_loss_func, psnr_func, ssim_func
from .metrics import msssim_fn
from .ortho_gaussian_renderer import render_frames


@torch.no_grad()
def evaluate(pc, dataset, pipe, bg_color, frame_ids=None, batch: int = 8, lpips_fn=None) -> dict:
    """Mean L1 / PSNR / SSIM / MS-SSIM (MS-SSIM only for frames at least 160 pixels high and large enough for 5 scales) of the
    rendered two-view frames, clamped to [0, 1], against ``dataset[i].image``; ``fps`` counts the whole loop's wall time,
    metrics excluded."""
    ids = list(range(dataset.len_z_frames)) if frame_ids is None else list(frame_ids)
    frames = [dataset[i] for i in ids]
    for _ in render_frames(frames[:min(len(frames), batch)], pc, pipe, bg_color, batch=batch):      # warm-up, as the reference does
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    images = [torch.clamp(img, 0.0, 1.0) for img in render_frames(frames, pc, pipe, bg_color, batch=batch)]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sums = {"l1": 0.0, "psnr": 0.0, "ssim": 0.0, "msssim": 0.0, "lpips": 0.0}
    n_ms = 0
    for fr, img in zip(frames, images):
        gt = torch.clamp(fr.image.to(img.device), 0.0, 1.0).permute(0, 2, 1).contiguous()
        sums["l1"] += float(l1_loss_func(img, gt).mean())
        sums["psnr"] += float(psnr_func(img, gt))
        sums["ssim"] += float(ssim_func(img, gt).mean())
        if min(img.shape[-2:]) > 160:
            sums["msssim"] += float(msssim_fn(img.unsqueeze(0), gt.unsqueeze(0)))
            n_ms += 1
        if lpips_fn is not None:
            sums["lpips"] += float(lpips_fn(img, gt, normalize=True))
    n = max(len(frames), 1)
    return {"frames": len(frames), "l1": sums["l1"] / n, "psnr": sums["psnr"] / n, "ssim": sums["ssim"] / n,
            "msssim": sums["msssim"] / n_ms if n_ms else float("nan"), "lpips": sums["lpips"] / n if lpips_fn is not None else None, "fps": len(frames) / elapsed if elapsed > 0 else float("inf")}


def _optimizer_state(optimizer):
    return optimizer.state_dict()


def _plain(obj):
    """NumPy scalars -> Python numbers, recursively (the learning-rate schedule leaves np.float64 in the optimizer's groups;
    a file holding them cannot be read back with ``weights_only=True``)."""
    import numpy as np
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_plain(v) for v in obj)
    if isinstance(obj, np.generic):
        return obj.item()
    return obj


def save_checkpoint(pc, path, iteration: int = 0):
    """Model parameters (reference-compatible state_dict keys), the per-anchor tensors, the densification statistics and
    the optimizer state in one file."""
    per_anchor = {n: getattr(pc, n).detach() for n in ("_anchor", "_offset", "_mask", "_anchor_feat", "_scaling", "_rotation", "_opacity")}
    stats = {n: getattr(pc, n) for n in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom", "max_radii2D")}
    torch.save({"iteration": int(iteration), "state_dict": pc.state_dict(), "per_anchor": per_anchor, "stats": stats,
                "decoded_version": bool(pc.decoded_version), "voxel_size": float(pc.voxel_size),
                "spatial_lr_scale": float(pc.spatial_lr_scale), "percent_dense": float(getattr(pc, "percent_dense", 0.0) or 0.0),
                "bounds": (tuple(float(v) for v in pc.bound_min_host), tuple(float(v) for v in pc.bound_max_host)),
                "optimizer": _plain(_optimizer_state(pc.optimizer)) if pc.optimizer is not None else None}, path)


def load_checkpoint(pc, path, training_args=None) -> int:
    """Restore what save_checkpoint wrote into a model built with the same hyper-parameters; with ``training_args`` the
    optimizer is set up and its state restored.  Returns the stored iteration."""
    ck = torch.load(path, map_location=pc.device, weights_only=True)      # tensors / numbers / tuples only: nothing is unpickled
    for n, t in ck["per_anchor"].items():
        setattr(pc, n, torch.nn.Parameter(t.to(pc.device).clone(), requires_grad=n not in ("_rotation", "_opacity")))
    pc.load_state_dict(ck["state_dict"], strict=True)
    pc.decoded_version, pc.voxel_size = ck["decoded_version"], ck["voxel_size"]
    lo, hi = ck["bounds"]
    pc.bound_min_host, pc.bound_max_host = tuple(lo), tuple(hi)
    pc.x_bound_min = torch.tensor([list(lo)], dtype=torch.float32, device=pc.device)
    pc.x_bound_max = torch.tensor([list(hi)], dtype=torch.float32, device=pc.device)
    pc.spatial_lr_scale = float(ck.get("spatial_lr_scale", pc.spatial_lr_scale))
    if training_args is not None:
        pc.training_setup(training_args)      # re-creates (zeroes) the densification statistics: restore them afterwards
        if ck["optimizer"] is not None:
            pc.optimizer.load_state_dict(ck["optimizer"])
    if ck.get("percent_dense"):
        pc.percent_dense = float(ck["percent_dense"])
    for n, t in ck["stats"].items():
        setattr(pc, n, t.to(pc.device) if isinstance(t, torch.Tensor) else t)
    return ck["iteration"]
