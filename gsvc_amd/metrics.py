"""Evaluation metrics of the reference's report loop on the device: PSNR and MS-SSIM.

``psnr_func`` and ``msssim_fn`` keep the names and argument meaning of reference utils/metric_utils.py:11-37.  The
reference takes MS-SSIM from the third-party package ``pytorch_msssim`` (not in its tree, version not pinned); this is a
restatement of that package's published algorithm (Wang et al. 2003 as implemented there: 11-tap Gaussian window with
sigma 1.5 applied separably WITHOUT padding, 5 scales halved by 2x2 average pooling, contrast-structure terms of the
first four scales and the full SSIM of the last, exponents 0.0448 / 0.2856 / 0.3001 / 0.2363 / 0.1333, negative terms
clipped to 0) — parity unpinned, checked against an independent NumPy / SciPy implementation in tests/test_golden_host.py.
LPIPS: gsvc_amd/lpips.py (needs a weights file).  Plain torch ops: these run a few times per evaluation, not per step.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .loss_utils import psnr_func  # noqa: F401  (re-export: reference utils/metric_utils.py:11-14)

MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _gauss_window(size: int, sigma: float, device, dtype):
    x = torch.arange(size, dtype=dtype, device=device) - size // 2
    g = torch.exp(-(x ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _filter(x, win):
    """Separable valid (unpadded) convolution of every channel with the 1-D window along H, then W."""
    c = x.shape[1]
    k = win.numel()
    if x.shape[2] >= k:
        x = F.conv2d(x, win.view(1, 1, k, 1).expand(c, 1, k, 1), groups=c)
    if x.shape[3] >= k:
        x = F.conv2d(x, win.view(1, 1, 1, k).expand(c, 1, 1, k), groups=c)
    return x


def _ssim_cs(x, y, win, data_range, K=(0.01, 0.03)):
    c1, c2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = _filter(x, win), _filter(y, win)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = _filter(x * x, win) - mu1_sq
    s2 = _filter(y * y, win) - mu2_sq
    s12 = _filter(x * y, win) - mu12
    cs_map = (2 * s12 + c2) / (s1 + s2 + c2)
    ssim_map = ((2 * mu12 + c1) / (mu1_sq + mu2_sq + c1)) * cs_map
    return ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)


def ms_ssim(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0, size_average: bool = True, win_size: int = 11,
            win_sigma: float = 1.5, weights=MS_WEIGHTS) -> torch.Tensor:
    """Multi-scale SSIM of two image batches [N, C, H, W] (a [C, H, W] image is taken as a batch of one)."""
    if x.dim() == 3:
        x, y = x.unsqueeze(0), y.unsqueeze(0)
    if x.shape != y.shape or x.dim() != 4:
        raise ValueError("ms_ssim: inputs must be two [N, C, H, W] tensors of the same shape")
    levels = len(weights)
    if min(x.shape[-2:]) <= (win_size - 1) * 2 ** (levels - 1):
        raise ValueError(f"ms_ssim: image sides must exceed {(win_size - 1) * 2 ** (levels - 1)} pixels for {levels} scales")
    x, y = x.float(), y.float()
    win = _gauss_window(win_size, win_sigma, x.device, x.dtype)
    mcs = []
    for i in range(levels):
        ssim_c, cs = _ssim_cs(x, y, win, data_range)
        if i < levels - 1:
            mcs.append(torch.relu(cs))
            pad = [s % 2 for s in x.shape[2:]]
            x = F.avg_pool2d(x, kernel_size=2, padding=pad)
            y = F.avg_pool2d(y, kernel_size=2, padding=pad)
    terms = torch.stack(mcs + [torch.relu(ssim_c)], dim=0)                      # [levels, N, C]
    w = torch.tensor(weights, dtype=x.dtype, device=x.device).view(-1, 1, 1)
    val = torch.prod(terms ** w, dim=0)
    return val.mean() if size_average else val.mean(1)


def msssim_fn(output, target):
    """reference utils/metric_utils.py:33-37"""
    assert output.size(-2) >= 160
    return ms_ssim(output.float().detach(), target.detach(), data_range=1, size_average=True)
