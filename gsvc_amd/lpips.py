"""LPIPS (learned perceptual image patch similarity, Zhang et al. 2018, v0.1) for the evaluation loop.

The reference evaluates ``lpips_fn(image, gt_image, normalize=True)`` with ``lpips_fn = lpips.LPIPS().cuda()`` — the third-party
``lpips`` package with its defaults: AlexNet features, linear heads v0.1 (reference utils/metric_utils.py:8,44,
utils/report_utils.py:154); its tree also carries ``lpipsPyTorch`` (AlexNet / VGG-16 variants of the same metric).  Both need
PRETRAINED weights (torchvision's ImageNet backbone + the LPIPS linear heads), which neither the reference tree nor this image
holds and which cannot be fetched here.  So the metric is built with a **weights-path argument**: the arithmetic — input scaling,
backbone feature taps, unit-normalisation over channels, squared difference, 1x1 linear heads, spatial mean, sum over the taps — is
this file; the numbers come from a file the user points at.  Accepted layouts of that file (a ``torch.save``d state dict, loaded
with ``weights_only=True``): the ``lpips`` package's own (``net.slice{k}.{i}.weight`` + ``lin{k}.model.1.weight``), or two files
(torchvision's ``features.{i}.weight`` backbone + the package's ``lin{k}.model.1.weight`` heads).  Without weights the module
refuses to evaluate, unless built with ``random_init=True`` (structural tests: tests/test_lpips_cpu.py checks the arithmetic against
a plain functional statement of the published algorithm on random weights — parity with the pretrained metric is therefore
"unpinned" until a weights file is supplied).  Plain torch ops on whatever device the images are on: the metric runs once per
evaluated frame, not per fitting step.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

# (out_channels, kernel, stride, padding) of the backbone's convolutions, "M" = 3x3/2 max pooling (AlexNet) or 2x2/2 (VGG-16),
# "T" = a feature tap behind the preceding ReLU
_ALEX = [(64, 11, 4, 2), "T", "M", (192, 5, 1, 2), "T", "M", (384, 3, 1, 1), "T", (256, 3, 1, 1), "T", (256, 3, 1, 1), "T"]
_VGG = [(64, 3, 1, 1), (64, 3, 1, 1), "T", "M", (128, 3, 1, 1), (128, 3, 1, 1), "T", "M", (256, 3, 1, 1), (256, 3, 1, 1), (256, 3, 1, 1), "T",
        "M", (512, 3, 1, 1), (512, 3, 1, 1), (512, 3, 1, 1), "T", "M", (512, 3, 1, 1), (512, 3, 1, 1), (512, 3, 1, 1), "T"]
# index of each convolution inside torchvision's ``features`` Sequential (what the pretrained files name the weights by)
_TV_INDEX = {"alex": [0, 3, 6, 8, 10], "vgg": [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]}
SHIFT = (-0.030, -0.088, -0.188)
SCALE = (0.458, 0.448, 0.450)


class LPIPS(nn.Module):
    def __init__(self, net: str = "alex", weights_path: str | None = None, lin_weights_path: str | None = None, random_init: bool = False):
        super().__init__()
        if net not in ("alex", "vgg"):
            raise ValueError("LPIPS: net must be 'alex' (the reference's evaluation default) or 'vgg'")
        self.net_type = net
        self.plan = _ALEX if net == "alex" else _VGG
        self.pool = (3, 2) if net == "alex" else (2, 2)
        convs, c_in, taps = [], 3, []
        for item in self.plan:
            if isinstance(item, tuple):
                convs.append(nn.Conv2d(c_in, item[0], item[1], item[2], item[3]))
                c_in = item[0]
            elif item == "T":
                taps.append(c_in)
        self.convs = nn.ModuleList(convs)
        self.lins = nn.ParameterList([nn.Parameter(torch.zeros(1, c, 1, 1)) for c in taps])
        self.register_buffer("shift", torch.tensor(SHIFT).view(1, 3, 1, 1))
        self.register_buffer("scale", torch.tensor(SCALE).view(1, 3, 1, 1))
        self.ready = False
        if weights_path is not None:
            self.load_pretrained(weights_path, lin_weights_path)
        elif random_init:
            g = torch.Generator().manual_seed(0)
            with torch.no_grad():
                for c in self.convs:
                    c.weight.copy_(torch.randn(c.weight.shape, generator=g) * (2.0 / (c.weight[0].numel())) ** 0.5)
                    c.bias.copy_(torch.randn(c.bias.shape, generator=g) * 0.05)
                for p in self.lins:
                    p.copy_(torch.rand(p.shape, generator=g) / p.shape[1])
            self.ready = True
        for p in self.parameters():
            p.requires_grad_(False)
        self.eval()

    def load_pretrained(self, weights_path: str, lin_weights_path: str | None = None):
        sd = dict(torch.load(weights_path, map_location="cpu", weights_only=True))
        if lin_weights_path is not None:
            sd.update(torch.load(lin_weights_path, map_location="cpu", weights_only=True))
        tv = _TV_INDEX[self.net_type]

        def find(cands):
            for k in cands:
                if k in sd:
                    return sd[k]
            raise KeyError(f"LPIPS weights: none of {cands} in the file(s)")
        # the ``lpips`` package groups torchvision's feature layers into slices but keeps torchvision's layer numbers as names
        slices = {"alex": [(1, 0), (2, 3), (3, 6), (4, 8), (5, 10)],
                  "vgg": [(1, 0), (1, 2), (2, 5), (2, 7), (3, 10), (3, 12), (3, 14), (4, 17), (4, 19), (4, 21), (5, 24), (5, 26), (5, 28)]}[self.net_type]
        with torch.no_grad():
            for conv, idx, (sl, _) in zip(self.convs, tv, slices):
                for part in ("weight", "bias"):
                    t = find([f"features.{idx}.{part}", f"net.slice{sl}.{idx}.{part}", f"net.layers.{idx}.{part}", f"layers.{idx}.{part}"])
                    getattr(conv, part).copy_(t)
            for k, p in enumerate(self.lins):
                p.copy_(find([f"lin{k}.model.1.weight", f"lins.{k}.model.1.weight", f"lin.{k}.1.weight", f"{k}.1.weight"]).view(p.shape))
        self.ready = True
        return self

    def features(self, x):
        """The backbone's taps (behind ReLU) of an already scaled input."""
        out, ci = [], 0
        for item in self.plan:
            if isinstance(item, tuple):
                x = F.relu(self.convs[ci](x))
                ci += 1
            elif item == "M":
                x = F.max_pool2d(x, self.pool[0], self.pool[1])
            else:
                out.append(x)
        return out

    @torch.no_grad()
    def forward(self, x: torch.Tensor, y: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        """LPIPS distance of two image batches [N, 3, H, W] (or [3, H, W]) -> [N, 1, 1, 1].  ``normalize=True``: the images are in
        [0, 1] and are mapped to [-1, 1] first (how the reference calls it)."""
        if not self.ready:
            raise RuntimeError("LPIPS needs pretrained weights: pass weights_path= (see gsvc_amd/lpips.py); they are not in this image")
        if x.dim() == 3:
            x, y = x.unsqueeze(0), y.unsqueeze(0)
        x, y = x.float(), y.float()
        if normalize:
            x, y = 2 * x - 1, 2 * y - 1
        fx = self.features((x - self.shift) / self.scale)
        fy = self.features((y - self.shift) / self.scale)
        total = 0
        for a, b, w in zip(fx, fy, self.lins):
            a = a / (torch.sqrt(torch.sum(a * a, dim=1, keepdim=True)) + 1e-10)
            b = b / (torch.sqrt(torch.sum(b * b, dim=1, keepdim=True)) + 1e-10)
            total = total + ((a - b) ** 2 * w).sum(dim=1, keepdim=True).mean(dim=(2, 3), keepdim=True)
        return total


def lpips_fn_from(weights_path: str, net: str = "alex", lin_weights_path: str | None = None, device="cuda"):
    """``lpips_fn`` of reference utils/metric_utils.py:44 (``lpips.LPIPS().cuda()``), from a weights file."""
    return LPIPS(net, weights_path, lin_weights_path).to(device)
