"""gsvc_amd.lpips (reference utils/metric_utils.py:44, utils/report_utils.py:154, lpipsPyTorch/): the arithmetic of the
metric on RANDOM weights against an independent functional statement of the published algorithm — a torchvision-shaped ``features``
Sequential tapped at the published layer numbers — and the two weight-file layouts.  The pretrained numbers are not in this image:
what is pinned here is the structure, not the metric's calibration."""
import os

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F


def _tv_features(net):
    """torchvision's ``alexnet().features`` / ``vgg16().features`` layer for layer (untrained)."""
    if net == "alex":
        return nn.Sequential(nn.Conv2d(3, 64, 11, 4, 2), nn.ReLU(), nn.MaxPool2d(3, 2), nn.Conv2d(64, 192, 5, padding=2), nn.ReLU(), nn.MaxPool2d(3, 2),
                             nn.Conv2d(192, 384, 3, padding=1), nn.ReLU(), nn.Conv2d(384, 256, 3, padding=1), nn.ReLU(),
                             nn.Conv2d(256, 256, 3, padding=1), nn.ReLU(), nn.MaxPool2d(3, 2))
    layers, c = [], 3
    for v in (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"):
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU()]
            c = v
    return nn.Sequential(*layers)


def _statement(features, lins, taps, x, y):
    """Zhang et al. 2018 as the lpipsPyTorch modules state it: z-score, features up to each tap (1-based layer numbers), unit
    normalisation over channels, squared difference, 1x1 convolution without bias, spatial mean, sum over taps."""
    mean = torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1)
    std = torch.tensor([.458, .448, .450]).view(1, 3, 1, 1)

    def feats(t):
        t = (t - mean) / std
        out = []
        for i, layer in enumerate(features, 1):
            t = layer(t)
            if i in taps:
                out.append(t / (torch.sqrt(torch.sum(t ** 2, dim=1, keepdim=True)) + 1e-10))
        return out
    res = [F.conv2d((a - b) ** 2, w).mean((2, 3), True) for a, b, w in zip(feats(x), feats(y), lins)]
    return torch.stack(res, 0).sum(0)


@pytest.mark.parametrize("net", ["alex", "vgg"])
def test_lpips_structure_matches_the_published_algorithm(net, tmp_path):
    from gsvc_amd.lpips import LPIPS
    torch.manual_seed(3)
    feats = _tv_features(net)
    taps = {"alex": [2, 5, 8, 10, 12], "vgg": [4, 9, 16, 23, 30]}[net]
    chans = {"alex": [64, 192, 384, 256, 256], "vgg": [64, 128, 256, 512, 512]}[net]
    lins = [torch.rand(1, c, 1, 1) / c for c in chans]
    # the two-file layout: torchvision's backbone names + the lpips package's head names
    torch.save({"features." + k: v for k, v in feats.state_dict().items()}, tmp_path / "backbone.pth")
    torch.save({f"lin{k}.model.1.weight": w for k, w in enumerate(lins)}, tmp_path / "lins.pth")
    m = LPIPS(net, str(tmp_path / "backbone.pth"), str(tmp_path / "lins.pth"))
    x, y = torch.rand(2, 3, 96, 128), torch.rand(2, 3, 96, 128)
    with torch.no_grad():
        want = _statement(feats, lins, taps, 2 * x - 1, 2 * y - 1)
    got = m(x, y, normalize=True)
    assert got.shape == (2, 1, 1, 1)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-7), (got.flatten(), want.flatten())
    assert float(m(x, x, normalize=True).abs().max()) == 0.0
    assert torch.allclose(m(x, y), m(y, x))
    assert torch.allclose(m(x[0], y[0], normalize=True), got[:1])            # a [3, H, W] image is a batch of one
    # the one-file layout of the lpips package (slices named by torchvision's layer numbers)
    sl = {"alex": [1, 2, 3, 4, 5], "vgg": [1, 1, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5]}[net]
    idx = [i for i, l in enumerate(feats) if isinstance(l, nn.Conv2d)]
    one = {f"net.slice{s}.{i}.{part}": getattr(feats[i], part).detach() for s, i in zip(sl, idx) for part in ("weight", "bias")}
    one.update({f"lin{k}.model.1.weight": w for k, w in enumerate(lins)})
    torch.save(one, tmp_path / "one.pth")
    assert torch.equal(LPIPS(net, str(tmp_path / "one.pth"))(x, y, normalize=True), got)


def test_lpips_refuses_to_run_without_weights():
    from gsvc_amd.lpips import LPIPS
    m = LPIPS("alex")
    with pytest.raises(RuntimeError, match="pretrained weights"):
        m(torch.rand(3, 64, 64), torch.rand(3, 64, 64))
    r = LPIPS("alex", random_init=True)
    v = r(torch.rand(3, 64, 64), torch.rand(3, 64, 64), normalize=True)
    assert v.shape == (1, 1, 1, 1) and float(v) > 0
    assert all(not p.requires_grad for p in r.parameters())
