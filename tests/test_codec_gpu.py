"""Entropy coder of the quantised attributes (csrc/ans.hip): exact round trips, coded size against the
model's own entropy, edge cases of the symbol range and of the model."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ideal_bits(sym, mu, sigma, smin, smax):
    """-log2 of the coder's own discretised model, in float64 on the host."""
    from scipy.special import erfc
    s, m, sg = sym.double().cpu().numpy(), mu.double().cpu().numpy(), sigma.double().cpu().numpy()
    R = smax - smin + 1
    M = 1 << 20

    def C(v):
        p = 0.5 * erfc(-((v - 0.5 - m) / sg) * 0.7071067811865476)
        c = np.floor(np.clip(p, 0, 1) * (M - R)) + (v - smin)
        c = np.where(v <= smin, 0.0, c)
        return np.where(v > smax, float(M), c)

    f = C(s + 1) - C(s)
    return float(-np.log2(f / M).sum())


@pytest.mark.parametrize("n,spread", [(1, 3.0), (1000, 0.4), (4096, 2.0), (4097, 8.0), (300_000, 3.0)])
def test_round_trip_and_size(n, spread):
    from gsvc_amd.codec import ans_decode, ans_encode
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(100 + n)
    mu = torch.randn(n, device=dev, generator=g) * 5.0
    sigma = torch.rand(n, device=dev, generator=g) * spread + 0.05
    sym = torch.round(mu + sigma * torch.randn(n, device=dev, generator=g)).to(torch.int32)
    smin, smax = int(sym.min()), int(sym.max())
    if smin == smax:
        smax += 1
    stream = ans_encode(sym, mu, sigma, smin, smax)
    back = ans_decode(stream, mu, sigma)
    assert torch.equal(back, sym)
    ideal = _ideal_bits(sym, mu, sigma, smin, smax)
    from gsvc_amd.codec import SEG_LEN
    n_seg = (n + SEG_LEN - 1) // SEG_LEN
    overhead = 8 * (32 + 4 * n_seg + 5 * n_seg)         # header, size table, final state + flush per segment
    assert 8 * len(stream) <= ideal * 1.002 + overhead + 64, (8 * len(stream), ideal)
    assert 8 * len(stream) >= ideal * 0.999


def test_model_edge_cases():
    """Symbols the model calls (almost) impossible, sigma at the 1e-9 clamp, symbols pinned to the range ends."""
    from gsvc_amd.codec import ans_decode, ans_encode
    dev = torch.device("cuda")
    n = 5000
    g = torch.Generator(device=dev).manual_seed(7)
    mu = torch.randn(n, device=dev, generator=g) * 50.0
    sigma = torch.full((n,), 1e-9, device=dev)
    sigma[::3] = 1e4
    sym = torch.randint(-300, 301, (n,), device=dev, generator=g, dtype=torch.int32)     # unrelated to the model
    sym[0], sym[1] = -300, 300
    stream = ans_encode(sym, mu, sigma, -300, 300)
    assert torch.equal(ans_decode(stream, mu, sigma), sym)
    # a range as wide as GSVC's +-15000 symbol clamp
    wide = torch.randint(-15000, 15001, (4096,), device=dev, generator=g, dtype=torch.int32)
    m2, s2 = torch.zeros(4096, device=dev), torch.full((4096,), 3000.0, device=dev)
    assert torch.equal(ans_decode(ans_encode(wide, m2, s2, -15000, 15000), m2, s2), wide)


def test_symbol_outside_range_is_an_error():
    from gsvc_amd import _lib
    from gsvc_amd.codec import ans_encode
    dev = torch.device("cuda")
    sym = torch.tensor([0, 5, 11], device=dev, dtype=torch.int32)
    with pytest.raises(_lib.GsvcError):
        ans_encode(sym, torch.zeros(3, device=dev), torch.ones(3, device=dev), 0, 10)


def test_encoder_decoder_gaussian_interface():
    """The reference-shaped pair: quantised features with a per-row step, de-quantised values back; the coded size
    agrees with EntropyGaussian(quantized=True) — the estimate estimate_final_bits is made of."""
    from gsvc_amd.codec import decoder_gaussian, encoder_gaussian
    from gsvc_amd.entropy_models import EntropyGaussian
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(3)
    rows, C = 20_000, 50
    mean = torch.randn(rows, C, device=dev, generator=g)
    scale = torch.rand(rows, C, device=dev, generator=g) * 2 + 0.05
    Q = (torch.rand(rows, 1, device=dev, generator=g) * 0.5 + 0.75).repeat(1, C)
    value = mean + scale * torch.randn(rows, C, device=dev, generator=g)
    x = torch.round(value / Q)
    bits, lo, hi, stream = encoder_gaussian(x, mean, scale, Q, -15000, 15000)
    back = decoder_gaussian(mean, scale, Q, stream=stream)
    assert torch.equal(back, x * Q)
    est = EntropyGaussian()(x, mean, scale, Q[:, :1], quantized=True).sum().item()
    assert abs(bits - est) <= 0.015 * est, (bits, est)
