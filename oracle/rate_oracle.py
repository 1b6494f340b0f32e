"""numpy restatement of the entropy-rate estimator and quantisers — TEST INFRASTRUCTURE ONLY.

Follows reference utils/entropy_models.py:32-68 (EntropyGaussian.forward, non-quantized branch),
:159-175 (Low_bound) and utils/encodings.py:395-449 (STE_multistep, UniformQuantizer clamp).  Pinned by the
golden vectors tests/golden/rate_*.npz generated from the reference's own Python (make_golden.py).
Computed in float64; the fp32 product is compared within the tolerance stated in the tests.
"""
import numpy as np
from scipy.special import erf

LOW_BOUND = 2.0 ** -16
CLAMP_STEPS = 15_000


def _cdf(v, mu, sigma):
    # torch.distributions.Normal.cdf
    return 0.5 * (1.0 + erf((v - mu) / sigma / np.sqrt(2.0)))


def _pdf(v, mu, sigma):
    z = (v - mu) / sigma
    return np.exp(-0.5 * z * z) / (sigma * np.sqrt(2.0 * np.pi))


def entropy_gaussian_bits(x, mean, scale, Q, x_mean=None):
    """bits and the clamped x.  Q: python scalar or array broadcastable to x (e.g. [n,1])."""
    x = np.asarray(x, np.float64)
    mean = np.asarray(mean, np.float64)
    scale = np.asarray(scale, np.float64)
    Q = np.asarray(Q, np.float64)
    if x_mean is None:
        x_mean = x.mean()
    lo = x_mean - CLAMP_STEPS * Q.mean()
    hi = x_mean + CLAMP_STEPS * Q.mean()
    xc = np.clip(x, lo, hi)
    lik = _cdf(xc + 0.5 * Q, mean, scale) - _cdf(xc - 0.5 * Q, mean, scale)
    bits = -np.log2(np.maximum(lik, LOW_BOUND))
    return bits, xc, lik, (lo, hi)


def entropy_gaussian_grads(x, mean, scale, Q, g, x_mean=None):
    """VJP of bits w.r.t. (x, mean, scale, Q) for upstream gradient g (same shape as x).  dQ has Q's shape.
    Low_bound: gradient passes only where the un-clamped likelihood >= 2^-16."""
    x = np.asarray(x, np.float64)
    g = np.asarray(g, np.float64)
    Qa = np.asarray(Q, np.float64)
    bits, xc, lik, (lo, hi) = entropy_gaussian_bits(x, mean, scale, Q, x_mean)
    mean = np.asarray(mean, np.float64)
    scale = np.asarray(scale, np.float64)
    pu = _pdf(xc + 0.5 * Qa, mean, scale)
    pl = _pdf(xc - 0.5 * Qa, mean, scale)
    zu = (xc + 0.5 * Qa - mean) / scale
    zl = (xc - 0.5 * Qa - mean) / scale
    dl = np.where(lik >= LOW_BOUND, -1.0 / (np.maximum(lik, LOW_BOUND) * np.log(2.0)), 0.0) * g
    dx = np.where((x >= lo) & (x <= hi), dl * (pu - pl), 0.0)
    dmean = -dl * (pu - pl)
    dscale = -dl * (zu * pu - zl * pl)
    dQ_full = dl * 0.5 * (pu + pl)
    if Qa.ndim == 0 or Qa.size == 1:
        dQ = dQ_full.sum().reshape(Qa.shape)
    else:
        axes = tuple(i for i, (a, b) in enumerate(zip(np.broadcast_shapes(Qa.shape, x.shape), Qa.shape)) if b == 1 and a != 1)
        dQ = dQ_full.sum(axis=axes, keepdims=True) if axes else dQ_full
    return dx, dmean, dscale, dQ


def ste_multistep(x, Q, input_mean=None):
    x = np.asarray(x, np.float64)
    Q = np.asarray(Q, np.float64)
    if input_mean is None:
        input_mean = x.mean()
    lo = int(input_mean / Q.mean() - CLAMP_STEPS)
    hi = int(input_mean / Q.mean() + CLAMP_STEPS)
    xq = np.clip(x / Q, lo, hi) * Q
    return np.round(xq / Q) * Q
