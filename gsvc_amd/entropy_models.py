"""Entropy-rate estimator of the GSVC hot path (host side of csrc/rate.hip).

Mirrors reference utils/entropy_models.py:32-68 (``EntropyGaussian``) and :159-175 (``Low_bound``):

    bits = -log2( max( Phi((x + Q/2 - mu)/sigma) - Phi((x - Q/2 - mu)/sigma), 2^-16 ) )

with x first clamped to ``x_mean -+ 15000 * mean(Q)``.  The whole chain (clamp, two normal CDFs, subtract,
lower bound, -log2) and its analytic backward — including the net Low_bound rule "gradient passes only where
the likelihood is >= 2^-16", which the reference evaluates through a NumPy round trip on the host — run as
one fused HIP kernel each way.  ``quantized=True`` (bit accounting of quantised symbols) scales
the model to symbol units and runs through the same kernel without a clamp.  The never-instantiated
Entropy_gaussian_clamp / Entropy_bernoulli / Entropy_factorized / UniverseQuant are out of scope.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib

CLAMP_STEPS = 15_000


class Low_bound(torch.autograd.Function):
    """clamp(min=2^-16) whose gradient passes only where the input was >= 2^-16 (net effect of the reference
    rule: it zeroes g where x < min before applying the (x >= min) | (g < 0) mask)."""

    min_val = 2 ** -16

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.clamp(x, min=Low_bound.min_val)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * (x >= Low_bound.min_val).to(g.dtype)


class _GaussianBits(torch.autograd.Function):
    """bits[n,c] through csrc/rate.hip; Q is a per-row [n] tensor or None (then Q_scalar)."""

    @staticmethod
    def forward(ctx, x, mean, scale, Q_rows, Q_scalar, x_lo, x_hi, per_row=False):
        x, mean, scale = x.contiguous(), mean.contiguous(), scale.contiguous()
        n, c = x.shape
        bits = torch.empty_like(x)
        q = Q_rows.contiguous() if Q_rows is not None else None
        _lib.check(_lib.lib().gsvc_rate_forward(_lib.ptr(x), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(q), float(Q_scalar),
                                                None, _lib.ptr(x_lo), _lib.ptr(x_hi), int(per_row), n, c, _lib.ptr(bits), None,
                                                _lib.current_stream(x.device)), "gsvc_rate_forward")
        ctx.save_for_backward(x, mean, scale, q, x_lo, x_hi)
        ctx.Q_scalar = float(Q_scalar)
        ctx.per_row = bool(per_row)
        return bits

    @staticmethod
    def backward(ctx, g):
        x, mean, scale, q, x_lo, x_hi = ctx.saved_tensors
        n, c = x.shape
        g = g.contiguous()
        dx, dmean, dscale = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        dQ = torch.zeros(n, device=x.device, dtype=x.dtype) if q is not None else None
        # the kernel multiplies by `weight`; feeding dL/dbits as the weight gives the vector-Jacobian product
        _lib.check(_lib.lib().gsvc_rate_backward(_lib.ptr(x), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(q), ctx.Q_scalar,
                                                 _lib.ptr(g), _lib.ptr(x_lo), _lib.ptr(x_hi), int(ctx.per_row), n, c, None,
                                                 _lib.ptr(dx),
                                                 _lib.ptr(dmean), _lib.ptr(dscale), _lib.ptr(dQ), None,
                                                 _lib.current_stream(x.device)), "gsvc_rate_backward")
        return dx, dmean, dscale, dQ, None, None, None, None


class EntropyGaussian(nn.Module):
    def __init__(self, Q=1):
        super().__init__()
        self.Q = Q

    def forward(self, x, mean, scale, Q=None, x_mean=None, quantized=False, row_bounds=None):
        """``row_bounds=(lo[n], hi[n])`` replaces the clamp bounds computed from x_mean / mean(Q) by one pair per
        row (used when several renders are batched into one call; each row carries its own render's bounds)."""
        if Q is None:
            Q = self.Q
        if quantized:
            # bit accounting of already quantised symbols (reference utils/entropy_models.py:56-59): the model is scaled
            # to symbol units, N(mean / Q, scale / Q) over [x - 1/2, x + 1/2], and nothing is clamped
            if row_bounds is not None:
                raise ValueError("quantized=True takes no clamp bounds")
            inf = torch.full((1,), float("inf"), device=x.device)
            return self.forward(x, mean / Q, scale / Q, Q=1.0, row_bounds=(-inf.expand(x.reshape(-1, x.shape[-1]).shape[0]),
                                                                           inf.expand(x.reshape(-1, x.shape[-1]).shape[0])))
        if not x.is_cuda:
            raise _lib.GsvcError("EntropyGaussian runs on the HIP kernels of csrc/rate.hip; CPU tensors are not supported")
        shape = x.shape
        c = shape[-1]
        x2 = x.reshape(-1, c)
        mean2 = mean.expand(shape).reshape(-1, c)
        scale2 = scale.expand(shape).reshape(-1, c)
        if x_mean is None:
            x_mean = x.mean()
        if isinstance(Q, torch.Tensor):
            q_mean = Q.mean()
            if Q.numel() == 1:
                q_rows, q_scalar = Q.reshape(1).expand(x2.shape[0]), 0.0
            elif Q.shape[-1] == 1 and Q.numel() == x2.shape[0]:
                q_rows, q_scalar = Q.reshape(-1), 0.0
            else:
                raise NotImplementedError("Q must be a scalar or one step per row ([n,1])")
        else:
            q_mean = torch.ones(1, device=x.device) * Q
            q_rows, q_scalar = None, float(Q)
        if row_bounds is not None:
            lo = row_bounds[0].detach().reshape(-1).float().contiguous()
            hi = row_bounds[1].detach().reshape(-1).float().contiguous()
        else:
            lo = (x_mean - CLAMP_STEPS * q_mean).detach().reshape(1).float().contiguous()
            hi = (x_mean + CLAMP_STEPS * q_mean).detach().reshape(1).float().contiguous()
        bits = _GaussianBits.apply(x2, mean2, scale2, q_rows, q_scalar, lo, hi, row_bounds is not None)
        return bits.view(shape)
