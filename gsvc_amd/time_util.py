"""Sin/cos positional embedding of the frame time and the anchor's z offset.

Same function as reference utils/time_util.py:7-55 with the arguments GSVC uses (``get_embedder(16, 1)``:
include_input, log-sampled frequencies 2^0..2^15, [sin, cos] per frequency -> 33 outputs per input dim),
written as one broadcasted tensor expression instead of 33 lambdas and a cat of 33 pieces.
"""
from __future__ import annotations

import torch


class Embedder:
    def __init__(self, input_dims: int = 1, num_freqs: int = 16, max_freq_log2: float | None = None,
                 include_input: bool = True, log_sampling: bool = True):
        if max_freq_log2 is None:
            max_freq_log2 = num_freqs - 1
        if log_sampling:
            self.freq_bands = 2.0 ** torch.linspace(0.0, max_freq_log2, steps=num_freqs)
        else:
            self.freq_bands = torch.linspace(2.0 ** 0.0, 2.0 ** max_freq_log2, steps=num_freqs)
        self.include_input = include_input
        self.input_dims = input_dims
        self.out_dim = input_dims * (2 * num_freqs + (1 if include_input else 0))

    def embed(self, x: torch.Tensor) -> torch.Tensor:
        """[..., d] -> [..., d*(2F+1)] ordered [x, sin(f0 x), cos(f0 x), sin(f1 x), cos(f1 x), ...]."""
        f = self.freq_bands.to(device=x.device, dtype=x.dtype)
        xf = x.unsqueeze(-2) * f.view(-1, 1)                       # [..., F, d]
        sc = torch.stack([torch.sin(xf), torch.cos(xf)], dim=-2)   # [..., F, 2, d]
        sc = sc.reshape(*x.shape[:-1], 2 * f.numel() * x.shape[-1])      # (explicit: -1 is ambiguous for an empty batch)
        return torch.cat([x, sc], dim=-1) if self.include_input else sc


def get_embedder(multires: int, i: int = 1):
    if i == -1:
        return torch.nn.Identity(), 3
    e = Embedder(input_dims=i, num_freqs=multires, max_freq_log2=multires - 1)
    return e.embed, e.out_dim
