"""Quantisers and the hash-grid encoder module of the GSVC hot path (host side).

Mirrors the names and numerics of reference utils/encodings.py:34-51 (get_binary_vxl_size), :375-392
(STE_binary), :395-432 (STE_multistep), :434-449 (UniformQuantizer), :452-482 (Quantize_anchor), :485-618
(_grid_encode autograd function) and :621-709 (GridEncoder).  The codec half of that file (ANS / torchac /
G-PCC glue) is out of scope.

The grid lookups run in the HIP kernels of csrc/grid.hip through gridencoder_backend (no CPU path).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

ANCHOR_ROUND_DIGITS = 16
Q_ANCHOR = 1.0 / (2 ** ANCHOR_ROUND_DIGITS - 1)
CLAMP_STEPS = 15_000  # symbols are clamped to mean/Q -+ 15000 steps


def get_binary_vxl_size(binary_vxl: torch.Tensor):
    """Bernoulli code length of a {0,1} table: n1*(-log2 p) + n0*(-log2 (1-p)) + 32 bits for p itself.
    Returns (p, bits (tensor, differentiable), megabytes (python float), element count)."""
    total = binary_vxl.numel()
    ones = binary_vxl.sum()
    zeros = total - ones
    p = torch.clamp(ones / total, min=1e-6, max=1 - 1e-6)
    bits = ones * (-torch.log2(p)) + zeros * (-torch.log2(1 - p))
    bits = bits + 32
    return p, bits, bits.item() / 8.0 / 1024 / 1024, total


class STE_binary(torch.autograd.Function):
    """sign() to {-1,+1} (0 -> +1) with a straight-through gradient inside [-1, 1]."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.where(x >= 0, 1.0, -1.0).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * (x.abs() <= 1)


class STE_binary_counted(torch.autograd.Function):
    """STE_binary whose forward also returns the number of +1 entries (csrc/quant.hip k_ste_binary_count): the hash-bit term of
    the fitting loss needs that count of every table it binarises."""

    @staticmethod
    def forward(ctx, x):
        from . import _lib
        xc = x.contiguous()
        y = torch.empty_like(xc)
        count = torch.empty(1, device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().gsvc_ste_binary_count(_lib.ptr(xc), xc.numel(), _lib.ptr(y), _lib.ptr(count),
                                                    _lib.current_stream(x.device)), "gsvc_ste_binary_count")
        ctx.save_for_backward(x)
        ctx.mark_non_differentiable(count)
        return y, count

    @staticmethod
    def backward(ctx, g, _g_count):
        (x,) = ctx.saved_tensors
        return g * (x.abs() <= 1)


class STE_binary_tables(torch.autograd.Function):
    """STE_binary_counted for several tables at once (csrc/quant.hip k_ste_binary_count_many): returns the binarised tables and
    ``counts`` [T], the +1 entries of each.  ``counts`` is differentiable — a table entry's share of its count is 1/2 — so a
    loss term written on the counts (the hash-bit term) reaches the tables inside this node's one backward launch."""

    @staticmethod
    def forward(ctx, *xs):
        import ctypes as C
        from . import _lib
        xs = [x.contiguous() for x in xs]
        T, dev = len(xs), xs[0].device
        sizes = [x.numel() for x in xs]
        flat = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
        ys = [v.view(x.shape) for v, x in zip(flat.split(sizes), xs)]
        counts = torch.empty(T, device=dev, dtype=torch.float32)
        ptrs = lambda ts: (C.c_void_p * T)(*[t.data_ptr() for t in ts])
        _lib.check(_lib.lib().gsvc_ste_binary_count_many(ptrs(xs), ptrs(ys), (C.c_int64 * T)(*sizes), T, _lib.ptr(counts),
                                                         _lib.current_stream(dev)), "gsvc_ste_binary_count_many")
        ctx.save_for_backward(*xs)
        ctx.set_materialize_grads(False)
        return (*ys, counts)

    @staticmethod
    def backward(ctx, *gs):
        import ctypes as C
        from . import _lib
        xs = ctx.saved_tensors
        T, dev = len(xs), xs[0].device
        g_ys = [None if g is None else g.contiguous() for g in gs[:T]]
        g_counts = gs[T]
        if g_counts is not None and g_counts.stride(0) not in (0, 1):
            g_counts = g_counts.contiguous()
        sizes = [x.numel() for x in xs]
        flat = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
        outs = [v.view(x.shape) for v, x in zip(flat.split(sizes), xs)]
        ptrs = lambda ts: (C.c_void_p * T)(*[None if t is None else t.data_ptr() for t in ts])
        _lib.check(_lib.lib().gsvc_ste_binary_backward_many(ptrs(xs), ptrs(g_ys), (C.c_int64 * T)(*sizes), T, _lib.ptr(g_counts),
                                                            0 if g_counts is None else g_counts.stride(0), ptrs(outs), _lib.current_stream(dev)), "gsvc_ste_binary_backward_many")
        return tuple(outs)


class CountBits(torch.autograd.Function):
    """Bernoulli code length of {-1, +1} tables from the counts of their +1 entries: bits = n1 (-log2 p) + n0 (-log2 (1 - p)) + 32,
    p = n1 / total clamped to [1e-6, 1 - 1e-6] (reference utils/encodings.py:34-51 over the concatenated tables,
    pipeline/train.py:456).  One launch each way (csrc/quant.hip k_table_bits)."""

    @staticmethod
    def forward(ctx, counts, total):
        from . import _lib
        counts = counts.contiguous()
        out = torch.empty(2, device=counts.device, dtype=torch.float32)
        _lib.check(_lib.lib().gsvc_table_bits(_lib.ptr(counts), counts.numel(), int(total), _lib.ptr(out),
                                              _lib.current_stream(counts.device)), "gsvc_table_bits")
        ctx.save_for_backward(out)
        ctx.T = counts.numel()
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        return (g * out[1]).expand(ctx.T), None


def binarized_tables(grids):
    """``embeddings()`` of several grids.  Inside a step scope (``step_cache``) the STE-binary tables that have not been
    binarised yet go through ONE launch (STE_binary_tables); the first grid's cache keeps the whole count vector for the
    hash-bit term."""
    caches = [getattr(g, "step_cache", None) for g in grids]
    if (1 < len(grids) <= 8 and all(c is not None and "emb" not in c for c in caches)
            and all(g.ste_binary and g.params.is_cuda and g.params.dtype == torch.float32 and g.params.numel() < (1 << 24) for g in grids)):
        *ys, counts = STE_binary_tables.apply(*[g.params for g in grids])
        for c, y in zip(caches, ys):
            c["emb"] = y
        caches[0]["counts_all"] = (counts, tuple(id(g) for g in grids))
    return [g.embeddings() for g in grids]


def _symbol_bounds(mean, Q):
    centre = mean / Q.mean().detach()
    return centre - CLAMP_STEPS, centre + CLAMP_STEPS


class STE_multistep(torch.autograd.Function):
    """Deterministic round-to-step with identity gradient; symbols clamped to int(mean/Q_mean -+ 15000)."""

    @staticmethod
    def forward(ctx, x, Q, input_mean=None):
        if input_mean is None:
            input_mean = x.mean()
        if not isinstance(Q, torch.Tensor):
            Q = torch.ones(1, device=x.device) * Q
        lo, hi = _symbol_bounds(input_mean, Q)
        x = torch.clamp(x / Q, min=int(lo.detach()), max=int(hi.detach())) * Q
        return torch.round(x / Q) * Q

    @staticmethod
    def backward(ctx, g):
        return g, None, None

    @staticmethod
    def quantize(x, Q, min_value, max_value):
        return torch.clamp(torch.round(x / Q), min=min_value, max=max_value)


class UniformQuantizer(nn.Module):
    """Training-time quantisation proxy: clamp to the symbol range, add U(-Q/2, Q/2) noise."""

    def forward(self, x, Q, input_mean=None):
        if input_mean is None:
            input_mean = x.mean()
        if not isinstance(Q, torch.Tensor):
            Q = torch.ones(1, device=x.device) * Q
        lo, hi = _symbol_bounds(input_mean, Q)
        x = torch.clamp(x / Q, min=lo.detach(), max=hi.detach()) * Q
        return x + torch.empty_like(x).uniform_(-0.5, 0.5) * Q


class Quantize_anchor(torch.autograd.Function):
    """Snap anchors to a 16-bit grid between the cube bounds (identity gradient)."""

    @staticmethod
    def _grid(anchors, min_v, max_v):
        interval = (max_v - min_v) * Q_ANCHOR + 1e-6
        q = torch.div(anchors - min_v, interval, rounding_mode="floor")
        return torch.clamp(q, 0, 2 ** ANCHOR_ROUND_DIGITS - 1), interval

    @staticmethod
    def forward(ctx, anchors, min_v, max_v):
        q, interval = Quantize_anchor._grid(anchors, min_v, max_v)
        return q * interval + min_v, q

    @staticmethod
    def backward(ctx, g, _g_q):
        return g, None, None

    @classmethod
    def quantized(cls, anchors, min_v, max_v):
        q, interval = cls._grid(anchors, min_v, max_v)
        return q, interval, min_v

    @classmethod
    def dequantized(cls, anchors_q, interval, min_v):
        return anchors_q * interval + min_v


class _grid_encode(torch.autograd.Function):
    """[N,D] points x [rows,C] table -> [N, L*C]; native layout of the kernels is [L,N,C]."""

    @staticmethod
    def forward(ctx, inputs, embeddings, offsets_list, resolutions_list, calc_grad_inputs=False, min_level_id=None,
                n_levels_calc=1, binary_vxl=None, PV=0):
        from . import gridencoder_backend as backend
        if binary_vxl is not None or not isinstance(min_level_id, int):
            raise NotImplementedError("binary_vxl / per-point min_level_id are not used by GSVC")
        inputs = inputs.contiguous()
        embeddings = embeddings.contiguous()
        N, D = inputs.shape
        C = embeddings.shape[1]
        hi = min_level_id + n_levels_calc
        offs = offsets_list[min_level_id:hi + 1].contiguous()
        ress = resolutions_list[min_level_id:hi].contiguous()
        out = torch.empty(n_levels_calc, N, C, device=inputs.device, dtype=embeddings.dtype)
        dy_dx = torch.empty(N, n_levels_calc * D * C, device=inputs.device, dtype=embeddings.dtype) if calc_grad_inputs else None
        backend.grid_encode_forward(inputs, embeddings, offs, ress, out, N, D, C, n_levels_calc, 0, 128, PV, dy_dx, None, None)
        ctx.save_for_backward(inputs, embeddings, offs, ress, dy_dx)
        ctx.dims = (N, D, C, n_levels_calc)
        return out.permute(1, 0, 2).reshape(N, n_levels_calc * C)

    @staticmethod
    def backward(ctx, grad):
        from . import gridencoder_backend as backend
        inputs, embeddings, offs, ress, dy_dx = ctx.saved_tensors
        N, D, C, Lc = ctx.dims
        grad = grad.view(N, Lc, C).permute(1, 0, 2).contiguous()
        g_emb = torch.zeros_like(embeddings)
        g_in = torch.zeros_like(inputs) if dy_dx is not None else None
        backend.grid_encode_backward(grad, inputs, embeddings, offs, ress, g_emb, N, D, C, Lc, 0, 128, dy_dx, g_in, None, None)
        return g_in, g_emb, None, None, None, None, None, None, None


grid_encode = _grid_encode.apply


def level_offsets(resolutions, num_dim: int, log2_hashmap_size: int):
    """Row offset of every level: min(2^log2, res^D) rounded up to a multiple of 8 (reference :655-663)."""
    cap = 2 ** log2_hashmap_size
    offs, o = [], 0
    for r in resolutions:
        rows = min(cap, int(r) ** num_dim)
        rows = int(np.ceil(rows / 8) * 8)
        offs.append(o)
        o += rows
    offs.append(o)
    return offs


class GridEncoder(nn.Module):
    def __init__(self, num_dim=3, n_features=2, resolutions_list=(16, 23, 32, 46, 64, 92, 128, 184, 256, 368, 512, 736),
                 log2_hashmap_size=19, ste_binary=True, ste_multistep=False, add_noise=False, Q=1):
        super().__init__()
        res = torch.tensor(resolutions_list).to(torch.int)
        self.num_dim, self.n_features = num_dim, n_features
        self.n_levels = res.numel()
        self.log2_hashmap_size = log2_hashmap_size
        self.output_dim = self.n_output_dims = self.n_levels * n_features
        self.ste_binary, self.ste_multistep, self.add_noise, self.Q = ste_binary, ste_multistep, add_noise, Q
        self.max_params = 2 ** log2_hashmap_size
        offs = level_offsets(res.tolist(), num_dim, log2_hashmap_size)
        self.register_buffer("offsets_list", torch.from_numpy(np.array(offs, dtype=np.int32)))
        self.register_buffer("resolutions_list", res)
        self.n_params = offs[-1] * n_features
        self.params = nn.Parameter(torch.empty(offs[-1], n_features))
        self.reset_parameters()

    def reset_parameters(self):
        self.params.data.uniform_(-1e-4, 1e-4)

    def embeddings(self, test_phase=False, outspace_params=None):
        p = nn.Parameter(outspace_params) if outspace_params is not None else self.params
        if self.ste_binary:
            # a fitting step reads the binarised table twice (the lookup and the hash-bit term of the loss): inside a step
            # scope (Trainer sets ``step_cache``) the second reader gets the first one's tensor and autograd node
            cache = getattr(self, "step_cache", None)
            if cache is not None and outspace_params is None:
                if "emb" not in cache:
                    if p.is_cuda and p.dtype == torch.float32 and p.numel() < (1 << 24):
                        cache["emb"], cache["ones"] = STE_binary_counted.apply(p)      # + the count the hash-bit term reads
                    else:
                        cache["emb"] = STE_binary.apply(p)
                return cache["emb"]
            return STE_binary.apply(p)
        if self.add_noise and not test_phase:
            return p + (torch.rand_like(p) - 0.5) * (1 / self.Q)
        if self.ste_multistep or (self.add_noise and test_phase):
            return STE_multistep.apply(p, self.Q)
        return p

    def forward(self, inputs, min_level_id=None, max_level_id=None, test_phase=False, outspace_params=None,
                binary_vxl=None, PV=0):
        lead = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.num_dim)
        emb = self.embeddings(test_phase, outspace_params)
        lo = 0 if min_level_id is None else max(min_level_id, 0)
        hi = self.n_levels if max_level_id is None else min(max_level_id, self.n_levels)
        out = grid_encode(inputs, emb, self.offsets_list, self.resolutions_list, inputs.requires_grad, lo, hi - lo,
                          binary_vxl, PV)
        return out.view(lead + [(hi - lo) * self.n_features])
