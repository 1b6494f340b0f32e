"""GaussianModel — the hot-path half of reference scene/gaussian_model.py.

What is here (reference line ranges in brackets):
  * EntropyContext, Mix3d2dEncoding [81-147], FiLM / GeneratorNet / EntropyParamsNet [150-232]
  * constructor wiring: 4 hash grids, two positional embedders, 3 generator MLPs, deform MLP, 3 entropy nets,
    quantiser and rate model [268-505]
  * getters get_anchor / get_scaling / get_mask / get_mask_anchor / get_rotation / get_*_mlp [641-700]
  * update_anchor_bound, calc_interp_feat [706-732], create_from_pcd (initialisation) [754-800]
  * training_setup / update_learning_rate: 15 Adam groups, eps 1e-15, exponential LR [833-1058, 1148-1154]
  * training_statis [1281-1314], calc_entropy_context [1569-1597], get_encoding_params [506-518]
Parameter / sub-module names equal the reference's, so a reference ``state_dict`` loads unchanged.
Densification lives in densify.py, the stream codec in stream_codec.py / mlp_codec.py, ply / checkpoint IO in io.py
(SURVEY.md section 8f); the methods here delegate.

Differences in mechanism, not in values: the device is a constructor argument instead of a hard-coded
"cuda"; ``get_anchor`` etc. stay properties but the renderer caches them per render instead of recomputing
per access; the 3-NN initial scale uses chunked torch.cdist instead of simple_knn.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn as nn

from .encodings import GridEncoder, Quantize_anchor, STE_binary, UniformQuantizer
from . import switches
from .entropy_models import EntropyGaussian
from .time_util import get_embedder


@dataclass
class EntropyContext:
    mean_feat: torch.Tensor
    scale_feat: torch.Tensor
    mean_scaling: torch.Tensor
    scale_scaling: torch.Tensor
    mean_offsets: torch.Tensor
    scale_offsets: torch.Tensor
    Q_feat_adj: torch.Tensor
    Q_scaling_adj: torch.Tensor
    Q_offsets_adj: torch.Tensor


class SampledEntropyContext:
    """The entropy context in the form the TRAINING rate needs it (reference ortho_gaussian_renderer/guassian.py:99-113,183-199):
    the quantisation-step adjustments of EVERY row — they scale the noise of every visible anchor — but the priors' mean / scale
    only on the rows the 5 % rate sample chose: ``calc_sampled_rate`` reads ``mean_* / scale_*`` at ``choose_idx`` and nowhere else,
    so the three ``dist_net``s (112 k of an EntropyParamsNet triple's 141 k multiply-adds per anchor) evaluated on the other rows
    produce numbers, and receive gradients, that are identically unused.  Exact, not an approximation: the same values and the
    same parameter gradients as the all-rows form (tests/test_train_gpu.py).

    ``q_raw``: the three quant_step networks' raw outputs [D, 1] (the adjustment exp(clamp(q, -10, 10)) is applied by the
    consumer's kernel: gsvc_q_rows_forward(raw = 1)); ``rows(idx)``: an EntropyContext whose mean / scale tensors hold row i =
    context row idx[i] (its Q_*_adj are None)."""

    MIN_ROWS = 256      # below this the module path (torch GEMMs) answers

    def __init__(self, pc, feature, q_raw):
        self.pc, self.feature, self.q_raw = pc, feature, q_raw

    @property
    def Q_feat_adj(self):
        return torch.exp(torch.clamp(self.q_raw[0], min=-10, max=10))

    @property
    def Q_scaling_adj(self):
        return torch.exp(torch.clamp(self.q_raw[1], min=-10, max=10))

    @property
    def Q_offsets_adj(self):
        return torch.exp(torch.clamp(self.q_raw[2], min=-10, max=10))

    def rows(self, idx) -> EntropyContext:
        from . import mlp
        pc = self.pc
        if switches.DETERMINISTIC and self.feature.is_cuda and self.feature.requires_grad:
            from .generate import IndexRows
            x = IndexRows.apply(self.feature, idx)      # (the sample names an anchor once per view that drew it: a fixed-order backward)
        else:
            x = self.feature.index_select(0, idx)
        nets = (pc.mlp_feature_enet, pc.mlp_scaling_enet, pc.mlp_offset_enet)
        chains = [list(net.dist_net)[0::2] for net in nets]
        if (x.is_cuda and x.shape[0] >= self.MIN_ROWS and all(isinstance(net.dist_net, GeluSequential) for net in nets)
                and all(mlp.usable(x, *c, min_rows=self.MIN_ROWS) for c in chains) and not switches.NO_MLP_CHAIN):
            raw = mlp.seq_gelu_many(x, chains)
        else:
            raw = [net.dist_net(x) for net in nets]
        out = []
        if x.is_cuda and not switches.NO_FUSED_CTX:
            # split + max(scale, 1e-9) as one launch each way per network (gsvc_ctx_post_*; its step output is computed of a dummy)
            q0 = torch.zeros(x.shape[0], 1, device=x.device, dtype=torch.float32)
            for p in raw:
                out += list(_CtxPost.apply(p, q0))[:2]
        else:
            for p in raw:
                half = p.shape[1] // 2
                mean, scale = p.split([half, p.shape[1] - half], dim=1)
                out += [mean, torch.clamp(scale, 1e-9)]
        return EntropyContext(*out, None, None, None)


@dataclass
class BitInfo:
    """Estimated size of the coded model in bits, per stream (reference scene/gaussian_model.py:55-65)."""
    bit_anchor: int
    bit_anchor_gpcc: int
    bit_feat: int
    bit_scaling: int
    bit_offsets: int
    bit_hash: int
    bit_masks: int
    bit_mlp: int
    bit_mlp_encoded: int


BIT2MB_SCALE = 8 * 1024 * 1024


def calc_symbol_min_max(x_mean, Q, bound=15000):
    """Symbol range of a quantised attribute: mean / mean step -+ 15000, truncated (reference :236-239)."""
    x_min = x_mean.mean() / Q.mean() - bound
    x_max = x_mean.mean() / Q.mean() + bound
    return int(x_min), int(x_max)


class Mix3d2dEncoding(nn.Module):
    """One 3-D hash grid on (x,y,z) plus three 2-D grids on (x,y), (x,z), (y,z); outputs concatenated."""

    def __init__(self, n_features, resolutions_list, log2_hashmap_size, resolutions_list_2D, log2_hashmap_size_2D,
                 ste_binary, ste_multistep, add_noise, Q):
        super().__init__()
        kw = dict(n_features=n_features, ste_binary=ste_binary, ste_multistep=ste_multistep, add_noise=add_noise, Q=Q)
        self.encoding_xyz = GridEncoder(num_dim=3, resolutions_list=resolutions_list, log2_hashmap_size=log2_hashmap_size, **kw)
        self.encoding_xy = GridEncoder(num_dim=2, resolutions_list=resolutions_list_2D, log2_hashmap_size=log2_hashmap_size_2D, **kw)
        self.encoding_xz = GridEncoder(num_dim=2, resolutions_list=resolutions_list_2D, log2_hashmap_size=log2_hashmap_size_2D, **kw)
        self.encoding_yz = GridEncoder(num_dim=2, resolutions_list=resolutions_list_2D, log2_hashmap_size=log2_hashmap_size_2D, **kw)
        self.output_dim = sum(e.output_dim for e in (self.encoding_xyz, self.encoding_xy, self.encoding_xz, self.encoding_yz))

    def forward(self, x):
        if (x.is_cuda and x.dim() == 2 and x.shape[1] == 3 and x.dtype == torch.float32 and not x.requires_grad
                and not switches.NO_FUSED_GRID
                and all(g.ste_binary for g in (self.encoding_xyz, self.encoding_xy, self.encoding_xz, self.encoding_yz))):
            # the four grids write their column blocks of the [N, 192] matrix and read the gradient from them (gsvc_grid_*_ex):
            # no coordinate slices, no [L, N, C] -> [N, L C] permutes, no cat — on either pass
            grids = (self.encoding_xyz, self.encoding_xy, self.encoding_xz, self.encoding_yz)
            if (not torch.is_grad_enabled() and all(g.n_features == 8 and g.params.is_contiguous() for g in grids)
                    and not switches.NO_PACKED_GRID):
                # inference (evaluation, the decoder): the tables as the bitstream carries them — one sign bit per entry, one byte
                # per row of 8 features — packed on the fly (one pass over the float tables) and looked up as bytes
                return _mix_grid_packed(x, grids)
            from .encodings import binarized_tables
            return _MixGridEncode.apply(x, grids, *binarized_tables(grids))
        xy, xz, yz = x[..., 0:2], x[..., 0::2], x[..., 1:3]     # slices, not list indices: their backward is a strided add, not a sort-based index_put
        return torch.cat([self.encoding_xyz(x), self.encoding_xy(xy), self.encoding_xz(xz), self.encoding_yz(yz)], dim=-1)


def _mix_grid_packed(x, grids):
    """Mix3d2dEncoding without autograd through bit-packed tables (csrc/grid.hip gsvc_pack_sign_bits + gsvc_grid_forward_packed):
    the same numbers as the lookup in the {-1, +1} float tables, from 1/32 of the table bytes."""
    import ctypes as C
    from . import _lib, switches
    x = x.contiguous()
    N = x.shape[0]
    total = sum(g.n_levels * g.n_features for g in grids)
    out = torch.empty(N, total, device=x.device, dtype=torch.float32)
    st, L = _lib.current_stream(x.device), _lib.lib()
    for g, (io, c0) in zip(grids, _MixGridEncode._layout(grids, x, total)):
        rows = g.params.shape[0]
        bits = torch.empty(rows, dtype=torch.uint8, device=x.device)
        _lib.check(L.gsvc_pack_sign_bits(_lib.ptr(g.params.detach()), rows, _lib.ptr(bits), st), "gsvc_pack_sign_bits")
        _lib.check(L.gsvc_grid_forward_packed(_lib.ptr(x), _lib.ptr(bits), _lib.ptr(g.offsets_list), _lib.ptr(g.resolutions_list),
                                              C.c_void_p(out.data_ptr() + 4 * c0), N, g.num_dim, g.n_levels, C.byref(io), st),
                   "gsvc_grid_forward_packed")
    return out


class _MixGridEncode(torch.autograd.Function):
    """Mix3d2dEncoding for positions that carry no gradient: grid k reads columns ``COLS[k]`` of x [N, 3] and owns columns
    [c0_k, c0_k + L_k C) of the output [N, sum L C] (level-major inside its block, as the concatenation of the per-grid
    [N, L C] outputs is)."""

    COLS = ((0, 1, 2), (0, 1), (0, 2), (1, 2))

    @staticmethod
    def _layout(grids, x, total):
        import ctypes as C
        from . import _lib
        out, c0 = [], 0
        for g, cols in zip(grids, _MixGridEncode.COLS):
            io = _lib.GridIOC()
            io.feat_level_stride, io.feat_point_stride, io.in_stride = g.n_features, total, x.shape[1]
            for k, c in enumerate(cols):
                io.in_col[k] = c
            out.append((io, c0))
            c0 += g.n_levels * g.n_features
        return out

    @staticmethod
    def _many(grids):
        return (not switches.NO_GRID_MANY and 1 <= len(grids) <= 4 and all(g.num_dim in (2, 3) for g in grids)
                and grids[0].n_features in (2, 4, 8) and all(g.n_features == grids[0].n_features for g in grids))

    @staticmethod
    def _jobs(grids, x, total, embs, feat_ptr, grad_embs):
        """gsvc_grid_many_job[len(grids)]: ``feat_ptr`` = the shared [N, total] matrix (outputs forward, gradient backward)."""
        from . import _lib
        jobs = (_lib.GridManyJobC * len(grids))()
        for k, (g, (io, c0)) in enumerate(zip(grids, _MixGridEncode._layout(grids, x, total))):
            j = jobs[k]
            j.embeddings = embs[k].data_ptr() if embs is not None else None
            j.features = feat_ptr + 4 * c0
            j.grad_embeddings = grad_embs[k].data_ptr() if grad_embs is not None else None
            j.offsets, j.resolutions = g.offsets_list.data_ptr(), g.resolutions_list.data_ptr()
            j.D, j.L, j.layout = g.num_dim, g.n_levels, io
        return jobs

    @staticmethod
    def forward(ctx, x, grids, *embs):
        import ctypes as C
        from . import _lib
        x = x.contiguous()
        N = x.shape[0]
        total = sum(g.n_levels * g.n_features for g in grids)
        out = torch.empty(N, total, device=x.device, dtype=torch.float32)
        embs = [e.contiguous() for e in embs]
        st = _lib.current_stream(x.device)
        if N == 0:                    # a view without a visible anchor: nothing to look up (and an empty tensor has no address to hand over)
            ctx.save_for_backward(x, *embs)
            ctx.grids, ctx.total = grids, total
            return out
        if _MixGridEncode._many(grids):
            # the four grids in ONE launch (blockIdx.y runs over their 12 + 3 x 4 levels)
            jobs = _MixGridEncode._jobs(grids, x, total, embs, out.data_ptr(), None)
            _lib.check(_lib.lib().gsvc_grid_forward_many(_lib.ptr(x), jobs, len(grids), N, grids[0].n_features, st), "gsvc_grid_forward_many")
        else:
            for g, e, (io, c0) in zip(grids, embs, _MixGridEncode._layout(grids, x, total)):
                _lib.check(_lib.lib().gsvc_grid_forward_ex(_lib.ptr(x), _lib.ptr(e), _lib.ptr(g.offsets_list), _lib.ptr(g.resolutions_list),
                                                           C.c_void_p(out.data_ptr() + 4 * c0), N, g.num_dim, g.n_features, g.n_levels,
                                                           C.byref(io), st), "gsvc_grid_forward_ex")
        ctx.save_for_backward(x, *embs)
        ctx.grids, ctx.total = grids, total
        return out

    @staticmethod
    def backward(ctx, grad):
        import ctypes as C
        from . import _lib
        x, *embs = ctx.saved_tensors
        grids, total = ctx.grids, ctx.total
        grad = grad.contiguous()
        N = x.shape[0]
        st = _lib.current_stream(x.device)
        outs = []
        # the tables' gradients: slices of one zeroed buffer (one fill, not one per grid)
        sizes = [e.numel() if ctx.needs_input_grad[2 + k] else 0 for k, e in enumerate(embs)]
        zeros = torch.zeros(sum(sizes), device=x.device, dtype=torch.float32).split(sizes)
        if N == 0:
            return (None, None, *[zeros[k].view(e.shape) if ctx.needs_input_grad[2 + k] else None for k, e in enumerate(embs)])
        if _MixGridEncode._many(grids) and all(ctx.needs_input_grad[2 + k] for k in range(len(grids))):
            ges = [zeros[k].view(e.shape) for k, e in enumerate(embs)]
            jobs = _MixGridEncode._jobs(grids, x, total, None, grad.data_ptr(), ges)
            _lib.check(_lib.lib().gsvc_grid_backward_many(_lib.ptr(x), jobs, len(grids), N, grids[0].n_features, st), "gsvc_grid_backward_many")
            return (None, None, *ges)
        for k, (g, e, (io, c0)) in enumerate(zip(grids, embs, _MixGridEncode._layout(grids, x, total))):
            if not ctx.needs_input_grad[2 + k]:
                outs.append(None)
                continue
            ge = zeros[k].view(e.shape)
            _lib.check(_lib.lib().gsvc_grid_backward_ex(C.c_void_p(grad.data_ptr() + 4 * c0), _lib.ptr(x), _lib.ptr(g.offsets_list),
                                                        _lib.ptr(g.resolutions_list), _lib.ptr(ge), N, g.num_dim, g.n_features,
                                                        g.n_levels, C.byref(io), st), "gsvc_grid_backward_ex")
            outs.append(ge)
        return (None, None, *outs)


MFMA_MAX_DIM = 192      # csrc/linear.hip keeps the whole weight matrix in LDS: in/out features <= 192


class _LinearMFMA(torch.autograd.Function):
    """y = x W^T + b (optionally followed by ReLU) through the weight-stationary MFMA kernel of csrc/linear.hip.
    Backward: dX = G W through the same kernel (W read input-major, no transposed copy); dW = G^T X and db through
    the row-split MFMA kernel ``gsvc_linear_wgrad``."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu=False):
        from . import _lib
        x, w = x.contiguous(), weight.contiguous()
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        b = bias.contiguous() if bias is not None else None
        _lib.check(_lib.lib().gsvc_linear_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, K, N, 0, int(relu),
                                                     _lib.current_stream(x.device)), "gsvc_linear_forward")
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        x, w, y = ctx.saved_tensors
        if y is not None:
            g = torch.ops.aten.threshold_backward(g, y, 0.0)     # ReLU': pass where the output was > 0
        g = g.contiguous()
        M, K = x.shape
        N = w.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(M, K, device=x.device, dtype=torch.float32)
            _lib.check(_lib.lib().gsvc_linear_forward(_lib.ptr(g), _lib.ptr(w), None, _lib.ptr(gx), M, N, K, 1, 0,
                                                         _lib.current_stream(x.device)), "gsvc_linear_forward")
        if ctx.needs_input_grad[1]:
            # dW = G^T X and db = column sums of G in one pass over G and X (rows split over the chip)
            want_b = ctx.has_bias and ctx.needs_input_grad[2]
            L = _lib.lib()
            ws_floats = int(L.gsvc_linear_wgrad_workspace(N, K))
            buf = torch.empty(N * K + N + ws_floats, device=x.device, dtype=torch.float32)
            gw = buf[:N * K].view(N, K)
            gb = buf[N * K:N * K + N] if want_b else None
            ws = buf[N * K + N:]
            _lib.check(L.gsvc_linear_wgrad(_lib.ptr(g), _lib.ptr(x), _lib.ptr(gw), _lib.ptr(gb), M, N, K, _lib.ptr(ws), ws_floats,
                                           _lib.current_stream(x.device)), "gsvc_linear_wgrad")
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(dim=0)
        return gx, gw, gb, None


class Linear(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) whose forward on a tall CUDA matrix runs on the MFMA kernel."""

    MIN_ROWS = 4096

    def _use_mfma(self, x):
        return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] >= self.MIN_ROWS
                and self.out_features <= MFMA_MAX_DIM and self.in_features <= MFMA_MAX_DIM)

    def forward(self, x):
        if self._use_mfma(x):
            return _LinearMFMA.apply(x, self.weight, self.bias)
        return super().forward(x)

    def forward_relu(self, x):
        """relu(linear(x)) with the ReLU fused into the kernel's store."""
        if self._use_mfma(x):
            return _LinearMFMA.apply(x, self.weight, self.bias, True)
        return torch.relu(super().forward(x))


class Sequential(nn.Sequential):
    """nn.Sequential (same child names, hence the reference's state_dict keys) that runs each Linear -> ReLU pair
    as one fused call."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, Linear) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                x = m.forward_relu(x)
                i += 2
            else:
                x = m(x)
                i += 1
        return x


class GeluSequential(nn.Sequential):
    """nn.Sequential of Linear -> GELU -> ... -> Linear (same child names, hence the reference's state_dict keys).  A tall CUDA
    matrix runs the whole chain as ONE autograd function with the GELUs on the GEMM epilogues (gsvc_amd/mlp.py)."""

    def forward(self, x):
        from . import mlp
        mods = list(self)
        linears = mods[0::2]
        if (all(isinstance(m, nn.Linear) for m in linears) and all(isinstance(m, nn.GELU) for m in mods[1::2])
                and len(mods) % 2 == 1 and mlp.usable(x, *linears) and not switches.NO_MLP_CHAIN):
            return mlp.seq_gelu(x, linears)
        return super().forward(x)


class FiLM(nn.Module):
    """gamma(cond) * x + beta(cond), both from 2-layer ReLU MLPs."""

    def __init__(self, condition_dim, input_dim):
        super().__init__()
        self.fc_gamma0 = Linear(condition_dim, condition_dim)
        self.fc_beta0 = Linear(condition_dim, condition_dim)
        self.fc_gamma1 = Linear(condition_dim, input_dim)
        self.fc_beta1 = Linear(condition_dim, input_dim)
        self.act = nn.ReLU()

    def forward(self, x, condition):
        gamma = self.fc_gamma1(self.fc_gamma0.forward_relu(condition))
        beta = self.fc_beta1(self.fc_beta0.forward_relu(condition))
        return gamma * x + beta


class GeneratorNet(nn.Module):
    def __init__(self, input_dim, output_dim, inner_dim, condition_dim, out_act=None):
        super().__init__()
        self.linear1 = Linear(input_dim, inner_dim)
        self.linear2 = Linear(inner_dim, inner_dim)
        self.act = nn.GELU()
        self.out_linear = Linear(inner_dim, output_dim)
        self.film = FiLM(condition_dim, inner_dim)
        self.out_act = nn.Identity() if out_act is None else out_act

    def trunk(self, feature):
        """The part that only sees the anchor feature (frame independent when the feature carries no noise)."""
        return self.linear2(self.act(self.linear1(feature)))

    def head(self, h, condition):
        return self.out_act(self.out_linear(self.film(h, condition)))

    def _fusable(self, feature):
        from . import mlp
        return (type(self.out_act).__name__ in ("Tanh", "Sigmoid", "Identity") and not switches.NO_MLP_CHAIN
                and mlp.usable(feature, self.linear1, self.linear2, self.out_linear))

    def film_nets(self, condition):
        """(gamma, beta) of this network's FiLM for ``condition`` as one autograd function, or None when the fused path does
        not apply.  They depend on the condition only: a caller may evaluate them ahead and pass ``film=`` to forward."""
        from . import mlp
        f = self.film
        if (not switches.NO_MLP_CHAIN
                and mlp.usable(condition, f.fc_gamma0, f.fc_gamma1, f.fc_beta0, f.fc_beta1)):
            return mlp.film_nets(f, condition)
        return None

    def forward(self, feature, condition, film=None):
        from . import mlp
        if self._fusable(feature):
            if film is None:
                film = self.film_nets(condition)
            if film is not None:
                return mlp.generator(self, feature, film=film)      # FiLM nets + trunk: two autograd functions
        if film is not None:
            return self.out_act(self.out_linear(film[0] * self.trunk(feature) + film[1]))
        return self.head(self.trunk(feature), condition)


class EntropyParamsNet(nn.Module):
    def __init__(self, input_dim, inner_dim, inner_dim2, output_dim, layer=2):
        super().__init__()
        if layer == 2:
            self.dist_net = GeluSequential(Linear(input_dim, inner_dim), nn.GELU(), Linear(inner_dim, output_dim * 2))
        else:
            assert layer == 3
            self.dist_net = GeluSequential(Linear(input_dim, inner_dim), nn.GELU(), Linear(inner_dim, inner_dim),
                                           nn.GELU(), Linear(inner_dim, output_dim * 2))
        self.quant_step_net = GeluSequential(Linear(input_dim, inner_dim2), nn.GELU(), Linear(inner_dim2, 1))

    def forward(self, x):
        params = self.dist_net(x)
        half = params.shape[1] // 2
        mean, scale = params.split([half, params.shape[1] - half], dim=1)     # split: one cat in backward
        return mean, scale, self.quant_step_net(x)


class _CtxPost(torch.autograd.Function):
    """(mean, max(scale, 1e-9), exp(clamp(q, -10, 10))) from an EntropyParamsNet's raw outputs params = [mean | scale] and q
    (reference scene/gaussian_model.py:1586-1596)."""

    @staticmethod
    def forward(ctx, params, q):
        from . import _lib
        params, q = params.contiguous(), q.contiguous()
        n, C = params.shape[0], params.shape[1] // 2
        f = lambda *sh: torch.empty(*sh, device=params.device, dtype=torch.float32)  # noqa: E731
        mean, scale, adj = f(n, C), f(n, C), f(n, 1)
        _lib.check(_lib.lib().gsvc_ctx_post_forward(_lib.ptr(params), _lib.ptr(q), n, C, _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(adj),
                                                    _lib.current_stream(params.device)), "gsvc_ctx_post_forward")
        ctx.save_for_backward(params, q, adj)
        ctx.set_materialize_grads(False)
        return mean, scale, adj

    @staticmethod
    def backward(ctx, g_mean, g_scale, g_adj):
        from . import _lib
        params, q, adj = ctx.saved_tensors
        n, C = params.shape[0], params.shape[1] // 2
        c = lambda g: g.contiguous() if g is not None else None  # noqa: E731
        g_mean, g_scale, g_adj = c(g_mean), c(g_scale), c(g_adj)
        if g_mean is None and g_scale is None and g_adj is None:
            return None, None
        dparams, dq = torch.empty_like(params), torch.empty_like(q)
        _lib.check(_lib.lib().gsvc_ctx_post_backward(_lib.ptr(params), _lib.ptr(q), _lib.ptr(adj), n, C, _lib.ptr(g_mean),
                                                     _lib.ptr(g_scale), _lib.ptr(g_adj), _lib.ptr(dparams), _lib.ptr(dq),
                                                     _lib.current_stream(params.device)), "gsvc_ctx_post_backward")
        return dparams, dq


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000, step_sub=0):
    """Log-linear interpolation lr_init -> lr_final over max_steps (reference utils/general_utils.py:49-82)."""

    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        else:
            delay_rate = 1.0
        t = np.clip((step - step_sub) / (max_steps - step_sub), 0, 1)
        return delay_rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)

    return helper


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def mean_3nn_dist2(points: torch.Tensor, chunk: int = 4096) -> torch.Tensor:
    """Mean squared distance to the 3 nearest neighbours (what simple_knn.distCUDA2 returns; reference
    scene/gaussian_model.py:762,784).  CUDA tensors: uniform-grid binning here, exact search in csrc/knn.hip
    (O(n) for the roughly uniform clouds GSVC starts from).  CPU tensors (host-side tests): chunked brute force."""
    n = points.shape[0]
    if points.is_cuda:
        return _mean_3nn_dist2_hip(points.float().contiguous())
    out = torch.empty(n, device=points.device, dtype=points.dtype)
    for s in range(0, n, chunk):
        d = torch.cdist(points[s:s + chunk], points) ** 2
        k = min(4, n)
        near = torch.topk(d, k, dim=1, largest=False).values[:, 1:]
        out[s:s + chunk] = near.mean(dim=1) if k > 1 else 0.0
    return out


def _mean_3nn_dist2_hip(pts: torch.Tensor, per_cell: float = 4.0, max_cells: int = 1 << 25) -> torch.Tensor:
    import ctypes as C
    from . import _lib
    n = pts.shape[0]
    if n == 0:
        return torch.empty(0, device=pts.device)
    lo, hi = pts.min(dim=0).values, pts.max(dim=0).values
    ext = (hi - lo).clamp_min(1e-12).double().cpu()
    lo_h = lo.double().cpu()
    # cell edge for ~per_cell points per cell over the occupied box (flat clouds: the thin axes get one cell)
    cells = max(1.0, min(n / per_cell, float(max_cells)))
    edge = float((ext.prod() / cells) ** (1.0 / 3.0))
    edge = max(edge, float(ext.max()) / 1024.0, 1e-12)
    dims = [max(1, int(float(e) / edge) + 1) for e in ext]
    while dims[0] * dims[1] * dims[2] > max_cells:
        edge *= 1.26
        dims = [max(1, int(float(e) / edge) + 1) for e in ext]
    gx, gy, gz = dims
    edge32 = float(torch.tensor(edge, dtype=torch.float32))
    origin = (C.c_float * 3)(*[float(v) for v in lo_h])
    o32 = torch.tensor([origin[0], origin[1], origin[2]], device=pts.device)
    # the kernel's own cell rule: floor((p - origin) / edge) in float32, clamped
    cxyz = torch.floor((pts - o32) / edge32).to(torch.int64)
    cxyz[:, 0].clamp_(0, gx - 1); cxyz[:, 1].clamp_(0, gy - 1); cxyz[:, 2].clamp_(0, gz - 1)
    cell = (cxyz[:, 2] * gy + cxyz[:, 1]) * gx + cxyz[:, 0]
    order = torch.argsort(cell)
    sorted_pts = pts.index_select(0, order).contiguous()
    counts = torch.bincount(cell, minlength=gx * gy * gz)
    cell_start = torch.zeros(gx * gy * gz + 1, dtype=torch.int32, device=pts.device)
    cell_start[1:] = torch.cumsum(counts, 0).to(torch.int32)
    out_sorted = torch.empty(n, device=pts.device)
    _lib.check(_lib.lib().gsvc_knn3_mean_dist2(_lib.ptr(sorted_pts), _lib.ptr(cell_start), origin, edge32, gx, gy, gz, n,
                                               _lib.ptr(out_sorted), _lib.current_stream(pts.device)), "gsvc_knn3_mean_dist2")
    out = torch.empty_like(out_sorted)
    out[order] = out_sorted
    return out


class GaussianModel(nn.Module):
    def __init__(self, model_config, feat_dim: int = 32, n_offsets: int = 5, voxel_size: float = 0.01,
                 update_depth: int = 3, update_init_factor: int = 100, update_hierachy_factor: int = 4,
                 use_feat_bank=False, n_features_per_level: int = 2, log2_hashmap_size: int = 19,
                 log2_hashmap_size_2D: int = 17,
                 resolutions_list=(18, 24, 33, 44, 59, 80, 108, 148, 201, 275, 376, 514),
                 resolutions_list_2D=(130, 258, 514, 1026), ste_binary: bool = True, ste_multistep: bool = False,
                 add_noise: bool = False, Q=1, use_2D: bool = True, decoded_version: bool = False, device=None):
        super().__init__()
        assert not use_feat_bank, "the feature bank is disabled in GSVC (gaussian_model.py:532)"
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        self.model_config = model_config
        self.feat_dim, self.n_offsets, self.voxel_size = feat_dim, n_offsets, voxel_size
        self.update_depth, self.update_init_factor, self.update_hierachy_factor = update_depth, update_init_factor, update_hierachy_factor
        self.use_feat_bank = use_feat_bank
        self.x_bound_min = torch.zeros(1, 3, device=self.device)
        self.x_bound_max = torch.ones(1, 3, device=self.device)
        self.bound_min_host, self.bound_max_host = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
        self.n_features_per_level = n_features_per_level
        self.log2_hashmap_size, self.log2_hashmap_size_2D = log2_hashmap_size, log2_hashmap_size_2D
        self.resolutions_list, self.resolutions_list_2D = resolutions_list, resolutions_list_2D
        self.ste_binary, self.ste_multistep, self.add_noise, self.Q = ste_binary, ste_multistep, add_noise, Q
        self.use_2D, self.decoded_version = use_2D, decoded_version

        for name in ("_anchor", "_offset", "_mask", "_anchor_feat", "_scaling", "_rotation", "_opacity"):
            setattr(self, name, torch.empty(0))
        self.opacity_accum = self.offset_gradient_accum = self.offset_denom = self.anchor_demon = torch.empty(0)
        self.max_radii2D = torch.empty(0)
        self.optimizer = None
        self.percent_dense = 0
        self.spatial_lr_scale = 0
        self.scaling_activation, self.scaling_inverse_activation = torch.exp, torch.log
        self.opacity_activation, self.inverse_opacity_activation = torch.sigmoid, inverse_sigmoid
        self.rotation_activation = torch.nn.functional.normalize

        grid_kw = dict(n_features=n_features_per_level, resolutions_list=resolutions_list,
                       log2_hashmap_size=log2_hashmap_size, ste_binary=ste_binary, ste_multistep=ste_multistep,
                       add_noise=add_noise, Q=Q)
        if use_2D:
            self.encoding_xyz = Mix3d2dEncoding(resolutions_list_2D=resolutions_list_2D,
                                                log2_hashmap_size_2D=log2_hashmap_size_2D, **grid_kw)
        else:
            self.encoding_xyz = GridEncoder(num_dim=3, **grid_kw)

        self.embed_time_fn, time_ch = get_embedder(model_config.time_multi_res, 1)
        self.embed_fn, z_ch = get_embedder(model_config.offset_multi_res, 1)
        cond = time_ch + z_ch
        inner = feat_dim * 2
        self.mlp_opacity = GeneratorNet(feat_dim, n_offsets, inner, cond, out_act=nn.Tanh())
        self.mlp_cov = GeneratorNet(feat_dim, 7 * n_offsets, inner, cond)
        self.mlp_color = GeneratorNet(feat_dim, 3 * n_offsets, inner, cond, out_act=nn.Sigmoid())
        self.mlp_deform = GeluSequential(
            Linear(feat_dim + cond, inner), nn.GELU(), Linear(inner, inner), nn.GELU(),
            Linear(inner, inner), nn.GELU(), Linear(inner, inner), nn.GELU(), Linear(inner, 3 * n_offsets))
        gdim = self.encoding_xyz.output_dim
        self.mlp_feature_enet = EntropyParamsNet(gdim, feat_dim * 3, feat_dim, feat_dim)
        self.mlp_scaling_enet = EntropyParamsNet(gdim, feat_dim * 2, feat_dim, 6, layer=3)
        self.mlp_offset_enet = EntropyParamsNet(gdim, feat_dim * 3, feat_dim, 3 * n_offsets)
        self.noise_quantizer = UniformQuantizer()
        self.entropy_gaussian = EntropyGaussian(Q=1)
        self.to(self.device)

    # ------------------------------------------------------------------ getters (reference :641-700)
    @property
    def get_scaling(self):
        return self._scaling if self.decoded_version else 1.0 * self.scaling_activation(self._scaling)

    @property
    def get_mask(self):
        if self.decoded_version:
            return self._mask
        s = torch.sigmoid(self._mask)
        return ((s > 0.01).float() - s).detach() + s  # hard threshold forward, sigmoid gradient

    @property
    def get_mask_anchor(self):
        with torch.no_grad():
            m = self._mask if self.decoded_version else (torch.sigmoid(self._mask) > 0.01).float()
            return (torch.sum(m, dim=1)[:, 0]) > 0

    @property
    def get_opacity_mlp(self):
        return self.mlp_opacity

    @property
    def get_cov_mlp(self):
        return self.mlp_cov

    @property
    def get_color_mlp(self):
        return self.mlp_color

    @property
    def get_deform_mlp(self):
        return self.mlp_deform

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_anchor(self):
        if self.decoded_version:
            return self._anchor
        return Quantize_anchor.apply(self._anchor, self.x_bound_min, self.x_bound_max)[0]

    @property
    def quantized_anchor(self):
        return Quantize_anchor.quantized(self._anchor, self.x_bound_min, self.x_bound_max)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    def get_encoding_params(self):
        enc = self.encoding_xyz
        tables = [enc.encoding_xyz.params, enc.encoding_xy.params, enc.encoding_xz.params, enc.encoding_yz.params] \
            if self.use_2D else [enc.params]
        p = torch.cat(tables, dim=0)
        return STE_binary.apply(p) if self.ste_binary else p

    @torch.no_grad()
    def update_anchor_bound(self, x_lim, y_lim, z_lim, bleed=0.1):
        lim = [x_lim * (1 + bleed), y_lim * (1 + bleed), z_lim * (1 + bleed)]  # python floats, rounded once to fp32
        self.x_bound_min = torch.tensor([lim], dtype=torch.float32, device=self.device)
        self.x_bound_max = torch.tensor([[-v for v in lim]], dtype=torch.float32, device=self.device)
        # the same fp32 values on the host (kernels take the bounds by value)
        self.bound_min_host = tuple(float(np.float32(v)) for v in lim)
        self.bound_max_host = tuple(float(np.float32(-v)) for v in lim)

    def calc_interp_feat(self, x):
        assert x.dim() == 2 and x.shape[1] == 3
        x = (x - self.x_bound_min) / (self.x_bound_max - self.x_bound_min)
        return self.encoding_xyz(x)

    def calc_entropy_context(self, anchor) -> EntropyContext:
        ctx = self.calc_interp_feat(anchor)
        if ctx.is_cuda and not switches.NO_FUSED_CTX:
            # split, scale clamp and step activation of each network's raw outputs as one launch each way (csrc/generate.hip)
            out = []
            nets = (self.mlp_feature_enet, self.mlp_scaling_enet, self.mlp_offset_enet)
            from . import mlp
            chains = [list(s)[0::2] for net in nets for s in (net.dist_net, net.quant_step_net)]
            if (all(isinstance(s, GeluSequential) for net in nets for s in (net.dist_net, net.quant_step_net))
                    and all(mlp.usable(ctx, *c) for c in chains) and not switches.NO_MLP_CHAIN):
                # the six sub-networks read the same feature matrix: one autograd function (gsvc_amd.mlp._SeqGeluMany)
                raw = mlp.seq_gelu_many(ctx, chains)
                for i in range(3):
                    out += list(_CtxPost.apply(raw[2 * i], raw[2 * i + 1]))
            else:
                for net in nets:
                    out += list(_CtxPost.apply(net.dist_net(ctx), net.quant_step_net(ctx)))
            mf, sf, qf, ms, ss, qs, mo, so, qo = out
            return EntropyContext(mf, sf, ms, ss, mo, so, qf, qs, qo)
        mean_f, scale_f, q_f = self.mlp_feature_enet(ctx)
        mean_s, scale_s, q_s = self.mlp_scaling_enet(ctx)
        mean_o, scale_o, q_o = self.mlp_offset_enet(ctx)
        adj = lambda q: torch.exp(torch.clamp(q, min=-10, max=10))  # noqa: E731
        return EntropyContext(mean_f, torch.clamp(scale_f, 1e-9), mean_s, torch.clamp(scale_s, 1e-9),
                              mean_o, torch.clamp(scale_o, 1e-9), adj(q_f), adj(q_s), adj(q_o))

    def calc_entropy_context_sampled(self, anchor) -> SampledEntropyContext:
        """``calc_entropy_context`` for the training rate: hash-grid feature + the three quant_step networks on every row, the
        dist networks deferred to the sampled rows (SampledEntropyContext.rows)."""
        from . import mlp
        ctx = self.calc_interp_feat(anchor)
        nets = (self.mlp_feature_enet, self.mlp_scaling_enet, self.mlp_offset_enet)
        chains = [list(net.quant_step_net)[0::2] for net in nets]
        if mlp.quant_step_nets_usable(ctx, [net.quant_step_net for net in nets]):
            q_raw = mlp.quant_step_nets(ctx, [net.quant_step_net for net in nets])      # one chain launch each way for all three
        elif (ctx.is_cuda and all(isinstance(net.quant_step_net, GeluSequential) for net in nets) and all(mlp.usable(ctx, *c) for c in chains)
                and not switches.NO_MLP_CHAIN):
            q_raw = mlp.seq_gelu_many(ctx, chains)          # the three first layers read ctx in one launch
        else:
            q_raw = tuple(net.quant_step_net(ctx) for net in nets)
        return SampledEntropyContext(self, ctx, q_raw)

    # ------------------------------------------------------------------ bit accounting
    def get_mlp_size(self, digit=32):
        n = sum(p.numel() for name, p in self.named_parameters() if "mlp" in name)
        return n * digit, n * digit / 8 / 1024 / 1024

    @torch.no_grad()
    def estimate_final_bits(self):
        """Size the codec would produce, from the entropy model alone: anchors kept by the mask, attributes rounded to
        their context-scaled steps and priced by the Gaussian model in symbol units, Bernoulli code length of the
        binarised hash tables and of the offset masks, raw MLP weights.  Same arithmetic and order as reference
        scene/gaussian_model.py:1599-1725; returns (log line, BitInfo)."""
        from .encodings import ANCHOR_ROUND_DIGITS, STE_multistep, get_binary_vxl_size
        K = self.n_offsets
        keep = self.get_mask_anchor
        anchor = self.get_anchor[keep]
        feat, offsets, scaling, mask = self._anchor_feat[keep], self._offset[keep], self.get_scaling[keep], self.get_mask[keep]
        ec = self.calc_entropy_context(anchor)
        Q_feat, Q_scaling, Q_offsets = 1 * ec.Q_feat_adj, 0.001 * ec.Q_scaling_adj, 0.2 * ec.Q_offsets_adj
        feat_min, feat_max = calc_symbol_min_max(ec.mean_feat, Q_feat)
        scaling_min, scaling_max = calc_symbol_min_max(ec.mean_scaling, Q_scaling)
        offsets_min, offsets_max = calc_symbol_min_max(ec.mean_offsets, Q_offsets)
        q_feat = STE_multistep.quantize(feat, Q_feat, feat_min, feat_max)
        q_scaling = STE_multistep.quantize(scaling, Q_scaling, scaling_min, scaling_max)
        q_offsets = STE_multistep.quantize(offsets, Q_offsets.unsqueeze(1), offsets_min, offsets_max).view(-1, 3 * K)
        mask3 = mask.repeat(1, 1, 3).view(-1, 3 * K)
        bit_feat = self.entropy_gaussian(q_feat, ec.mean_feat, ec.scale_feat, Q_feat, quantized=True)
        bit_scaling = self.entropy_gaussian(q_scaling, ec.mean_scaling, ec.scale_scaling, Q_scaling, quantized=True)
        bit_offsets = self.entropy_gaussian(q_offsets, ec.mean_offsets, ec.scale_offsets, Q_offsets, quantized=True) * mask3
        bit_anchor = anchor.shape[0] * 3 * ANCHOR_ROUND_DIGITS
        bit_feat, bit_scaling, bit_offsets = bit_feat.sum().item(), bit_scaling.sum().item(), bit_offsets.sum().item()
        tables = self.get_encoding_params()
        bit_hash = get_binary_vxl_size((tables + 1) / 2)[1].item() if self.ste_binary else tables.numel() * 32
        bit_masks = get_binary_vxl_size(mask)[1].item()
        bit_mlp = self.get_mlp_size()[0]
        info = BitInfo(bit_anchor, bit_anchor / 2, bit_feat, bit_scaling, bit_offsets, bit_hash, bit_masks, bit_mlp,
                       int(bit_mlp * 0.3))
        mb = lambda b: round(b / BIT2MB_SCALE, 4)  # noqa: E731
        total = bit_anchor + bit_feat + bit_scaling + bit_offsets + bit_hash + bit_masks + bit_mlp
        log = (f"Estimated sizes in MB: anchor {mb(bit_anchor)}, feat {mb(bit_feat)}, scaling {mb(bit_scaling)}, "
               f"offsets {mb(bit_offsets)}, hash {mb(bit_hash)}, masks {mb(bit_masks)}, MLPs {mb(bit_mlp)}, Total {mb(total)}")
        return log, info

    # ------------------------------------------------------------------ MLP weight coding (reference :1727-1835, gsvc_amd/mlp_codec.py)
    def quantize_model(self, replace=True):
        from . import mlp_codec
        return mlp_codec.quantize_model(self, replace)

    def encode_mlp(self, file_path):
        from . import mlp_codec
        return mlp_codec.encode_mlp(self, file_path)

    # ------------------------------------------------------------------ files (reference :556-639, 1156-1240, 1505-1540; gsvc_amd/io.py)
    def capture(self):
        from . import io
        return io.capture(self)

    def restore(self, model_args, training_args):
        from . import io
        return io.restore(self, model_args, training_args)

    def init_anchor_params(self, anchor_num):
        from . import io
        return io.init_anchor_params(self, anchor_num)

    def save_ply(self, path):
        from . import io
        return io.save_ply(self, path)

    def load_ply_sparse_gaussian(self, path):
        from . import io
        return io.load_ply_sparse_gaussian(self, path)

    def save_mlp_checkpoints(self, path):
        from . import io
        return io.save_mlp_checkpoints(self, path)

    def load_mlp_checkpoints(self, path):
        from . import io
        return io.load_mlp_checkpoints(self, path)

    # ------------------------------------------------------------------ initialisation (reference :748-800)
    def voxelize_sample(self, data, voxel_size=0.01):
        np.random.shuffle(data)
        return np.unique(np.round(data / voxel_size), axis=0) * voxel_size

    def create_from_points(self, points: np.ndarray, spatial_lr_scale: float = 1.0):
        self.spatial_lr_scale = spatial_lr_scale
        dev = self.device
        if self.voxel_size <= 0:
            d = mean_3nn_dist2(torch.tensor(points).float().to(dev))
            self.voxel_size = torch.kthvalue(d, int(d.shape[0] * 0.5)).values.item()
        pts = torch.tensor(np.asarray(self.voxelize_sample(points, voxel_size=self.voxel_size))).float().to(dev)
        A, K = pts.shape[0], self.n_offsets
        dist2 = torch.clamp_min(mean_3nn_dist2(pts), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 6)
        rots = torch.zeros(A, 4, device=dev)
        rots[:, 0] = 1
        self._anchor = nn.Parameter(pts.requires_grad_(True))
        self._offset = nn.Parameter(torch.zeros(A, K, 3, device=dev).requires_grad_(True))
        self._mask = nn.Parameter(torch.ones(A, K, 1, device=dev).requires_grad_(True))
        self._anchor_feat = nn.Parameter(torch.zeros(A, self.feat_dim, device=dev).requires_grad_(True))
        self._scaling = nn.Parameter(scales.requires_grad_(True))
        self._rotation = nn.Parameter(rots.requires_grad_(False))
        self._opacity = nn.Parameter(inverse_sigmoid(0.1 * torch.ones(A, 1, device=dev)).requires_grad_(False))
        self.max_radii2D = torch.zeros(A, device=dev)

    def create_from_pcd(self, pcd, spatial_lr_scale: float):
        self.create_from_points(np.asarray(pcd.points), spatial_lr_scale)

    # ------------------------------------------------------------------ optimiser (reference :833-1058)
    def register_training_params(self, name, module, lr, scheduler_func=None):
        assert name not in self.net_params_registry
        self.net_params_registry[name] = {"params": module if isinstance(module, list) else module.parameters(),
                                          "lr": lr, "name": name}
        self.scheduler_registry[name] = scheduler_func if scheduler_func is not None else (lambda x: lr)

    def training_setup(self, training_args):
        self.net_params_registry, self.scheduler_registry = OrderedDict(), OrderedDict()
        self.percent_dense = training_args.percent_dense
        A, K, dev = self._anchor.shape[0], self.n_offsets, self.device
        self.opacity_accum = torch.zeros(A, 1, device=dev)
        self.offset_gradient_accum = torch.zeros(A * K, 1, device=dev)
        self.offset_denom = torch.zeros(A * K, 1, device=dev)
        self.anchor_demon = torch.zeros(A, 1, device=dev)
        ta, sl = training_args, self.spatial_lr_scale

        def sched(prefix, scale=1.0, **kw):
            return get_expon_lr_func(lr_init=getattr(ta, prefix + "_lr_init") * scale, lr_final=getattr(ta, prefix + "_lr_final") * scale,
                                     lr_delay_mult=getattr(ta, prefix + "_lr_delay_mult"), max_steps=getattr(ta, prefix + "_lr_max_steps"), **kw)

        reg = self.register_training_params
        reg("anchor", [self._anchor], ta.position_lr_init * sl, sched("position", sl))
        reg("offset", [self._offset], ta.offset_lr_init * sl, sched("offset", sl))
        reg("mask", [self._mask], ta.mask_lr_init * sl, sched("mask", sl))
        reg("anchor_feat", [self._anchor_feat], ta.feature_lr)
        reg("opacity", [self._opacity], ta.opacity_lr)
        reg("scaling", [self._scaling], ta.scaling_lr)
        reg("rotation", [self._rotation], ta.rotation_lr)
        reg("mlp_opacity", self.mlp_opacity, ta.mlp_opacity_lr_init, sched("mlp_opacity"))
        reg("mlp_cov", self.mlp_cov, ta.mlp_cov_lr_init, sched("mlp_cov"))
        reg("mlp_color", self.mlp_color, ta.mlp_color_lr_init, sched("mlp_color"))
        reg("encoding_xyz", self.encoding_xyz, ta.encoding_xyz_lr_init,
            sched("encoding_xyz", step_sub=0 if self.ste_binary else 10000))
        reg("mlp_deform", self.mlp_deform, ta.mlp_deform_lr_init, sched("mlp_deform"))
        for name in ("mlp_feature_enet", "mlp_scaling_enet", "mlp_offset_enet"):
            reg(name, getattr(self, name), ta.mlp_entropy_net_lr_init, sched("mlp_entropy_net"))
        # on the GPU: one launch updates every tensor (gsvc_amd/optim.py, csrc/adam.hip); same rule as the reference's Adam
        if self._anchor.is_cuda:
            from .optim import FusedAdam
            self.optimizer = FusedAdam(self.net_params_registry.values(), lr=0.0, eps=1e-15)
        else:
            self.optimizer = torch.optim.Adam(self.net_params_registry.values(), lr=0.0, eps=1e-15)

    def update_learning_rate(self, iteration):
        for group in self.optimizer.param_groups:
            group["lr"] = self.scheduler_registry[group["name"]](iteration)

    # ------------------------------------------------------------------ densification / pruning (:1242-1505, gsvc_amd/densify.py)
    def replace_tensor_to_optimizer(self, tensor, name):
        from . import densify
        return densify.replace_tensor_to_optimizer(self, tensor, name)

    def cat_tensors_to_optimizer(self, tensors_dict):
        from . import densify
        return densify.cat_tensors_to_optimizer(self, tensors_dict)

    def prune_anchor(self, mask):
        from . import densify
        return densify.prune_anchor(self, mask)

    def anchor_growing(self, grads, threshold, offset_mask):
        from . import densify
        return densify.anchor_growing(self, grads, threshold, offset_mask)

    def adjust_anchor(self, check_interval=100, success_threshold=0.8, grad_threshold=0.0002, min_opacity=0.005):
        from . import densify
        return densify.adjust_anchor(self, check_interval, success_threshold, grad_threshold, min_opacity)

    # ------------------------------------------------------------------ densification statistics (:1281-1314)
    @torch.no_grad()
    def training_statis_many(self, renders):
        """``training_statis`` of the un-compacted renders of one step in one pass: the accumulators are sums over renders,
        so the rows of all renders are concatenated and scattered with ONE index_add per accumulator (instead of four)."""
        if not renders or not all(r.dense for r in renders):
            for r in renders:
                self.training_statis(r)
            return
        K, A = self.n_offsets, self.opacity_accum.shape[0]
        batch = getattr(renders[0].generated_gaussians, "batch", None)
        if batch is not None and len(renders) > 1 and getattr(batch, "vis", None) is not None:
            vi, op_all = batch.vis, batch.neural_opacity
        else:
            vi, op_all = torch.cat([r.visible_index for r in renders]), torch.cat([r.neural_opacity for r in renders])
        if batch is not None and getattr(batch, "viewspace", None) is not None and len(renders) > 1:
            # rasterize_many: one leaf, one radii tensor for all renders
            seen, g = batch.seen, batch.viewspace.grad
        else:
            seen, g = torch.cat([r.visibility_filter for r in renders]), torch.cat([r.viewspace_points.grad for r in renders])
        accs = (self.opacity_accum, self.anchor_demon, self.offset_gradient_accum, self.offset_denom)
        if switches.DETERMINISTIC and vi.is_cuda:
            # an anchor's rows (one per render that sees it) added in row order: per-row values side by side, one sorted scatter
            from .generate import det_scatter_rows
            op = op_all.detach().view(-1).clamp_min(0).view(-1, K)
            w = seen.to(torch.float32).view(-1, K)
            gn = torch.norm(g[:, :2], dim=-1).view(-1, K) * w
            vals = torch.cat([op.sum(dim=1, keepdim=True), torch.ones_like(op[:, :1]), gn, w], dim=1)
            tot = det_scatter_rows(vi, vals, A)
            self.opacity_accum.add_(tot[:, 0:1])
            self.anchor_demon.add_(tot[:, 1:2])
            self.offset_gradient_accum.view(A, K).add_(tot[:, 2:2 + K])
            self.offset_denom.view(A, K).add_(tot[:, 2 + K:2 + 2 * K])
            return
        if (vi.is_cuda and seen.dtype == torch.bool and g.dtype == torch.float32 and g.dim() == 2 and g.stride(1) == 1
                and all(a.dtype == torch.float32 and a.is_contiguous() for a in accs) and not switches.NO_FUSED_STATIS):
            # one launch (csrc/rate.hip k_training_statis) for the clamp, the sums, the gradient norms and the four scatters
            from . import _lib
            op = op_all.detach().contiguous()
            _lib.check(_lib.lib().gsvc_training_statis(_lib.ptr(vi.contiguous()), _lib.ptr(op), _lib.ptr(seen.contiguous()), _lib.ptr(g),
                                                       g.stride(0), vi.shape[0], K, *[_lib.ptr(a) for a in accs],
                                                       _lib.current_stream(vi.device)), "gsvc_training_statis")
            return
        op = op_all.detach().view(-1).clamp_min(0).view(-1, K)
        self.opacity_accum.index_add_(0, vi, op.sum(dim=1, keepdim=True))
        self.anchor_demon.index_add_(0, vi, torch.ones(vi.shape[0], 1, device=vi.device, dtype=self.anchor_demon.dtype))
        w = seen.to(self.offset_denom.dtype).view(-1, K)
        gn = torch.norm(g[:, :2], dim=-1).view(-1, K) * w
        self.offset_gradient_accum.view(A, K).index_add_(0, vi, gn.to(self.offset_gradient_accum.dtype))
        self.offset_denom.view(A, K).index_add_(0, vi, w)

    @torch.no_grad()
    def training_statis(self, render_results):
        """Accumulate, through the nested masks visible anchor -> opacity>0 -> radius>0: per-anchor positive
        opacity sums and visit counts, per-offset screen-gradient norms and counts."""
        K = self.n_offsets
        op = render_results.neural_opacity.detach().view(-1).clamp_min(0).view(-1, K)
        if render_results.dense:
            # un-compacted results: every visible anchor carries its K slots, so the nested masks become weights and
            # the scatter is by anchor row — no boolean indexing, no host synchronisation
            vi = render_results.visible_index
            A = self.opacity_accum.shape[0]
            self.opacity_accum.index_add_(0, vi, op.sum(dim=1, keepdim=True))
            self.anchor_demon.index_add_(0, vi, torch.ones(vi.shape[0], 1, device=vi.device, dtype=self.anchor_demon.dtype))
            w = render_results.visibility_filter.to(self.offset_denom.dtype).view(-1, K)
            gn = torch.norm(render_results.viewspace_points.grad[:, :2], dim=-1).view(-1, K) * w
            self.offset_gradient_accum.view(A, K).index_add_(0, vi, gn.to(self.offset_gradient_accum.dtype))
            self.offset_denom.view(A, K).index_add_(0, vi, w)
            return
        vis = render_results.visible_mask
        self.opacity_accum[vis] += op.sum(dim=1, keepdim=True)
        self.anchor_demon[vis] += 1
        # flat (anchor*K + slot) index of every rasterised Gaussian that ended up with radius > 0
        slots = torch.arange(vis.shape[0] * K, device=vis.device).view(-1, K)[vis].reshape(-1)
        slots = slots[render_results.selection_mask][render_results.visibility_filter]
        g = render_results.viewspace_points.grad[render_results.visibility_filter, :2]
        self.offset_gradient_accum[slots] += torch.norm(g, dim=-1, keepdim=True)
        self.offset_denom[slots] += 1
