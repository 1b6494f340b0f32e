"""Whole-MLP autograd functions for the generator / deformation / entropy-parameter networks.

The reference evaluates these networks layer by layer (scene/gaussian_model.py:150-232, 411-501: ``nn.Linear`` ->
``nn.GELU`` -> ..., FiLM = two 2-layer ReLU nets + ``gamma * x + beta``).  Layer-by-layer autograd costs one Python
autograd function per Linear plus one graph node per activation / product / sum: ~70 functions and ~450 backward nodes per
fitting step, i.e. more host time than the GPU needs for the kernels.  Here a network is ONE autograd function: its forward
launches the MFMA kernels of csrc/linear.hip and the elementwise pieces back to back and keeps the intermediates it needs,
its backward walks the layers in reverse — the same arithmetic in the same order (tested against the layer-by-layer modules:
outputs and every gradient), without the graph bookkeeping in between.

Activations ride on the GEMM kernels' epilogue (gsvc_linear_forward_ex):
  forward   ReLU in place; GELU stores the pre-activation and the activated value side by side (the backward needs the
            former, the next layer and its weight gradient the latter); tanh / sigmoid at the output
  backward  dX = (G W) * f'(saved) — the derivative of the PREVIOUS layer's activation is applied where dX is produced
            (GELU' from the saved pre-activation, ReLU' from the saved output).
FiLM's ``gamma * h + beta`` and its product rule run as streaming kernels (gsvc_film_forward / _backward): the epilogue forms
(GSVC_LIN_FILM / _FILM_GRAD, kept in the library) move the same bytes in 64-byte pieces of 400-byte rows, 25-35 us slower per call.
A GeneratorNet is two functions: the FiLM conditioning networks (they read the condition only and can be issued ahead of
everything that depends on the anchor features) and the trunk.
"""
from __future__ import annotations

import os

import torch

from . import _lib, switches

MFMA_MAX_DIM = 192
MIN_ROWS = 4096
# rows from which the multi-product launches (gsvc_linear_forward_shared_input / gsvc_linear_accumulate_many) are used: they hold a
# wave's row fragments / accumulators across the products but re-stage every product's weight image per workgroup — a fixed cost
# that a few thousand rows do not amortise (the rate sample's ~10 k rows: 124 us against 3 x 16 us as layer launches; same-process
# A/B of the fitting step 7.36 -> 7.28 ms): switches.MANY_MIN_ROWS, default 24 576.  A HOST-bound step (few visible rows: the host,
# not the GPU, sets its time — gsvc_amd.generate sets ``host_bound_step`` per generation pass) takes them at any size: there a
# launch saved is worth more than the microseconds its kernel loses (configs[3]: 4.86 -> 4.66 ms per step, round 5)
host_bound_step = False


def _many_min_rows():
    return 0 if host_bound_step else switches.MANY_MIN_ROWS


# epilogue codes of gsvc_linear_forward_ex (include/gsvc_hip.h)
EPI_NONE, EPI_RELU, EPI_GELU_DUAL, EPI_TANH, EPI_SIGMOID, EPI_MUL_GELU_GRAD, EPI_MUL_RELU_MASK, EPI_FILM, EPI_FILM_GRAD, EPI_ADD = range(10)

_workspaces = {}


def _workspace(dev, floats):
    """One weight-gradient workspace per device AND stream, grown on demand: the launches of a stream run in order, so they can
    share it — two streams cannot (the entropy networks' backward runs on the small-work stream next to the generators' on the
    step's stream: gsvc_amd/generate.py finish_deferred_rate)."""
    key = (dev.type, dev.index, _lib.current_stream(dev).value)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < floats:
        ws = torch.empty(max(floats, 1 << 22), device=dev, dtype=torch.float32)
        _workspaces[key] = ws
    return ws


def linear_ex(x, w, b=None, epi=EPI_NONE, aux1=None, aux2=None, y2=None, y3=None, w_in_out=False, out=None):
    """Y = epilogue(X W^T + b) through gsvc_linear_forward_ex; returns Y (and fills y2 / y3 where the epilogue has them).
    w_in_out: W is [K, N] (the dX = G W product of a backward pass)."""
    M, K = x.shape
    N = w.shape[1] if w_in_out else w.shape[0]
    y = out if out is not None else torch.empty(M, N, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().gsvc_linear_forward_ex(
        _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, K, N, int(w_in_out), int(epi), _lib.ptr(aux1), _lib.ptr(aux2),
        _lib.ptr(y2), _lib.ptr(y3), _lib.current_stream(x.device)), "gsvc_linear_forward_ex")
    return y


class WgradBatch:
    """Weight gradients of one network's backward pass: the layers' row-split products go out together at ``flush`` (one launch
    for up to twelve of them: gsvc_linear_wgrad_partial_many; their partial sums land in their own regions of the workspace),
    followed by ONE small launch that adds up the slots of all layers."""

    ARENA = 1 << 26     # floats (a product asks for 256 slots of N K + N; the batch launch uses 256 slots in all)

    def __init__(self, dev):
        import ctypes as C
        self.dev, self.C = dev, C
        self.ws = _workspace(dev, self.ARENA)
        self.off, self.jobs, self.keep = 0, [], []
        self.stream = _lib.current_stream(dev)

    def add(self, g, x, want_b=True):
        """dW = G^T X [N, K] and db = column sums of G [N]; the tensors are complete after flush()."""
        M, N = g.shape
        K = x.shape[1]
        need = int(_lib.lib().gsvc_linear_wgrad_workspace(N, K))
        if self.off + need > self.ws.numel():
            self.flush()
        buf = torch.empty(N * K + N, device=self.dev, dtype=torch.float32)
        gw = buf[:N * K].view(N, K)
        gb = buf[N * K:] if want_b else None
        if M == 0:
            buf.zero_()
            return gw, gb
        region = self.ws[self.off:self.off + need]
        self.jobs.append((g, x, region, need, bool(want_b), M, N, K, gw, gb))
        self.keep.append(buf)
        self.off += need
        return gw, gb

    def flush(self):
        if self.jobs:
            L = _lib.lib()
            n = len(self.jobs)
            part = (_lib.WgradPartialJobC * n)()
            for i, (g, x, region, need, want_b, M, N, K, gw, gb) in enumerate(self.jobs):
                part[i] = _lib.WgradPartialJobC(g.data_ptr(), x.data_ptr(), region.data_ptr(), M, need, int(want_b), N, K, 0)
            _lib.check(L.gsvc_linear_wgrad_partial_many(part, n, self.stream), "gsvc_linear_wgrad_partial_many")
            red = (_lib.WgradReduceJobC * n)()
            for i, (g, x, region, need, want_b, M, N, K, gw, gb) in enumerate(self.jobs):
                red[i] = _lib.WgradReduceJobC(region.data_ptr(), gw.data_ptr(), gb.data_ptr() if want_b else None, part[i].slots_used, N, K)
            _lib.check(L.gsvc_linear_wgrad_reduce_many(red, n, self.stream), "gsvc_linear_wgrad_reduce_many")
        self.off, self.jobs, self.keep = 0, [], []


def usable(x, *linears, min_rows=MIN_ROWS):
    """The fused path needs a tall fp32 CUDA matrix and layers that fit the weight-stationary kernel."""
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] >= min_rows and
            all(l.in_features <= MFMA_MAX_DIM and l.out_features <= MFMA_MAX_DIM and l.bias is not None for l in linears))


class _SeqGelu(torch.autograd.Function):
    """Linear -> GELU -> Linear -> GELU ... -> Linear (no activation after the last layer): mlp_deform and the two
    sub-networks of every EntropyParamsNet.  params = (W0, b0, W1, b1, ...)."""

    @staticmethod
    def forward(ctx, x, *params):
        x = x.contiguous()
        n = len(params) // 2
        acts, zs = [x], []
        h = x
        for i in range(n):
            w, b = params[2 * i].contiguous(), params[2 * i + 1].contiguous()
            if i + 1 < n:
                a = torch.empty(h.shape[0], w.shape[0], device=h.device, dtype=torch.float32)
                z = linear_ex(h, w, b, EPI_GELU_DUAL, y2=a)
                zs.append(z)
                acts.append(a)
                h = a
            else:
                h = linear_ex(h, w, b)
        ctx.n = n
        ctx.save_for_backward(*acts, *zs, *[params[2 * i] for i in range(n)])
        return h

    @staticmethod
    def backward(ctx, g):
        n = ctx.n
        saved = ctx.saved_tensors
        acts, zs, ws = saved[:n], saved[n:2 * n - 1], saved[2 * n - 1:]
        g = g.contiguous()
        grads = [None] * (2 * n)
        wg = WgradBatch(g.device)
        for i in range(n - 1, -1, -1):
            w = ws[i].contiguous()
            if ctx.needs_input_grad[1 + 2 * i]:
                grads[2 * i], grads[2 * i + 1] = wg.add(g, acts[i])
            if i > 0:
                g = linear_ex(g, w, None, EPI_MUL_GELU_GRAD, aux1=zs[i - 1], w_in_out=True)      # through layer i and GELU i-1
            elif ctx.needs_input_grad[0]:
                g = linear_ex(g, w, None, w_in_out=True)
            else:
                g = None
        wg.flush()
        return (g, *grads)


class _SeqGeluMany(torch.autograd.Function):
    """Several Linear -> GELU -> ... -> Linear networks that read the SAME input matrix — the six sub-networks of the three
    EntropyParamsNets all read the hash-grid feature of the visible anchors (reference scene/gaussian_model.py:198-232,
    1569-1597) — as ONE autograd function: the forward runs the chains back to back, the backward walks them in turn with the
    first layers' input gradients ACCUMULATED in place by the product's epilogue (GSVC_LIN_ADD: autograd summed them with five
    [rows, 192] additions per step) and every weight gradient of every chain in one batch (two launches + one slot reduce).
    sizes = layers per chain; params = (W, b) of every layer, chain after chain."""

    @staticmethod
    def forward(ctx, x, sizes, *params):
        x = x.contiguous()
        first = _first_layers_shared_input(x, sizes, params)      # [(z, a)] per chain, or None: layer by layer
        outs, saved, at = [], [x], 0
        for c, n in enumerate(sizes):
            h = x
            for i in range(n):
                w, b = params[at + 2 * i].contiguous(), params[at + 2 * i + 1].contiguous()
                if i == 0 and first is not None:
                    z, a = first[c]
                    saved += [z, a]
                    h = a
                elif i + 1 < n:
                    a = torch.empty(h.shape[0], w.shape[0], device=h.device, dtype=torch.float32)
                    z = linear_ex(h, w, b, EPI_GELU_DUAL, y2=a)
                    saved += [z, a]
                    h = a
                else:
                    h = linear_ex(h, w, b)
            outs.append(h)
            at += 2 * n
        ctx.sizes = tuple(sizes)
        ctx.set_materialize_grads(False)          # an unused output arrives as None, not as a matrix of zeros
        ctx.save_for_backward(*saved, *params)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        sizes = ctx.sizes
        n_saved = 1 + 2 * sum(n - 1 for n in sizes)
        t = ctx.saved_tensors
        saved, params = t[:n_saved], t[n_saved:]
        x = saved[0]
        grads = [None] * len(params)
        wg = WgradBatch(x.device)
        gx = None
        first = []          # (g, W) of the chains' first layers: their input gradients meet in gx
        at, sv = 0, 1
        for c, n in enumerate(sizes):
            g = gs[c]
            zs = [saved[sv + 2 * i] for i in range(n - 1)]
            acts = [x] + [saved[sv + 2 * i + 1] for i in range(n - 1)]
            sv += 2 * (n - 1)
            if g is not None:
                g = g.contiguous()
                for i in range(n - 1, -1, -1):
                    w = params[at + 2 * i].contiguous()
                    if ctx.needs_input_grad[2 + at + 2 * i]:
                        grads[at + 2 * i], grads[at + 2 * i + 1] = wg.add(g, acts[i])
                    if i > 0:
                        g = linear_ex(g, w, None, EPI_MUL_GELU_GRAD, aux1=zs[i - 1], w_in_out=True)
                    elif ctx.needs_input_grad[0]:
                        first.append((g, w))
            at += 2 * n
        if first:
            gx = _sum_of_products(first, x.shape[1])
        wg.flush()
        return (gx, None, *grads)


def _first_layers_shared_input(x, sizes, params):
    """The chains' first layers (Linear + GELU, every chain has a second layer) read the same x: one launch that keeps the rows'
    fragments in registers across the layers (gsvc_linear_forward_shared_input).  Returns [(z, a)] per chain, or None when the
    shapes are not the kernel's (the caller then runs layer by layer)."""
    M, K = x.shape
    if (switches.NO_SHARED_INPUT or not (2 <= len(sizes) <= 8) or any(n < 2 for n in sizes)
            or not _many_min_rows() <= M <= 65536
            or K > MFMA_MAX_DIM or K % 4 or x.data_ptr() % 16):
        return None
    ws, at = [], 0
    for n in sizes:
        w, b = params[at], params[at + 1]
        if w.shape[0] > 160 or not w.is_contiguous() or w.data_ptr() % 16 or not b.is_contiguous():
            return None
        ws.append((w, b))
        at += 2 * n
    jobs = (_lib.SharedInputJobC * len(ws))()
    out = []
    for i, (w, b) in enumerate(ws):
        z = torch.empty(M, w.shape[0], device=x.device, dtype=torch.float32)
        a = torch.empty_like(z)
        jobs[i] = _lib.SharedInputJobC(w.data_ptr(), b.data_ptr(), z.data_ptr(), a.data_ptr(), w.shape[0], 0)
        out.append((z, a))
    _lib.check(_lib.lib().gsvc_linear_forward_shared_input(_lib.ptr(x), M, K, jobs, len(ws), _lib.current_stream(x.device)),
               "gsvc_linear_forward_shared_input")
    return out


def _sum_of_products(pairs, N):
    """sum over (g [M, K], W [K, N]) of g W: one launch that keeps the rows' accumulators in registers across the products
    (gsvc_linear_accumulate_many: at most 8 products, M <= 65536), else product by product with the accumulate epilogue."""
    g0 = pairs[0][0]
    M = g0.shape[0]
    if (2 <= len(pairs) <= 8 and _many_min_rows() <= M <= 65536 and N <= MFMA_MAX_DIM and not switches.NO_ACCUM_MANY
            and all(g.shape[1] <= MFMA_MAX_DIM and g.shape[1] % 2 == 0 and g.is_contiguous() and w.is_contiguous()
                    and g.data_ptr() % 8 == 0 for g, w in pairs)):
        out = torch.empty(M, N, device=g0.device, dtype=torch.float32)
        jobs = (_lib.AccumJobC * len(pairs))()
        for i, (g, w) in enumerate(pairs):
            jobs[i] = _lib.AccumJobC(g.data_ptr(), w.data_ptr(), g.shape[1], 0)
        _lib.check(_lib.lib().gsvc_linear_accumulate_many(jobs, len(pairs), _lib.ptr(out), M, N, _lib.current_stream(g0.device)),
                   "gsvc_linear_accumulate_many")
        return out
    gx = None
    for g, w in pairs:
        if gx is None:
            gx = linear_ex(g, w, None, w_in_out=True)
        else:
            linear_ex(g, w, None, EPI_ADD, aux1=gx, w_in_out=True, out=gx)      # gx += g W, in place
    return gx


class _QuantStepNets(torch.autograd.Function):
    """The three quant_step networks (Linear -> GELU -> Linear(. -> 1)) on the same rows as one chain launch each way
    (csrc/mlp_chain.hip k_quant_nets_fwd / _bwd); their six weight gradients as one batch.  params = (W1, b1, W2, b2) x 3."""

    @staticmethod
    def forward(ctx, x, *params):
        import ctypes as C
        x = x.contiguous()
        params = [p.contiguous() for p in params]
        M, dev = x.shape[0], x.device
        H = params[0].shape[0]
        per = (M * H + 3) // 4 * 4                                                       # every matrix starts 16-byte aligned
        buf = torch.empty(6 * per + 3 * M, device=dev, dtype=torch.float32)              # z, a [M, H] x 3, q [M] x 3
        zs = [buf[i * per:i * per + M * H].view(M, H) for i in range(3)]
        as_ = [buf[(3 + i) * per:(3 + i) * per + M * H].view(M, H) for i in range(3)]
        qs = [buf[6 * per + i * M:6 * per + (i + 1) * M].view(M, 1) for i in range(3)]
        nets = (_lib.QuantStepNetC * 3)()
        for i in range(3):
            nets[i] = _lib.QuantStepNetC(*[t.data_ptr() for t in params[4 * i:4 * i + 4]])
        _lib.check(_lib.lib().gsvc_quant_step_nets_forward(nets, _lib.ptr(x), M, x.shape[1], H, _ptr_array(zs), _ptr_array(as_), _ptr_array(qs),
                                                           _lib.current_stream(dev)), "gsvc_quant_step_nets_forward")
        ctx.save_for_backward(x, buf, *params)
        ctx.geom = (M, H, per)
        ctx.set_materialize_grads(False)
        return tuple(qs)

    @staticmethod
    def backward(ctx, *gq):
        x, buf, *params = ctx.saved_tensors
        M, H, per = ctx.geom
        dev = x.device
        if all(g is None for g in gq):
            return (None,) * (1 + len(params))
        zs = [buf[i * per:i * per + M * H].view(M, H) for i in range(3)]
        as_ = [buf[(3 + i) * per:(3 + i) * per + M * H].view(M, H) for i in range(3)]
        gq = [None if g is None else g.contiguous().view(M, 1) for g in gq]
        dbuf = torch.empty(3 * per, device=dev, dtype=torch.float32)
        dzs = [dbuf[i * per:i * per + M * H].view(M, H) for i in range(3)]
        dX = torch.empty_like(x)
        nets = (_lib.QuantStepNetC * 3)()
        for i in range(3):
            nets[i] = _lib.QuantStepNetC(*[t.data_ptr() for t in params[4 * i:4 * i + 4]])
        import ctypes as C
        dq = (C.c_void_p * 3)(*[None if g is None else g.data_ptr() for g in gq])
        _lib.check(_lib.lib().gsvc_quant_step_nets_backward(nets, _ptr_array(zs), dq, M, x.shape[1], H, _ptr_array(dzs), _lib.ptr(dX),
                                                            _lib.current_stream(dev)), "gsvc_quant_step_nets_backward")
        grads = [None] * len(params)
        wg = WgradBatch(dev)
        for i in range(3):
            if gq[i] is None:
                continue
            if ctx.needs_input_grad[1 + 4 * i]:
                grads[4 * i], grads[4 * i + 1] = wg.add(dzs[i], x)
            if ctx.needs_input_grad[3 + 4 * i]:
                grads[4 * i + 2], grads[4 * i + 3] = wg.add(gq[i], as_[i])
        wg.flush()
        return (dX if ctx.needs_input_grad[0] else None, *grads)


def quant_step_nets_usable(x, nets):
    """The fused kernels are instantiated for 192 -> 50 -> 1 with biases, on a tall contiguous fp32 CUDA matrix."""
    if switches.NO_QUANT_CHAIN or switches.NO_MLP_CHAIN or len(nets) != 3:
        return False
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] >= MIN_ROWS and x.shape[1] == 192):
        return False
    for seq in nets:
        mods = list(seq)
        if len(mods) != 3 or not isinstance(mods[0], torch.nn.Linear) or not isinstance(mods[1], torch.nn.GELU) or not isinstance(mods[2], torch.nn.Linear):
            return False
        l1, l2 = mods[0], mods[2]
        if (l1.in_features, l1.out_features, l2.in_features, l2.out_features) != (192, 50, 50, 1) or l1.bias is None or l2.bias is None:
            return False
    return True


def quant_step_nets(x, nets):
    """(q_0, q_1, q_2) [M, 1]: the raw outputs of the three quant_step networks on the rows x."""
    params = []
    for seq in nets:
        mods = list(seq)
        params += [mods[0].weight, mods[0].bias, mods[2].weight, mods[2].bias]
    return _QuantStepNets.apply(x, *params)


def seq_gelu_many(x, chains):
    """chains: lists of nn.Linear (Linear -> GELU -> ... -> Linear each); returns one output per chain."""
    params = []
    for linears in chains:
        for l in linears:
            params += [l.weight, l.bias]
    return _SeqGeluMany.apply(x, tuple(len(c) for c in chains), *params)


ACT_NONE, ACT_TANH, ACT_SIGMOID = 0, 1, 2


class _FilmNets(torch.autograd.Function):
    """The two conditioning networks of a GeneratorNet's FiLM: gamma = fc_gamma1(relu(fc_gamma0(c))), beta likewise (reference
    scene/gaussian_model.py:150-166).  They read the condition only (frame time and z embedding), not the anchor features, so a
    fitting step can issue them before anything that waits for the entropy context — four large GEMMs that keep the GPU busy
    while the host queues the context's many small launches.  params = fc_gamma0 (W, b), fc_gamma1, fc_beta0, fc_beta1."""

    @staticmethod
    def forward(ctx, condition, *params):
        condition = condition.contiguous()
        Wg0, bg0, Wg1, bg1, Wb0, bb0, Wb1, bb1 = [p.contiguous() for p in params]
        cb = linear_ex(condition, Wb0, bb0, EPI_RELU)
        beta = linear_ex(cb, Wb1, bb1)
        cg = linear_ex(condition, Wg0, bg0, EPI_RELU)
        gamma = linear_ex(cg, Wg1, bg1)
        ctx.save_for_backward(condition, cb, cg, Wg0, Wg1, Wb0, Wb1)
        return gamma, beta

    @staticmethod
    def backward(ctx, gg, gbeta):
        condition, cb, cg, Wg0, Wg1, Wb0, Wb1 = ctx.saved_tensors
        gg, gbeta = gg.contiguous(), gbeta.contiguous()
        P = [None] * 8
        wg = WgradBatch(gg.device)
        P[6], P[7] = wg.add(gbeta, cb)
        gcb = linear_ex(gbeta, Wb1, None, EPI_MUL_RELU_MASK, aux1=cb, w_in_out=True)
        P[4], P[5] = wg.add(gcb, condition)
        P[2], P[3] = wg.add(gg, cg)
        gcg = linear_ex(gg, Wg1, None, EPI_MUL_RELU_MASK, aux1=cg, w_in_out=True)
        P[0], P[1] = wg.add(gcg, condition)
        gcond = None
        if ctx.needs_input_grad[0]:
            gcond = linear_ex(gcb, Wb0, None, w_in_out=True)
            gcond += linear_ex(gcg, Wg0, None, w_in_out=True)
        wg.flush()
        return (gcond, *P)


class _GeneratorTrunk(torch.autograd.Function):
    """out_act(out_linear(gamma * linear2(GELU(linear1(feature))) + beta)) given the FiLM pair (reference
    scene/gaussian_model.py:168-196).  params = linear1 (W, b), linear2, out_linear."""

    @staticmethod
    def forward(ctx, feature, gamma, beta, act, *params):
        feature, gamma, beta = feature.contiguous(), gamma.contiguous(), beta.contiguous()
        W1, b1, W2, b2, W3, b3 = [p.contiguous() for p in params]
        M, dev = feature.shape[0], feature.device
        a1 = torch.empty(M, W1.shape[0], device=dev, dtype=torch.float32)
        z1 = linear_ex(feature, W1, b1, EPI_GELU_DUAL, y2=a1)
        h = linear_ex(a1, W2, b2)
        if h.numel() % 4 == 0:
            # the FiLM combine as a streaming kernel (whole-line accesses; as a GEMM epilogue, EPI_FILM, the same traffic moves
            # in 64-byte pieces of 400-byte rows and costs ~25 us more per call)
            x3 = torch.empty_like(h)
            _lib.check(_lib.lib().gsvc_film_forward(_lib.ptr(gamma), _lib.ptr(h), _lib.ptr(beta), _lib.ptr(x3), h.numel(),
                                                    _lib.current_stream(dev)), "gsvc_film_forward")
        else:
            x3 = torch.addcmul(beta, gamma, h)
        y = linear_ex(x3, W3, b3, (EPI_NONE, EPI_TANH, EPI_SIGMOID)[act])
        ctx.act = act
        ctx.save_for_backward(feature, z1, a1, h, gamma, x3, y, W1, W2, W3)
        return y

    @staticmethod
    def backward(ctx, gy):
        feature, z1, a1, h, gamma, x3, y, W1, W2, W3 = ctx.saved_tensors
        need = ctx.needs_input_grad
        gy = gy.contiguous()
        if ctx.act == ACT_TANH:
            go = torch.ops.aten.tanh_backward(gy, y)
        elif ctx.act == ACT_SIGMOID:
            go = torch.ops.aten.sigmoid_backward(gy, y)
        else:
            go = gy
        P = [None] * 6
        wg = WgradBatch(gy.device)
        P[4], P[5] = wg.add(go, x3)
        dev = feature.device
        gbeta = linear_ex(go, W3, None, w_in_out=True)          # d x3 = d beta
        if h.numel() % 4 == 0:
            gg, gh = torch.empty_like(h), torch.empty_like(h)
            _lib.check(_lib.lib().gsvc_film_backward(_lib.ptr(gbeta), _lib.ptr(h), _lib.ptr(gamma), _lib.ptr(gg), _lib.ptr(gh),
                                                     gg.numel(), _lib.current_stream(dev)), "gsvc_film_backward")
        else:
            gg, gh = gbeta * h, gbeta * gamma
        P[2], P[3] = wg.add(gh, a1)
        gz1 = linear_ex(gh, W2, None, EPI_MUL_GELU_GRAD, aux1=z1, w_in_out=True)
        P[0], P[1] = wg.add(gz1, feature)
        gfeat = linear_ex(gz1, W1, None, w_in_out=True) if need[0] else None
        wg.flush()
        return (gfeat, gg if need[1] else None, gbeta if need[2] else None, None, *P)


def film_nets(film, condition):
    """(gamma, beta) of a FiLM module for a tall CUDA condition matrix (one autograd function)."""
    params = []
    for l in (film.fc_gamma0, film.fc_gamma1, film.fc_beta0, film.fc_beta1):
        params += [l.weight, l.bias]
    return _FilmNets.apply(condition, *params)


def generator(net, feature, condition=None, film=None):
    """GeneratorNet forward as two autograd functions; ``film``: a (gamma, beta) pair computed ahead with ``film_nets``."""
    act = {"Tanh": ACT_TANH, "Sigmoid": ACT_SIGMOID, "Identity": ACT_NONE}[type(net.out_act).__name__]
    gamma, beta = film if film is not None else film_nets(net.film, condition)
    params = []
    for l in (net.linear1, net.linear2, net.out_linear):
        params += [l.weight, l.bias]
    return _GeneratorTrunk.apply(feature, gamma, beta, act, *params)


def seq_gelu(x, linears):
    params = []
    for l in linears:
        params += [l.weight, l.bias]
    return _SeqGelu.apply(x, *params)


# ---------------------------------------------------------------------------------------------------------------------------
# Whole-network chain kernels (csrc/mlp_chain.hip): the three GeneratorNets and mlp_deform of a generation pass as ONE autograd
# function.  Forward: 2 launches per network (FiLM nets | trunk, layers 1-2 | 3-5), a 16-row block's activations stay in
# registers across the layers; backward: 2 chain launches per network + the row-split weight-gradient kernels + one slot reduce,
# all issued from C.  The anchor features feed all four networks: their gradient is accumulated in place by the kernels
# (no fan-out additions).  Reference: scene/gaussian_model.py:150-196, 468-489; ortho_gaussian_renderer/guassian.py:251-273.
GEN_FIELDS = ("W1", "b1", "W2", "b2", "W3", "b3", "Wg0", "bg0", "Wg1", "bg1", "Wb0", "bb0", "Wb1", "bb1")


def _generator_params(net):
    f = net.film
    out = []
    for l in (net.linear1, net.linear2, net.out_linear, f.fc_gamma0, f.fc_gamma1, f.fc_beta0, f.fc_beta1):
        out += [l.weight, l.bias]
    return out


def _act_code(net):
    return {"Tanh": ACT_TANH, "Sigmoid": ACT_SIGMOID, "Identity": ACT_NONE}.get(type(net.out_act).__name__)


def chain_usable(feat, cond, gens, deform_linears):
    """The chain kernels exist for the production widths only (feature 50, condition 66, hidden 100, outputs 10 / 30 / 70 and
    30) and need tall contiguous fp32 CUDA matrices; the condition must not require a gradient."""
    if switches.NO_MLP_CHAIN or switches.NO_MLP_FUSED:
        return False
    if not (feat.is_cuda and cond.is_cuda and feat.dim() == cond.dim() == 2 and feat.dtype == cond.dtype == torch.float32
            and feat.shape[0] == cond.shape[0] >= MIN_ROWS and not cond.requires_grad):
        return False
    if feat.shape[1] != 50 or cond.shape[1] != 66:
        return False
    for net in gens:
        if _act_code(net) is None or net.linear1.out_features != 100 or net.out_linear.out_features not in (10, 30, 70):
            return False
        if any(l.bias is None for l in (net.linear1, net.linear2, net.out_linear, net.film.fc_gamma0, net.film.fc_gamma1,
                                        net.film.fc_beta0, net.film.fc_beta1)):
            return False
    if len(deform_linears) != 5 or any(l.bias is None for l in deform_linears):
        return False
    dims = [(l.in_features, l.out_features) for l in deform_linears]
    return dims == [(116, 100), (100, 100), (100, 100), (100, 100), (100, 30)]


def _gen_desc(params, act, out_dim):
    d = _lib.GeneratorNetC()
    for name, p in zip(GEN_FIELDS, params):
        setattr(d, name, p.data_ptr())
    d.feat_dim, d.cond_dim, d.hidden_dim, d.out_dim, d.out_act = 50, 66, 100, out_dim, act
    return d


def _deform_desc(params):
    d = _lib.DeformNetC()
    for i in range(5):
        d.W[i], d.b[i] = params[2 * i].data_ptr(), params[2 * i + 1].data_ptr()
    d.feat_dim, d.cond_dim, d.hidden_dim, d.out_dim = 50, 66, 100, 30
    return d


def _film_desc(film):
    """gsvc_film_rows of (cond_film [Mf, 66], row_of [M] int32, src_a [Mf] int32, src_b [Mf] int32), or NULL."""
    if film is None:
        return None
    import ctypes as C
    cond_f, row_of, src_a, src_b = film
    d = _lib.FilmRowsC(int(cond_f.shape[0]), cond_f.data_ptr(), row_of.data_ptr(), src_a.data_ptr(), src_b.data_ptr())
    return C.byref(d)


# ---- the weight gradients' own stream (gsvc_set_wgrad_stream) ----------------------------------------------------------------
# Trainer.step opens ``wgrad_overlap`` around its backward: inside it _GenerateAll.backward's dW products run on a side stream
# behind the chain kernels while the step's stream carries the feature gradient on; leaving it, the step's stream waits for them.
_WGRAD_STREAM = {}
_wgrad_active = None      # the side stream while a ``wgrad_overlap`` block is open
_wgrad_calls = 0          # _GenerateAll.backward calls inside the open block: only the FIRST one's products leave the step's stream — a
                          # second call's gradients are ADDED to the first's by autograd, on the step's stream, before the block's end


class wgrad_overlap:
    def __init__(self, dev):
        self.dev = dev

    def __enter__(self):
        global _wgrad_active
        if self.dev.type != "cuda" or switches.NO_WGRAD_OVERLAP or _wgrad_active is not None:
            self.on = False
            return self
        st = _WGRAD_STREAM.get(self.dev.index)
        if st is None:
            st = _WGRAD_STREAM[self.dev.index] = torch.cuda.Stream(device=self.dev)
        _lib.check(_lib.lib().gsvc_set_wgrad_stream(st.cuda_stream), "gsvc_set_wgrad_stream")
        global _wgrad_calls
        _wgrad_active, _wgrad_calls, self.on = st, 0, True
        return self

    def __exit__(self, *exc):
        global _wgrad_active
        if self.on:
            _lib.check(_lib.lib().gsvc_set_wgrad_stream(None), "gsvc_set_wgrad_stream")
            torch.cuda.current_stream(self.dev).wait_stream(_wgrad_active)      # the optimizer, the reducer: behind the last product
            _wgrad_active = None
        return False


def _ptr_array(tensors):
    import ctypes as C
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class _GenerateAll(torch.autograd.Function):
    """(opacity, color, cov, deform) outputs of the three GeneratorNets and mlp_deform for the rows (feature, condition).
    params = 3 x 14 generator tensors (GEN_FIELDS order) + 5 x (W, b) of mlp_deform; acts = the generators' output activations.
    The three generators run as ONE pair of launches each way (gsvc_generators_*: workgroup b serves network b % 3)."""

    @staticmethod
    def forward(ctx, feat, cond, acts, film, *params):
        import ctypes as C
        feat, cond = feat.contiguous(), cond.contiguous()
        params = [p.contiguous() for p in params]
        M, dev = feat.shape[0], feat.device
        L, st = _lib.lib(), _lib.current_stream(dev)
        nets = (_lib.GeneratorNetC * 3)()
        outs, saved = [], []
        Mf = int(film[0].shape[0]) if film is not None else 0
        for g in range(3):
            pg = params[14 * g:14 * (g + 1)]
            d = _gen_desc(pg, acts[g], pg[4].shape[0])
            C.memmove(C.byref(nets[g]), C.byref(d), C.sizeof(d))
            saved.append(torch.empty(int(L.gsvc_generator_saved_floats(C.byref(d), M, Mf)), device=dev, dtype=torch.float32))
            outs.append(torch.empty(M, pg[4].shape[0], device=dev, dtype=torch.float32))
        _lib.check(L.gsvc_generators_forward(nets, 3, _lib.ptr(feat), _lib.ptr(cond), M, _film_desc(film), _ptr_array(saved),
                                             _ptr_array(outs), st), "gsvc_generators_forward")
        pd = params[42:52]
        dd = _deform_desc(pd)
        sv = torch.empty(int(L.gsvc_deform_saved_floats(C.byref(dd), M)), device=dev, dtype=torch.float32)
        y = torch.empty(M, 30, device=dev, dtype=torch.float32)
        _lib.check(L.gsvc_deform_forward(C.byref(dd), _lib.ptr(feat), _lib.ptr(cond), M, _lib.ptr(sv), _lib.ptr(y), st), "gsvc_deform_forward")
        outs.append(y)
        saved.append(sv)
        ctx.acts = tuple(acts)
        ctx.film = film          # index tensors and the FiLM rows' condition: no gradient flows to them
        ctx.save_for_backward(feat, cond, *outs[:3], *saved, *params)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gys):
        import ctypes as C
        t = ctx.saved_tensors
        feat, cond, ys, saved, params = t[0], t[1], t[2:5], t[5:9], t[9:]
        M, dev = feat.shape[0], feat.device
        L, st = _lib.lib(), _lib.current_stream(dev)
        need = ctx.needs_input_grad
        # an output nobody used has no gradient: it contributes zeros (rare: every training mode reads all four)
        gys = [g.contiguous() if g is not None else torch.zeros(M, n, device=dev, dtype=torch.float32) for g, n in
               zip(gys, [y.shape[1] for y in ys] + [30])]
        flat = torch.empty(sum(p.numel() for p in params), device=dev, dtype=torch.float32)      # every weight / bias gradient, back to back
        grads, at = [], 0
        for p in params:
            grads.append(flat[at:at + p.numel()].view(p.shape))
            at += p.numel()
        nets = (_lib.GeneratorNetC * 3)()
        gds = (_lib.GeneratorGradsC * 3)()
        total = 0
        for g in range(3):
            pg = params[14 * g:14 * (g + 1)]
            d = _gen_desc(pg, ctx.acts[g], pg[4].shape[0])
            C.memmove(C.byref(nets[g]), C.byref(d), C.sizeof(d))
            for name, v in zip(GEN_FIELDS, grads[14 * g:14 * (g + 1)]):
                setattr(gds[g], name, v.data_ptr())
            total += (int(L.gsvc_generator_scratch_floats(C.byref(d), M, int(ctx.film[0].shape[0]) if ctx.film is not None else 0)) + 3) // 4 * 4
        dd = _deform_desc(params[42:52])
        side = _wgrad_active if (_wgrad_active is not None and dev.type == "cuda") else None
        later_call = False
        if side is not None:
            global _wgrad_calls
            _wgrad_calls += 1
            if _wgrad_calls > 1:
                # (the per-render form of a step: several generation passes) the sum of this call's and the first call's weight gradients
                # is formed on this stream: wait for the first call's products, keep this call's here
                torch.cuda.current_stream(dev).wait_stream(side)
                _lib.check(L.gsvc_set_wgrad_stream(None), "gsvc_set_wgrad_stream")
                later_call, later_side, side = True, side, None
        if side is None:
            total = max(total, int(L.gsvc_deform_scratch_floats(C.byref(dd), M)))      # the deformation network reuses the generators' scratch
            scratch = scratch_d = torch.empty(total, device=dev, dtype=torch.float32)
        else:
            # the generators' weight gradients are still reading their scratch (on the side stream) when the deformation network's chain
            # kernels write theirs: two buffers; and everything those products touch outlives this function on THAT stream
            scratch = torch.empty(total, device=dev, dtype=torch.float32)
            scratch_d = torch.empty(int(L.gsvc_deform_scratch_floats(C.byref(dd), M)), device=dev, dtype=torch.float32)
            for x in (feat, cond, *saved, scratch, scratch_d, flat, gys[3], *((ctx.film[0],) if ctx.film is not None else ())):
                x.record_stream(side)
        hold = side is not None
        if hold:
            # both networks' chain kernels first, then every product beside what follows: a chain workgroup needs its whole CU, and the
            # generators' products queued ahead of mlp_deform's chain kernels made those wait for them (6.81 -> 6.78 ms per step)
            _lib.check(L.gsvc_wgrad_hold(1), "gsvc_wgrad_hold")
        F_ = feat.shape[1]
        per = (M * F_ + 3) // 4 * 4                                                      # every buffer starts 16-byte aligned
        gflat = torch.empty(4 * per, device=dev, dtype=torch.float32)                   # three generators' + the sum
        gfeats = [gflat[i * per:i * per + M * F_].view(M, F_) for i in range(4)]
        gen_gf = gfeats[:3]
        try:
            _lib.check(L.gsvc_generators_backward(nets, 3, _lib.ptr(feat), _lib.ptr(cond), M, _film_desc(ctx.film), _ptr_array(saved[:3]),
                                                  _ptr_array(ys), _ptr_array(gys[:3]), _lib.ptr(scratch), _ptr_array(gen_gf), gds, st),
                       "gsvc_generators_backward")
            gd = _lib.DeformGradsC()
            for i in range(5):
                gd.W[i], gd.b[i] = grads[42 + 2 * i].data_ptr(), grads[43 + 2 * i].data_ptr()
            # the deformation network's backward runs behind the generators' (same stream): it adds their three feature gradients to
            # its own in the pass that writes gfeat
            _lib.check(L.gsvc_deform_backward(C.byref(dd), _lib.ptr(feat), _lib.ptr(cond), M, _lib.ptr(saved[3]), _lib.ptr(gys[3]),
                                              _lib.ptr(scratch_d), _lib.ptr(gfeats[3]), 0, _ptr_array(gen_gf), 3, C.byref(gd), st),
                       "gsvc_deform_backward")
        finally:
            if hold:      # (also after an error: nothing stays held, the library leaves the hold)
                rc = L.gsvc_wgrad_flush(st)
                L.gsvc_wgrad_hold(0)
            if later_call:
                L.gsvc_set_wgrad_stream(later_side.cuda_stream)
        if hold:
            _lib.check(rc, "gsvc_wgrad_flush")
        return (gfeats[3] if need[0] else None, None, None, None, *grads)


def generate_all(gens, deform_linears, feat, cond, film=None):
    """(opacity_raw, color, scale_rot, neural_offset) = the three generators and mlp_deform on (feat, cond) rows.
    ``film`` = (cond_film, row_of, src_a, src_b): the generators' FiLM networks run on the rows ``cond_film`` only (one per
    (frame, anchor): the opposite views of a frame share the condition) — include/gsvc_hip.h gsvc_film_rows."""
    params = []
    for net in gens:
        params += _generator_params(net)
    for l in deform_linears:
        params += [l.weight, l.bias]
    if film is not None:
        film = tuple(t.contiguous() for t in film)
        assert film[0].dtype == torch.float32 and all(t.dtype == torch.int32 for t in film[1:])
    acts = tuple(_act_code(n) for n in gens)
    if not torch.is_grad_enabled():
        return _generate_all_inference(feat, cond, acts, film, params)
    return _GenerateAll.apply(feat, cond, acts, film, *params)


def _generate_all_inference(feat, cond, acts, film, params):
    """generate_all without autograd (the decoder's render loop, evaluation): the chain kernels leave out every store only a
    backward would read (gsvc_generators_forward_inference, gsvc_deform_forward_inference)."""
    import ctypes as C
    feat, cond = feat.contiguous(), cond.contiguous()
    params = [p.detach().contiguous() for p in params]
    M, dev = feat.shape[0], feat.device
    L, st = _lib.lib(), _lib.current_stream(dev)
    nets = (_lib.GeneratorNetC * 3)()
    outs, scratch = [], []
    Mf = int(film[0].shape[0]) if film is not None else 0
    for g in range(3):
        pg = params[14 * g:14 * (g + 1)]
        d = _gen_desc(pg, acts[g], pg[4].shape[0])
        C.memmove(C.byref(nets[g]), C.byref(d), C.sizeof(d))
        scratch.append(torch.empty(int(L.gsvc_generator_inference_floats(C.byref(d), M, Mf)), device=dev, dtype=torch.float32))
        outs.append(torch.empty(M, pg[4].shape[0], device=dev, dtype=torch.float32))
    _lib.check(L.gsvc_generators_forward_inference(nets, 3, _lib.ptr(feat), _lib.ptr(cond), M, _film_desc(film), _ptr_array(scratch),
                                                   _ptr_array(outs), st), "gsvc_generators_forward_inference")
    dd = _deform_desc(params[42:52])
    sc = torch.empty(max(M, 1) * 100, device=dev, dtype=torch.float32)
    y = torch.empty(M, 30, device=dev, dtype=torch.float32)
    _lib.check(L.gsvc_deform_forward_inference(C.byref(dd), _lib.ptr(feat), _lib.ptr(cond), M, _lib.ptr(sc), _lib.ptr(y), st),
               "gsvc_deform_forward_inference")
    outs.append(y)
    return tuple(outs)
