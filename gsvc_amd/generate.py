"""Anchor -> neural Gaussians for one frame (host orchestration around the MLPs, the hash grid and the rate
kernel).

Same interface and values as reference ortho_gaussian_renderer/guassian.py: ``GenerateMode`` (:21-26),
``RatePack`` (:34-39), ``GeneratedGaussians`` (:42-56), ``calc_sampled_rate`` (:73-132) and
``generate_neural_gaussians`` (:134-310).  The random draws (quantisation noise for feat / scaling / offsets,
then the 5 % anchor sample) happen in the reference's order so a seeded run reproduces its fixtures.

What is organised differently: the visible-anchor subset is taken once with an index gather, the
"opacity > 0" compaction uses one index list for all per-Gaussian tensors, and getters that the reference
re-evaluates over all anchors several times per call are evaluated once.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass
from enum import Enum

import torch

from . import switches
from .encodings import STE_multistep


class GenerateMode(Enum):
    TRAINING_FULL_PRECISION = 0
    TRAINING_QUANTIZED = 1
    TRAINING_ENTROPY = 2
    TRAININ_STE_ENTROPY = 3  # (sic) the reference's spelling is API
    DECODING_AS_IS = 4


@dataclass
class RatePack:
    bit_per_param: torch.Tensor = None
    bit_per_feat_param: torch.Tensor = None
    bit_per_scaling_param: torch.Tensor = None
    bit_per_offsets_param: torch.Tensor = None


@dataclass
class GeneratedGaussians:
    xyz: torch.Tensor
    color: torch.Tensor
    opacity: torch.Tensor
    scaling: torch.Tensor
    rot: torch.Tensor
    neural_opacity: torch.Tensor = None
    visable_mask: torch.Tensor = None  # (sic)
    mask: torch.Tensor = None
    bit_per_param: torch.Tensor = None
    bit_per_feat_param: torch.Tensor = None
    bit_per_scaling_param: torch.Tensor = None
    bit_per_offsets_param: torch.Tensor = None
    concatenated_all: torch.Tensor = None
    time_sub: float = None
    # extras of the un-compacted ("dense") batched path, None otherwise
    visible_index: torch.Tensor = None   # int64 indices of the visible anchors (what visable_mask.nonzero() gives)
    world_xyz: torch.Tensor = None       # anchor + offsets * scaling[:, :3] before the bound clamp, per Gaussian
    batch: object = None                 # the renders' shared un-split tensors (scaling, neural_opacity, mask, seg_offsets)


BASE_Q_FEAT, BASE_Q_SCALING, BASE_Q_OFFSETS = 1, 0.001, 0.2
SAMPLE_RATE = 0.05


def calc_sampled_rate(pc, visible_mask, feat, grid_scaling, grid_offsets, Q_feat, Q_scaling, Q_offsets, entropy_context):
    """Bits per parameter estimated on a 5 % Bernoulli sample of the visible anchors that still own at least
    one live offset, scaled by the fraction of such anchors."""
    K = pc.n_offsets
    vis = _as_index(visible_mask)
    offset_masks = _visible_mask(pc, vis)
    with torch.no_grad():
        mask_anchor = (torch.sum(offset_masks, dim=1)[:, 0]) > 0     # get_mask_anchor restricted to the visible anchors
    keep_rate = (mask_anchor.sum() / mask_anchor.numel()).detach()
    # same draw as the reference's rand_like(anchor[:, 0]) (one uniform per visible anchor)
    chosen = (torch.rand_like(feat[:, 0]) <= SAMPLE_RATE) & mask_anchor.to(torch.bool)
    sel = chosen.nonzero(as_tuple=False).squeeze(1)
    take = lambda t: t.index_select(0, sel)  # noqa: E731
    ec = entropy_context
    bit_feat = pc.entropy_gaussian(take(feat), take(ec.mean_feat), take(ec.scale_feat), take(Q_feat), pc._anchor_feat.mean())
    bit_scaling = pc.entropy_gaussian(take(grid_scaling), take(ec.mean_scaling), take(ec.scale_scaling), take(Q_scaling),
                                      pc.get_scaling.mean())
    bit_offsets = pc.entropy_gaussian(take(grid_offsets).view(-1, 3 * K), take(ec.mean_offsets), take(ec.scale_offsets),
                                      take(Q_offsets), pc._offset.mean())
    bit_offsets = bit_offsets * take(offset_masks).repeat(1, 1, 3).view(-1, 3 * K)
    sf, ss, so = bit_feat.sum(), bit_scaling.sum(), bit_offsets.sum()
    nf, ns, no = bit_feat.numel(), bit_scaling.numel(), bit_offsets.numel()
    return RatePack(bit_per_param=(sf + ss + so) / (nf + ns + no) * keep_rate,
                    bit_per_feat_param=sf / nf * keep_rate,
                    bit_per_scaling_param=ss / ns * keep_rate,
                    bit_per_offsets_param=so / no * keep_rate)


def _visible_mask(pc, vis):
    """pc.get_mask[vis] evaluated on the gathered rows only (same values: the activation is elementwise)."""
    raw = pc._mask.index_select(0, vis)
    if pc.decoded_version:
        return raw
    s = torch.sigmoid(raw)
    return ((s > 0.01).float() - s).detach() + s


def _visible_scaling(pc, vis):
    raw = pc._scaling.index_select(0, vis)
    return raw if pc.decoded_version else 1.0 * pc.scaling_activation(raw)


def det_scatter_rows(idx, src, D, out=None):
    """dst[idx[i]] += src[i] with every target's rows added in list order (GSVC_DETERMINISTIC): a stable sort of the targets, then one
    thread per (target, channel) walks its run (csrc/generate.hip k_segment_rows_sum) — no float atomics, the same bits every run.
    ``out``: a [D, C] tensor to add to (each touched row receives ONE add of its run's sum); else a zero-filled one is returned."""
    from . import _lib
    n = int(idx.shape[0])
    C_ = max(int(torch.Size(src.shape[1:]).numel()), 1)          # (an empty list still has a row width)
    src2 = src.reshape(n, C_).contiguous()
    dst = out if out is not None else torch.zeros(D, C_, device=src.device, dtype=torch.float32)
    if n:
        sorted_idx, order = torch.sort(idx.contiguous(), stable=True)
        _lib.check(_lib.lib().gsvc_segment_rows_sum(_lib.ptr(src2), _lib.ptr(order), _lib.ptr(sorted_idx), n, C_, _lib.ptr(dst),
                                                    1 if out is not None else 0, _lib.current_stream(src.device)), "gsvc_segment_rows_sum")
    return dst


class IndexRows(torch.autograd.Function):
    """t.index_select(0, idx) whose backward adds the rows of a repeated index in list order (GSVC_DETERMINISTIC; torch's own
    backward is an index_add_ with float atomics)."""

    @staticmethod
    def forward(ctx, t, idx):
        ctx.save_for_backward(idx)
        ctx.shape = t.shape
        return t.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return det_scatter_rows(idx, g, ctx.shape[0]).view(ctx.shape), None


class _GatherFeat(torch.autograd.Function):
    """feat = _anchor_feat[vis] (csrc/generate.hip k_gather_rows with the feature group only; the scatter-add is its backward).
    ``seen`` / ``rank`` (a StepPlan's flattened view masks and their inclusive scan) select the atomic-free backward."""

    @staticmethod
    def forward(ctx, feat_p, vis, seen=None, rank=None):
        from . import _lib
        feat_p, vis = feat_p.contiguous(), vis.contiguous()
        rows, F = vis.shape[0], feat_p.shape[1]
        feat = torch.empty(rows, F, device=feat_p.device, dtype=torch.float32)
        _lib.check(_lib.lib().gsvc_gather_rows_forward(_lib.ptr(feat_p), None, None, None, _lib.ptr(vis), rows, F, 0, 0, 0, _lib.ptr(feat),
                                                       None, None, None, _lib.current_stream(feat_p.device)), "gsvc_gather_rows_forward")
        ctx.save_for_backward(vis, seen, rank)
        ctx.shape = feat_p.shape
        return feat

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        vis, seen, rank = ctx.saved_tensors
        A, F = ctx.shape
        if seen is not None:
            d = torch.empty(ctx.shape, device=vis.device, dtype=torch.float32)
            _lib.check(_lib.lib().gsvc_gather_rows_backward_ranked(None, None, _lib.ptr(seen), _lib.ptr(rank), seen.numel() // A, A, F, 0, 0, 0,
                                                                   _lib.ptr(g.contiguous()), None, None, None, _lib.ptr(d), None, None,
                                                                   None, _lib.current_stream(vis.device)),
                       "gsvc_gather_rows_backward_ranked")
            return d, None, None, None
        if switches.DETERMINISTIC:          # rows of no plan (a step's first / repeated attempt): a sorted scatter instead of float atomics
            return det_scatter_rows(vis, g, A), None, None, None
        d = torch.zeros(ctx.shape, device=vis.device, dtype=torch.float32)
        _lib.check(_lib.lib().gsvc_gather_rows_backward(None, None, _lib.ptr(vis), vis.shape[0], ctx.shape[1], 0, 0, 0,
                                                        _lib.ptr(g.contiguous()), None, None, None, _lib.ptr(d), None, None, None,
                                                        _lib.current_stream(vis.device)), "gsvc_gather_rows_backward")
        return d, None, None, None


class _GatherRows(torch.autograd.Function):
    """(offsets, scaling, mask) of the rows ``vis`` from the per-anchor parameter tensors with the getters' activations applied
    (csrc/generate.hip k_gather_rows): one launch each way instead of three gathers + activations and three scatter-adds.  The
    features are gathered apart (_GatherFeat): their gradient is complete only after the MLPs' backward, these three right
    behind the rasterizer's — a data-parallel step starts their all-reduce that much earlier (gsvc_amd.dist.GradReducer)."""

    @staticmethod
    def forward(ctx, offset_p, scaling_p, mask_p, vis, decoded, seen=None, rank=None):
        from . import _lib
        dev = offset_p.device
        offset_p, scaling_p, mask_p, vis = (t.contiguous() for t in (offset_p, scaling_p, mask_p, vis))
        rows, K, S = vis.shape[0], offset_p.shape[1], scaling_p.shape[1]
        f = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)  # noqa: E731
        off, scal, mask = f(rows, K, 3), f(rows, S), f(rows, K, 1)
        _lib.check(_lib.lib().gsvc_gather_rows_forward(None, _lib.ptr(offset_p), _lib.ptr(scaling_p), _lib.ptr(mask_p),
                                                       _lib.ptr(vis), rows, 0, K, S, int(decoded), None, _lib.ptr(off),
                                                       _lib.ptr(scal), _lib.ptr(mask), _lib.current_stream(dev)),
                   "gsvc_gather_rows_forward")
        ctx.save_for_backward(scaling_p, mask_p, vis, seen, rank)
        ctx.dims = (offset_p.shape, K, S, bool(decoded))
        ctx.set_materialize_grads(False)      # an output nothing differentiates (detached STE modes) gives no gradient, not zeros
        return off, scal, mask

    @staticmethod
    def backward(ctx, g_off, g_scal, g_mask):
        from . import _lib
        scaling_p, mask_p, vis, seen, rank = ctx.saved_tensors
        oshape, K, S, decoded = ctx.dims
        dev = vis.device
        need = ctx.needs_input_grad
        c = lambda g: g.contiguous() if g is not None else None  # noqa: E731
        shapes = [oshape if need[0] and g_off is not None else None, scaling_p.shape if need[1] and g_scal is not None else None,
                  mask_p.shape if need[2] and g_mask is not None else None]
        g_off, g_scal, g_mask = c(g_off), c(g_scal), c(g_mask)
        if seen is not None:      # rows = a step plan's R ascending lists: every output element is written once, no atomics
            A = oshape[0]
            d_off, d_scal, d_mask = (torch.empty(sh, device=dev, dtype=torch.float32) if sh is not None else None for sh in shapes)
            _lib.check(_lib.lib().gsvc_gather_rows_backward_ranked(_lib.ptr(scaling_p), _lib.ptr(mask_p), _lib.ptr(seen), _lib.ptr(rank),
                                                                   seen.numel() // A, A, 0, K, S, int(decoded), None, _lib.ptr(g_off),
                                                                   _lib.ptr(g_scal), _lib.ptr(g_mask), None, _lib.ptr(d_off),
                                                                   _lib.ptr(d_scal), _lib.ptr(d_mask), _lib.current_stream(dev)),
                       "gsvc_gather_rows_backward_ranked")
            return d_off, d_scal, d_mask, None, None, None, None
        if switches.DETERMINISTIC:
            A = oshape[0]
            d_off = det_scatter_rows(vis, g_off, A).view(oshape) if shapes[0] is not None else None
            d_scal = d_mask = None
            if shapes[1] is not None:
                d_scal = det_scatter_rows(vis, g_scal, A).view(scaling_p.shape)
                if not decoded:
                    d_scal = d_scal * torch.exp(scaling_p)
            if shapes[2] is not None:
                d_mask = det_scatter_rows(vis, g_mask, A).view(mask_p.shape)
                if not decoded:
                    sg = torch.sigmoid(mask_p)
                    d_mask = d_mask * (sg * (1.0 - sg))
            return d_off, d_scal, d_mask, None, None, None, None
        d_off, d_scal, d_mask = _zeros_many(shapes, dev)
        _lib.check(_lib.lib().gsvc_gather_rows_backward(_lib.ptr(scaling_p), _lib.ptr(mask_p), _lib.ptr(vis), vis.shape[0], 0, K, S,
                                                        int(decoded), None, _lib.ptr(g_off), _lib.ptr(g_scal),
                                                        _lib.ptr(g_mask), None, _lib.ptr(d_off), _lib.ptr(d_scal),
                                                        _lib.ptr(d_mask), _lib.current_stream(dev)), "gsvc_gather_rows_backward")
        return d_off, d_scal, d_mask, None, None, None, None


def _gather_rows(pc, vis, ranks=None, parts="all"):
    """(feat, grid_offsets, grid_scaling, offset_masks) of the visible rows.  ``ranks`` = (seen, rank) of a StepPlan whose
    concatenated lists ``vis`` is: the backward then adds each anchor's rows in view order instead of with atomics.
    ``parts``: "all", "feat" (the features only) or "rows" (offsets, scaling, masks only) — a fitting step gathers the latter
    three late in its forward, so that their backward — the last contribution to the gradients of _offset / _scaling / _mask —
    runs early (autograd runs later-created nodes first)."""
    fused = (pc._anchor_feat.is_cuda and vis.dtype == torch.int64 and pc._mask.dim() == 3 and pc._mask.shape[2] == 1
             and pc._offset.dim() == 3 and pc._offset.shape[2] == 3 and not switches.NO_FUSED_GATHER)
    out = ()
    if fused:
        use = ranks is not None and ranks[0].numel() <= 8 * pc._anchor_feat.shape[0] and not switches.NO_RANKED_GATHER
        seen, rank = ranks if use else (None, None)      # the ranked kernel holds at most 8 views
        if parts != "rows":
            out += (_GatherFeat.apply(pc._anchor_feat, vis, seen, rank),)
        if parts != "feat":
            out += tuple(_GatherRows.apply(pc._offset, pc._scaling, pc._mask, vis, bool(pc.decoded_version), seen, rank))
        return out
    if parts != "rows":
        out += (pc._anchor_feat.index_select(0, vis),)
    if parts != "feat":
        out += (pc._offset.index_select(0, vis), _visible_scaling(pc, vis), _visible_mask(pc, vis))
    return out


def host_values(values, device, dtype=None):
    """A small tensor of host numbers on ``device`` WITHOUT blocking the host: ``torch.tensor(values, device=cuda)`` copies from
    pageable memory, i.e. the call returns only when the copy has run — behind everything already queued on the stream.  With
    the queue kept full across step boundaries (Trainer._early_tail) each such call stalled the host for the ~2 ms of queued
    work.  Pinned staging + an asynchronous copy returns at once (the pinned block is recycled by the caching host allocator
    once the copy has run)."""
    device = torch.device(device)
    if device.type != "cuda":
        return torch.tensor(values, dtype=dtype, device=device)
    return torch.tensor(values, dtype=dtype, pin_memory=True).to(device, non_blocking=True)


def _as_index(mask_or_index):
    """Boolean mask -> int64 index list (one nonzero); index tensors pass through.  Gathers by index have a
    scatter-add backward (atomics) instead of the sort-based index_put a boolean mask triggers."""
    if mask_or_index.dtype == torch.bool:
        return mask_or_index.nonzero(as_tuple=False).squeeze(1)
    return mask_or_index


class StepPlan:
    """Every data-dependent index list of one batched generation pass, computed WITHOUT a host synchronisation and read back
    with one copy: the visible anchors of each of the R views, the distinct anchors of their union (entropy context) and the
    5 % rate sample.  ``torch.nonzero`` / boolean-mask indexing read a count back per call — six device round trips at the
    start of a fitting step during which the GPU has nothing queued; a plan is built at the TAIL of the previous step (the
    kernels queue behind the optimizer, the nine counts travel to pinned host memory behind them), so the next step starts
    with one wait for that copy instead.  Index lists come from ``torch.nonzero_static`` (capacity = all anchors) and are
    cut to their true length once the counts are on the host (``resolve``)."""

    def __init__(self, frames, pc, visible_masks, geometry, sample: bool):
        dev = visible_masks[0].device
        A = visible_masks[0].shape[0]
        R = len(visible_masks)
        self.frames, self.visible_masks, self.geometry, self.R = frames, visible_masks, geometry, R
        self.key = (A, id(pc._anchor), pc._anchor._version, pc._scaling._version, pc._mask._version)
        # the R views side by side: ONE scan gives every view's ranks and count, ONE compaction of the flattened [R, A] mask
        # every view's index list (view r's list is the segment behind the r earlier views' counts, minus r * A)
        fused = (dev.type == "cuda" and R <= 16 and all(m.dtype == torch.bool and m.is_contiguous() for m in visible_masks)
                 and pc._mask.is_contiguous() and pc._mask.dtype == torch.float32 and not switches.NO_FUSED_PLAN)
        chosen = None
        if fused:
            # the masks of the plan in one launch (csrc/generate.hip k_plan_masks): views side by side, their union, the rate sample
            import ctypes
            from . import _lib
            M = torch.empty(R, A, dtype=torch.bool, device=dev)
            present = torch.empty(A, dtype=torch.bool, device=dev)
            u = torch.rand(R, A, device=dev) if sample else None
            chosen = torch.empty(R, A, dtype=torch.bool, device=dev) if sample else None
            vp = (ctypes.c_void_p * R)(*[m.data_ptr() for m in visible_masks])
            _lib.check(_lib.lib().gsvc_plan_masks(vp, R, A, _lib.ptr(pc._mask), pc._mask.numel() // max(A, 1), int(bool(pc.decoded_version)),
                                                  _lib.ptr(u), float(SAMPLE_RATE), _lib.ptr(M), _lib.ptr(present), _lib.ptr(chosen),
                                                  _lib.current_stream(dev)), "gsvc_plan_masks")
        else:
            M = torch.stack(visible_masks)                               # [R, A] bool
            present = M.any(dim=0)
        from . import dist as gdist
        self._dp = gdist.active()
        self._gmax = self._gmax_work = None
        self._A, self._sel_flat, self._ends = A, None, fused
        if fused:
            # scans, counts and index lists in three launches (csrc/generate.hip gsvc_plan_scans): c - 1 is the row of (r, a) in the
            # concatenated rows, view r's index list is the segment of the flat list behind the r earlier views' counts (minus r A),
            # pos maps an anchor to its row of the distinct list, the sample is listed by row
            L, st = _lib.lib(), _lib.current_stream(dev)
            big = torch.empty(3 * R * A + 2 * A + R + 2, dtype=torch.int64, device=dev)
            c, self._flat, sel_flat = big[:R * A], big[R * A:2 * R * A], big[2 * R * A:3 * R * A]
            self.pos, self._distinct = big[3 * R * A:3 * R * A + A], big[3 * R * A + A:3 * R * A + 2 * A]
            counts = big[3 * R * A + 2 * A:]                         # [R view ends | chosen pairs | distinct anchors]
            scratch = torch.empty(int(L.gsvc_plan_scans_scratch_bytes(R, A)), dtype=torch.uint8, device=dev)
            _lib.check(L.gsvc_plan_scans(_lib.ptr(M), _lib.ptr(chosen), _lib.ptr(present), R, A, _lib.ptr(scratch), _lib.ptr(c), _lib.ptr(self._flat),
                                         _lib.ptr(sel_flat) if sample else None, ctypes.c_void_p(self.pos.data_ptr()),
                                         ctypes.c_void_p(self._distinct.data_ptr()), ctypes.c_void_p(counts.data_ptr()), st), "gsvc_plan_scans")
            self.ranks = (M.view(-1), c)                                 # for the atomic-free gather backward (_gather_rows)
            if sample:
                self._sel_flat = sel_flat
            n_distinct = counts[R + 1:R + 2]
        else:
            # scan of the FLATTENED mask (a 1-D scan is one fast pass; a [R, A] scan along dim 1 runs row by row): c - 1 is the
            # row of (r, a) in the concatenated rows, the view boundaries give the counts
            c = torch.cumsum(M.view(-1), dim=0)
            self.ranks = (M.view(-1), c)
            ends = c[A - 1::A]
            pos_incl = torch.cumsum(present, dim=0)
            self.pos = pos_incl - 1                                      # anchor -> row of the distinct list
            n_distinct = pos_incl[-1:]
            parts = [torch.diff(ends, prepend=ends.new_zeros(1))]
            if sample:
                # the rate sample in anchor space: a row is (render r, visible anchor a); its position in the concatenated rows is
                # (visible anchors of the renders before r) + (rank of a among render r's visible anchors) = c - 1
                with torch.no_grad():
                    live = pc.get_mask.reshape(A, -1).sum(dim=1) > 0      # mask_anchor for every anchor
                chosen = M & live.unsqueeze(0) & (torch.rand(R, A, device=dev) <= SAMPLE_RATE)
                parts.append(chosen.sum().reshape(1))
            else:
                parts.append(c.new_zeros(1))
            parts.append(n_distinct)
            counts = torch.cat(parts)
        # data parallel: the largest distinct-anchor count over the ranks (the capacity of the sparse gradient exchange:
        # gsvc_amd.dist.GradReducer.set_sparse) rides along; its collective runs on the plan's own process group
        if self._dp:
            # asynchronous, and waited for only when the next step resolves the plan: a rank that builds its plan from inside the
            # backward (early tail) must not stop there until a rank that builds it after the backward arrives — that rank's
            # backward waits for collectives the first one has yet to launch
            from . import dist as gdist
            self._gmax = n_distinct.clone()
            self._gmax_work = torch.distributed.all_reduce(self._gmax, op=torch.distributed.ReduceOp.MAX, group=gdist.plan_group(),
                                                           async_op=True)
        # the counts leave for the host: the next step waits for this copy only
        self._sample = sample
        self._host = torch.empty(R + 2, dtype=torch.int64, pin_memory=True)
        self._host.copy_(counts, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()
        if not fused:
            self._flat = torch.nonzero_static(M.view(-1), size=R * A).squeeze(1)
            self._distinct = torch.nonzero_static(present, size=A).squeeze(1)
            if sample:
                pick = torch.nonzero_static(chosen.view(-1), size=R * A).squeeze(1)       # in (r, a) order = row order
                self._sel_flat = c.index_select(0, pick.clamp_min(0)) - 1
        self.vis_list = self.distinct = self.sel = self.vis_all = None
        self.distinct_cap = None

    def matches(self, pc) -> bool:
        return self.key == (self.visible_masks[0].shape[0], id(pc._anchor), pc._anchor._version, pc._scaling._version,
                            pc._mask._version)

    def resolve(self):
        """The one wait: cut the padded lists to their lengths."""
        if self.vis_list is None:
            _blocked_wait(self._event)
            n = self._host.tolist()
            R, A = self.R, self._A
            if self._ends:                   # the fused scans report the scan at each view's end: counts are the differences
                n = [n[r] - (n[r - 1] if r else 0) for r in range(R)] + n[R:]
            self.vis_list, at = [], 0
            for r in range(R):
                # the fused scans list anchors, torch.nonzero_static positions of the flattened [R, A] mask
                self.vis_list.append(self._flat[at:at + n[r]] if self._ends else self._flat[at:at + n[r]] - r * A)
                at += n[r]
            self.vis_all = self._flat[:at] if self._ends else None      # the views' lists concatenated (they are one list already)
            self.distinct = self._distinct[:n[R + 1]]
            if self._gmax_work is not None:
                self._gmax_work.wait()
                self.distinct_cap = int(self._gmax.item())      # queued a whole step ago: complete by now
            if self._sel_flat is not None:
                self.sel = self._sel_flat[:n[R]]
        return self


HOST_TIMES = {}        # region -> [calls, seconds] of HOST time, filled when GSVC_HOST_TIMES=1 (tools/ab/host_regions.py)
_REGION_MODE = 2 if os.environ.get("GSVC_HOST_TIMES") else (1 if os.environ.get("GSVC_REGIONS") else 0)


class _HostTimer:
    __slots__ = ("name", "t0")

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        e = HOST_TIMES.setdefault(self.name, [0, 0.0])
        e[0] += 1
        e[1] += time.perf_counter() - self.t0


_NULL = None


def region(name):
    """Named range: torch.profiler ranges when GSVC_REGIONS=1, host wall-clock per region when GSVC_HOST_TIMES=1 (both read once,
    at import: diagnostics), otherwise a shared no-op context."""
    global _NULL
    if _REGION_MODE == 0:
        if _NULL is None:
            import contextlib
            _NULL = contextlib.nullcontext()
        return _NULL
    if _REGION_MODE == 2:
        return _HostTimer(name)
    return torch.profiler.record_function(name)


class _GenTail(torch.autograd.Function):
    """Tail of the generation for un-compacted renders (csrc/generate.hip): returns
    (neural_opacity[n,1], mask[n], scaling[n,3], rot[n,4], world[n,3], xyz[n,3])."""

    @staticmethod
    def forward(ctx, op_raw, offset_mask, grid_offsets, neural_offset, scale_rot, grid_scaling, anchor, K, bmin, bmax):
        import ctypes as C
        from . import _lib
        dev = op_raw.device
        op_raw, offset_mask = op_raw.contiguous().view(-1), offset_mask.contiguous().view(-1)
        grid_offsets, neural_offset = grid_offsets.contiguous().view(-1, 3), neural_offset.contiguous().view(-1, 3)
        scale_rot, grid_scaling, anchor = scale_rot.contiguous(), grid_scaling.contiguous(), anchor.contiguous()
        rows = grid_scaling.shape[0]
        n = rows * K
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)  # noqa: E731
        no, scaling, rot, world, xyz = f(n, 1), f(n, 3), f(n, 4), f(n, 3), f(n, 3)
        mask = torch.empty(n, dtype=torch.bool, device=dev)
        lo, hi = (C.c_float * 3)(*bmin), (C.c_float * 3)(*bmax)
        _lib.check(_lib.lib().gsvc_gen_tail_forward(
            _lib.ptr(op_raw), _lib.ptr(offset_mask), _lib.ptr(grid_offsets), _lib.ptr(neural_offset), _lib.ptr(scale_rot),
            _lib.ptr(grid_scaling), _lib.ptr(anchor), lo, hi, rows, K, _lib.ptr(no), _lib.ptr(mask), _lib.ptr(scaling), _lib.ptr(rot),
            _lib.ptr(world), _lib.ptr(xyz), _lib.current_stream(dev)), "gsvc_gen_tail_forward")
        ctx.save_for_backward(op_raw, offset_mask, grid_offsets, neural_offset, scale_rot, grid_scaling, world)
        ctx.K, ctx.bounds, ctx.anchor_grad = K, (lo, hi), anchor.requires_grad
        ctx.mark_non_differentiable(mask)
        ctx.set_materialize_grads(False)      # an output nothing differentiates (world without the optical loss) arrives as None, not zeros
        return no, mask, scaling, rot, world, xyz

    @staticmethod
    def backward(ctx, g_no, _g_mask, g_scaling, g_rot, g_world, g_xyz):
        from . import _lib
        op_raw, offset_mask, grid_offsets, neural_offset, scale_rot, grid_scaling, world = ctx.saved_tensors
        dev = op_raw.device
        rows, K = grid_scaling.shape[0], ctx.K
        n = rows * K
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)  # noqa: E731
        d_op, d_om, d_off, d_sr, d_gs = f(n), f(n), f(n, 3), f(n, 7), f(rows, 6)
        d_anchor = f(rows, 3) if ctx.anchor_grad else None
        c = lambda t: t.contiguous() if t is not None else None  # noqa: E731
        g_no, g_scaling, g_rot, g_world, g_xyz = c(g_no), c(g_scaling), c(g_rot), c(g_world), c(g_xyz)
        _lib.check(_lib.lib().gsvc_gen_tail_backward(
            _lib.ptr(op_raw), _lib.ptr(offset_mask), _lib.ptr(grid_offsets), _lib.ptr(neural_offset), _lib.ptr(scale_rot),
            _lib.ptr(grid_scaling), _lib.ptr(world), ctx.bounds[0], ctx.bounds[1], rows, K, _lib.ptr(g_no), _lib.ptr(g_scaling),
            _lib.ptr(g_rot), _lib.ptr(g_world), _lib.ptr(g_xyz), _lib.ptr(d_op), _lib.ptr(d_om), _lib.ptr(d_off), _lib.ptr(d_sr),
            _lib.ptr(d_gs), _lib.ptr(d_anchor), _lib.current_stream(dev)), "gsvc_gen_tail_backward")
        return d_op, d_om, d_off, d_off, d_sr, d_gs, d_anchor, None, None, None


def _sync(t):
    if t.is_cuda:
        torch.cuda.synchronize()


def generate_neural_gaussians(frame, pc, visible_mask=None, mode=GenerateMode.TRAINING_FULL_PRECISION):
    time_sub = 0
    all_anchor = pc.get_anchor
    if visible_mask is None:
        visible_mask = torch.ones(all_anchor.shape[0], dtype=torch.bool, device=all_anchor.device)
    K = pc.n_offsets
    vis = _as_index(visible_mask)            # one nonzero for the five per-anchor gathers
    anchor = all_anchor.index_select(0, vis)
    feat = pc._anchor_feat.index_select(0, vis)
    grid_offsets = pc._offset.index_select(0, vis)
    grid_scaling = _visible_scaling(pc, vis)
    offset_masks = _visible_mask(pc, vis)
    rate = RatePack()
    Q_feat, Q_scaling, Q_offsets = BASE_Q_FEAT, BASE_Q_SCALING, BASE_Q_OFFSETS

    if mode in (GenerateMode.TRAINING_FULL_PRECISION, GenerateMode.DECODING_AS_IS):
        pass
    elif mode == GenerateMode.TRAINING_QUANTIZED:
        feat = pc.noise_quantizer(feat, Q_feat)
        grid_scaling = pc.noise_quantizer(grid_scaling, Q_scaling)
        grid_offsets = pc.noise_quantizer(grid_offsets, Q_offsets)
    elif mode == GenerateMode.TRAINING_ENTROPY:
        ec = pc.calc_entropy_context(anchor)
        Q_feat, Q_scaling, Q_offsets = Q_feat * ec.Q_feat_adj, Q_scaling * ec.Q_scaling_adj, Q_offsets * ec.Q_offsets_adj
        feat = pc.noise_quantizer(feat, Q_feat)
        grid_scaling = pc.noise_quantizer(grid_scaling, Q_scaling)
        grid_offsets = pc.noise_quantizer(grid_offsets, Q_offsets.unsqueeze(1))
        rate = calc_sampled_rate(pc, vis, feat, grid_scaling, grid_offsets, Q_feat, Q_scaling, Q_offsets, ec)
    elif mode == GenerateMode.TRAININ_STE_ENTROPY:
        _sync(anchor)
        t1 = time.time()
        ec = pc.calc_entropy_context(anchor)
        Q_feat = Q_feat * ec.Q_feat_adj.detach()
        Q_scaling = Q_scaling * ec.Q_scaling_adj.detach()
        Q_offsets = Q_offsets * ec.Q_offsets_adj.detach()
        feat = STE_multistep.apply(feat, Q_feat, pc._anchor_feat.mean()).detach()
        grid_scaling = STE_multistep.apply(grid_scaling, Q_scaling, pc.get_scaling.mean()).detach()
        grid_offsets = STE_multistep.apply(grid_offsets, Q_offsets.unsqueeze(1), pc._offset.mean()).detach()
        _sync(anchor)
        time_sub = time.time() - t1
        rate = calc_sampled_rate(pc, vis, feat, grid_scaling, grid_offsets, Q_feat, Q_scaling, Q_offsets, ec)
    else:
        raise ValueError(f"Unknown mode {mode}")

    # conditioning: embedding of the frame's z (same for every anchor) and of the anchor's z offset to it
    cam = frame.cam_pos.to(anchor.device)
    ob_view = (anchor - cam)[:, 2:]
    time_emb = pc.embed_time_fn(torch.zeros_like(ob_view) + cam[-1])
    z_emb = pc.embed_fn(ob_view)
    pe = torch.cat([time_emb, z_emb], dim=1)

    V = anchor.shape[0]
    neural_opacity = pc.get_opacity_mlp(feat, pe).reshape(-1, 1) * offset_masks.view(-1, 1)
    mask = (neural_opacity > 0.0).view(-1)
    color = pc.get_color_mlp(feat, pe).reshape(V * K, 3)
    scale_rot = pc.get_cov_mlp(feat, pe).reshape(V * K, 7)
    neural_offset = pc.get_deform_mlp(torch.cat([feat, pe], dim=1)).reshape(V * K, 3)
    offsets = grid_offsets.view(-1, 3) + neural_offset

    # per-Gaussian table [scaling(6) | anchor(3) | colour(3) | scale_rot(7) | offset(3)], kept because the
    # optical-flow loss of the training step reads it (reference utils/loss_utils.py:108-118)
    per_anchor = torch.cat([grid_scaling, anchor], dim=-1)
    concatenated_all = torch.cat([per_anchor.repeat_interleave(K, dim=0), color, scale_rot, offsets], dim=-1)
    alive_idx = mask.nonzero(as_tuple=False).squeeze(1)
    alive = concatenated_all.index_select(0, alive_idx)
    scaling_rep, anchor_rep = alive[:, 0:6], alive[:, 6:9]
    color_a, scale_rot_a, offsets_a = alive[:, 9:12], alive[:, 12:19], alive[:, 19:22]

    scaling = scaling_rep[:, 3:] * torch.sigmoid(scale_rot_a[:, :3])
    rot = pc.rotation_activation(scale_rot_a[:, 3:7])
    xyz = torch.clamp(anchor_rep + offsets_a * scaling_rep[:, :3], pc.x_bound_min, pc.x_bound_max)
    return GeneratedGaussians(
        xyz=xyz, color=color_a, opacity=neural_opacity.index_select(0, alive_idx),
        scaling=scaling, rot=rot,
        neural_opacity=neural_opacity, visable_mask=visible_mask, mask=mask,
        bit_per_param=rate.bit_per_param, bit_per_feat_param=rate.bit_per_feat_param,
        bit_per_scaling_param=rate.bit_per_scaling_param, bit_per_offsets_param=rate.bit_per_offsets_param,
        concatenated_all=concatenated_all, time_sub=time_sub)


# --------------------------------------------------------------------------------------------------------------
# Batched generation: the fitting step renders 4 views per step (2 adjacent frames x 2 opposite views).  Their
# anchor -> Gaussian generation is the same arithmetic on different rows, so the rows of all R renders are
# concatenated and every MLP / grid / rate / elementwise op runs once on the batch (4x fewer launches, 4x larger
# GEMMs) instead of once per render.  Per-render statistics the reference takes over one render's visible set
# (clamp bounds from the visible mean, sampled-rate means, keep rate) are taken per segment, so each render's
# values are what a stand-alone `generate_neural_gaussians` call computes; only the order of the random draws
# differs (one draw over the batch instead of R draws).
class _Segments:
    def __init__(self, counts, device):
        self.counts = [int(c) for c in counts]
        self.R = len(self.counts)
        self.bounds = [0]
        for c in self.counts:
            self.bounds.append(self.bounds[-1] + c)
        self.rows = self.bounds[-1]
        self._device = device
        self._both = self._seg_id = None

    def _tensors(self):
        if self._both is None:
            self._both = host_values(list(self.counts) + list(self.bounds), self._device)      # one staged copy for both
        return self._both

    # device copies of the counts / bounds and the row -> render map: built on first use (the fused paths take the bounds by value
    # and never ask; eagerly they cost a staged copy, an arange and a repeat_interleave per batch)
    @property
    def counts_t(self):
        return self._tensors()[:len(self.counts)]

    @property
    def bounds_t(self):
        return self._tensors()[len(self.counts):]

    @property
    def seg_id(self):
        if self._seg_id is None:
            # output_size: without it repeat_interleave reads the total back from the device (a host synchronisation)
            self._seg_id = torch.repeat_interleave(torch.arange(self.R, device=self._device), self.counts_t, output_size=self.rows)
        return self._seg_id

    def sums(self, x):
        """Per-segment sums of a [rows] tensor as float32 through one scan (an index_add_ onto R addresses serialises on
        its atomics: 28 us for 200 k rows)."""
        c = torch.cumsum(x.to(torch.float32) if x.dtype != torch.bool else x, dim=0)
        c = torch.cat([c.new_zeros(1), c])
        e = c.index_select(0, self.bounds_t)
        return (e[1:] - e[:-1]).to(torch.float32)

    def mean(self, x):
        """Per-segment mean of x over all trailing dims, returned per row ([rows] + [1]*(x.dim()-1))."""
        flat = x.reshape(x.shape[0], -1)
        sums = torch.zeros(self.R, device=x.device, dtype=x.dtype).index_add_(0, self.seg_id, flat.sum(dim=1))
        means = sums / (self.counts_t.to(x.dtype) * flat.shape[1]).clamp_min(1)
        return means.index_select(0, self.seg_id).view([-1] + [1] * (x.dim() - 1))

    def slices(self, per_row=1):
        return [slice(self.bounds[r] * per_row, self.bounds[r + 1] * per_row) for r in range(self.R)]


def _seg_bounds(x, Q, seg, x_mean=None):
    """centre -+ 15000 steps per row, centre = mean(x)/mean(Q) of the row's render (encodings.py:398-409,434-447)."""
    xm = seg.mean(x) if x_mean is None else x_mean
    qm = seg.mean(Q).view([-1] + [1] * (x.dim() - 1)) if isinstance(Q, torch.Tensor) else Q
    centre = (xm / qm).detach()
    return centre - 15_000, centre + 15_000


class _NoiseQuant(torch.autograd.Function):
    """clamp(x / Q, centre_r -+ 15000) * Q + U(-1/2, 1/2) * Q per render r, fused (csrc/quant.hip)."""

    @staticmethod
    def forward(ctx, x2, q_rows, q_scalar, noise, offsets):
        import ctypes as C
        from . import _lib
        dev = x2.device
        x2, noise = x2.contiguous(), noise.contiguous()
        q = q_rows.contiguous() if q_rows is not None else None
        R = len(offsets) - 1
        seg = (C.c_int64 * (R + 1))(*[int(v) for v in offsets])
        L = _lib.lib()
        scratch = torch.empty(max(int(L.gsvc_noise_quant_scratch_floats(seg, R)), 1), dtype=torch.float32, device=dev)
        centre = torch.empty(R, dtype=torch.float32, device=dev)
        y = torch.empty_like(x2)
        _lib.check(L.gsvc_noise_quant_forward(_lib.ptr(x2), _lib.ptr(q), float(q_scalar), _lib.ptr(noise), seg, R, x2.shape[1],
                                              _lib.ptr(scratch), _lib.ptr(centre), _lib.ptr(y), _lib.current_stream(dev)),
                   "gsvc_noise_quant_forward")
        ctx.save_for_backward(x2, q, noise, centre)
        ctx.seg, ctx.R, ctx.q_scalar = seg, R, float(q_scalar)
        return y

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        x2, q, noise, centre = ctx.saved_tensors
        dev = x2.device
        dx = torch.empty_like(x2)
        dq = torch.empty(x2.shape[0], dtype=torch.float32, device=dev) if q is not None else None
        _lib.check(_lib.lib().gsvc_noise_quant_backward(_lib.ptr(g.contiguous()), _lib.ptr(x2), _lib.ptr(q), ctx.q_scalar,
                                                        _lib.ptr(noise), _lib.ptr(centre), ctx.seg, ctx.R, x2.shape[1],
                                                        _lib.ptr(dx), _lib.ptr(dq), _lib.current_stream(dev)),
                   "gsvc_noise_quant_backward")
        return dx, dq, None, None, None


def _seg_noise_quant(x, Q, seg):
    if x.is_cuda and x.shape[0] > 0 and x[0].numel() <= 256:
        # fused path: x viewed as [rows, C]; Q is a python number or one step per row
        x2 = x.reshape(x.shape[0], -1)
        noise = torch.empty_like(x2).uniform_(-0.5, 0.5)
        if isinstance(Q, torch.Tensor):
            y = _NoiseQuant.apply(x2, Q.reshape(-1), 0.0, noise, seg.bounds)
        else:
            y = _NoiseQuant.apply(x2, None, float(Q), noise, seg.bounds)
        return y.view(x.shape)
    lo, hi = _seg_bounds(x, Q, seg)
    x = torch.clamp(x / Q, min=lo, max=hi) * Q
    return x + torch.empty_like(x).uniform_(-0.5, 0.5) * Q


def _seg_ste(x, Q, seg, x_mean):
    if (x.is_cuda and x.dtype == torch.float32 and x.shape[0] > 0 and x[0].numel() <= 256 and 1 <= seg.R <= 8
            and isinstance(x_mean, torch.Tensor) and x_mean.is_cuda and x_mean.numel() == 1 and x_mean.dtype == torch.float32
            and (not isinstance(Q, torch.Tensor) or (Q.dtype == torch.float32 and Q.numel() == x.shape[0])) and not switches.NO_FUSED_STE):
        # three launches (the step's per-render mean, the truncated bounds, the quantiser) instead of a dozen tensor operations;
        # the same expression operation by operation (csrc/quant.hip k_ste_fwd).  The result carries no graph: the reference
        # detaches it (guassian.py:205-207)
        import ctypes as C
        from . import _lib
        with torch.no_grad():
            x2 = x.detach().reshape(x.shape[0], -1).contiguous()
            q = Q.detach().reshape(-1).contiguous() if isinstance(Q, torch.Tensor) else None
            R = seg.R
            offs = (C.c_int64 * (R + 1))(*[int(v) for v in seg.bounds])
            L = _lib.lib()
            scratch = torch.empty(max(int(L.gsvc_noise_quant_scratch_floats(offs, R)), 1) + 2 * R, dtype=torch.float32, device=x.device)
            y = torch.empty_like(x2)
            _lib.check(L.gsvc_ste_quant_forward(_lib.ptr(x2), _lib.ptr(q), 0.0 if q is not None else float(Q), _lib.ptr(x_mean.detach()), offs, R,
                                                x2.shape[1], _lib.ptr(scratch), C.c_void_p(scratch.data_ptr() + 4 * (scratch.numel() - 2 * R)),
                                                _lib.ptr(y), _lib.current_stream(x.device)), "gsvc_ste_quant_forward")
        return y.view(x.shape)
    lo, hi = _seg_bounds(x, Q, seg, x_mean)
    x = torch.clamp(x / Q, min=torch.trunc(lo), max=torch.trunc(hi)) * Q
    return (x + (torch.round(x / Q) * Q - x).detach()).detach()


def _zeros_many(shapes, device):
    """Zero tensors of the given shapes (None -> None) carved from ONE zero-filled buffer: one fill launch instead of one per tensor
    (a backward that scatters into a dozen dense gradients otherwise starts with a dozen memsets)."""
    sizes = [0 if sh is None else int(torch.Size(sh).numel()) for sh in shapes]
    padded = [(n + 3) // 4 * 4 for n in sizes]           # keep every tensor 16-byte aligned
    flat = torch.zeros(max(sum(padded), 1), device=device, dtype=torch.float32)
    out, at = [], 0
    for sh, n, p in zip(shapes, sizes, padded):
        out.append(None if sh is None else flat[at:at + n].view(sh))
        at += p
    return out


class _SampledRate(torch.autograd.Function):
    """S[R, 3]: bits of the sampled rows' features / scalings / offsets summed per render (csrc/rate.hip k_rate_sample): gathers
    by row index, per-render clamp bounds, the offset mask and the per-render sums in one launch; the backward writes straight
    into dense gradients of the tensors the rows were gathered from."""

    @staticmethod
    def forward(ctx, feat, grid_scaling, grid_offsets, offset_masks, Q_feat, Q_scaling, Q_offsets, mean_f, scale_f, mean_s, scale_s,
                mean_o, scale_o, sel, sel_ctx, x_mean, bounds, K):
        import ctypes as C
        from . import _lib
        dev = feat.device
        xs = [feat.contiguous(), grid_scaling.contiguous(), grid_offsets.contiguous().view(grid_offsets.shape[0], -1)]
        means = [mean_f.contiguous(), mean_s.contiguous(), mean_o.contiguous()]
        scales = [scale_f.contiguous(), scale_s.contiguous(), scale_o.contiguous()]
        Qs = [Q_feat.contiguous().view(-1), Q_scaling.contiguous().view(-1), Q_offsets.contiguous().view(-1)]
        mask = offset_masks.contiguous().view(offset_masks.shape[0], -1)
        sel = sel.contiguous()
        sel_ctx = sel_ctx.contiguous() if sel_ctx is not None else None
        R = len(bounds) - 1
        d = _lib.RateSampleC()
        for g in range(3):
            d.x[g], d.mean[g], d.scale[g], d.Q[g] = xs[g].data_ptr(), means[g].data_ptr(), scales[g].data_ptr(), Qs[g].data_ptr()
            d.C[g] = int(xs[g].shape[1])
        x_mean = x_mean.contiguous()
        d.x_mean = x_mean.data_ptr()
        d.mask, d.sel = mask.data_ptr(), sel.data_ptr() if sel.numel() else None
        d.sel_ctx = sel_ctx.data_ptr() if (sel_ctx is not None and sel_ctx.numel()) else None
        rb = (C.c_int64 * (R + 1))(*[int(b) for b in bounds])
        d.row_bounds, d.K, d.renders, d.n_sel = rb, int(K), R, int(sel.shape[0])
        L = _lib.lib()
        scratch = torch.empty(int(L.gsvc_rate_sample_scratch_floats(d.n_sel)), device=dev, dtype=torch.float32)
        S = torch.empty(R, 3, device=dev, dtype=torch.float32)
        _lib.check(L.gsvc_rate_sample_forward(C.byref(d), _lib.ptr(scratch), _lib.ptr(S), _lib.current_stream(dev)),
                   "gsvc_rate_sample_forward")
        ctx.save_for_backward(*xs, *means, *scales, *Qs, mask, sel, sel_ctx if sel_ctx is not None else sel, scratch, x_mean)
        ctx.desc, ctx.rb, ctx.has_ctx = d, rb, sel_ctx is not None
        ctx.shapes = (grid_offsets.shape, offset_masks.shape, Q_feat.shape, Q_scaling.shape, Q_offsets.shape)
        return S

    @staticmethod
    def backward(ctx, gS):
        import ctypes as C
        from . import _lib
        t = ctx.saved_tensors
        xs, means, scales, Qs, mask, scratch = t[0:3], t[3:6], t[6:9], t[9:12], t[12], t[15]
        dev = mask.device
        need = ctx.needs_input_grad
        sh = lambda t, ok: t.shape if ok else None  # noqa: E731
        zs = _zeros_many([sh(xs[g], need[g]) for g in range(3)] + [sh(mask, need[3])] + [sh(Qs[g], need[4 + g]) for g in range(3)] +
                         [sh(means[g], need[7 + 2 * g]) for g in range(3)] + [sh(scales[g], need[8 + 2 * g]) for g in range(3)], dev)
        dx, dmask, dQ, dmean, dscale = zs[0:3], zs[3], zs[4:7], zs[7:10], zs[10:13]
        arr = lambda ts: (C.c_void_p * 3)(*[x.data_ptr() if x is not None else None for x in ts])  # noqa: E731
        _lib.check(_lib.lib().gsvc_rate_sample_backward(C.byref(ctx.desc), _lib.ptr(scratch), _lib.ptr(gS.contiguous()), arr(dx),
                                                        arr(dmean), arr(dscale), arr(dQ), _lib.ptr(dmask),
                                                        _lib.current_stream(dev)), "gsvc_rate_sample_backward")
        go_shape, m_shape, qf, qs, qo = ctx.shapes
        v = lambda x, sh: x.view(sh) if x is not None else None  # noqa: E731
        return (dx[0], dx[1], v(dx[2], go_shape), v(dmask, m_shape), v(dQ[0], qf), v(dQ[1], qs), v(dQ[2], qo),
                dmean[0], dscale[0], dmean[1], dscale[1], dmean[2], dscale[2], None, None, None, None, None)


class _QRows(torch.autograd.Function):
    """(Q_feat, Q_scaling, Q_offsets) per row = base step x the entropy context's adjustment of the row's anchor (csrc/generate.hip
    k_q_rows_fwd / _bwd), three [rows, 1] tensors carved from one buffer.  ``raw``: the three inputs are the quant_step networks'
    raw outputs and the adjustment exp(clamp(q, -10, 10)) is applied inside (forward and backward)."""

    @staticmethod
    def forward(ctx, adj_f, adj_s, adj_o, ctx_row, q_f, q_s, q_o, raw=False):
        from . import _lib
        adj = [a.contiguous().view(-1) for a in (adj_f, adj_s, adj_o)]
        D, dev = adj[0].shape[0], adj[0].device
        ctx_row = ctx_row.contiguous() if ctx_row is not None else None
        rows = ctx_row.shape[0] if ctx_row is not None else D
        out = torch.empty(3, rows, 1, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().gsvc_q_rows_forward(_lib.ptr(adj[0]), _lib.ptr(adj[1]), _lib.ptr(adj[2]), _lib.ptr(ctx_row), float(q_f), float(q_s),
                                                  float(q_o), rows, int(raw), _lib.ptr(out), _lib.current_stream(dev)), "gsvc_q_rows_forward")
        ctx.save_for_backward(ctx_row, *(adj if raw else ()))
        ctx.q, ctx.rows, ctx.D, ctx.shapes, ctx.raw = (float(q_f), float(q_s), float(q_o)), rows, D, (adj_f.shape, adj_s.shape, adj_o.shape), bool(raw)
        ctx.set_materialize_grads(False)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_f, g_s, g_o):
        from . import _lib
        ctx_row, *adj = ctx.saved_tensors
        gs = [None if g is None else g.contiguous() for g in (g_f, g_s, g_o)]
        dev = next(g.device for g in gs if g is not None) if any(g is not None for g in gs) else None
        if dev is None:
            return (None,) * 8
        adj = adj if ctx.raw else [None, None, None]
        if switches.DETERMINISTIC and ctx_row is not None:
            # the rows of a distinct anchor (one per view that sees it) summed in row order, then the adjustment's derivative once
            z = torch.zeros(ctx.rows, dtype=torch.float32, device=dev)
            gsum = det_scatter_rows(ctx_row, torch.stack([z if g is None else g.reshape(-1) for g in gs], dim=1), ctx.D)      # [D, 3]
            outs = []
            for k in range(3):
                f = ctx.q[k]
                if ctx.raw:
                    v = adj[k]
                    f = torch.where((v >= -10.0) & (v <= 10.0), ctx.q[k] * torch.exp(v), torch.zeros_like(v))
                outs.append((gsum[:, k] * f).view(ctx.shapes[k]))
            return outs[0], outs[1], outs[2], None, None, None, None, None
        gadj = torch.empty(3, ctx.D, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().gsvc_q_rows_backward(_lib.ptr(gs[0]), _lib.ptr(gs[1]), _lib.ptr(gs[2]), _lib.ptr(ctx_row), *ctx.q, ctx.rows, ctx.D,
                                                   _lib.ptr(adj[0]), _lib.ptr(adj[1]), _lib.ptr(adj[2]), int(ctx.raw), _lib.ptr(gadj),
                                                   _lib.current_stream(dev)), "gsvc_q_rows_backward")
        return gadj[0].view(ctx.shapes[0]), gadj[1].view(ctx.shapes[1]), gadj[2].view(ctx.shapes[2]), None, None, None, None, None


class _RateNorm(torch.autograd.Function):
    """(out [R, 4], sum of out[:, 0]): the per-render bit sums S [R, 3] divided by the coded parameters of each group and of all
    three, times the render's keep rate (csrc/rate.hip k_rate_normalise; reference guassian.py:110-132)."""

    @staticmethod
    def forward(ctx, S, offset_masks, sel, bounds, dims, K):
        import ctypes as C
        from . import _lib
        dev = S.device
        R = len(bounds) - 1
        S, sel = S.contiguous(), sel.contiguous()
        om = offset_masks.contiguous()
        L = _lib.lib()
        scratch = torch.empty(int(L.gsvc_rate_normalise_scratch_bytes()) // 4, dtype=torch.int32, device=dev)
        res = torch.empty(8 * R + 1, dtype=torch.float32, device=dev)
        out, coef, total = res[:4 * R].view(R, 4), res[4 * R:8 * R], res[8 * R:]
        _lib.check(L.gsvc_rate_normalise_forward(_lib.ptr(S), _lib.ptr(om), int(K), _lib.ptr(sel) if sel.numel() else None, sel.numel(),
                                                 (C.c_int64 * (R + 1))(*[int(b) for b in bounds]), R, (C.c_float * 3)(*dims),
                                                 _lib.ptr(scratch), _lib.ptr(out), C.c_void_p(coef.data_ptr()), C.c_void_p(total.data_ptr()),
                                                 _lib.current_stream(dev)), "gsvc_rate_normalise_forward")
        ctx.save_for_backward(coef)
        ctx.R = R
        ctx.set_materialize_grads(False)
        return out, total.view(())

    @staticmethod
    def backward(ctx, g_out, g_sum):
        from . import _lib
        (coef,) = ctx.saved_tensors
        if g_out is None and g_sum is None:
            return None, None, None, None, None, None
        gS = torch.empty(ctx.R, 3, dtype=torch.float32, device=coef.device)
        g_out = None if g_out is None else g_out.contiguous()
        g_sum = None if g_sum is None else g_sum.contiguous()
        _lib.check(_lib.lib().gsvc_rate_normalise_backward(_lib.ptr(coef), _lib.ptr(g_out), _lib.ptr(g_sum), ctx.R, _lib.ptr(gS),
                                                           _lib.current_stream(coef.device)), "gsvc_rate_normalise_backward")
        return gS, None, None, None, None, None


_IOTA = {}


def _iota(dev, n):
    """arange(n) int64 on ``dev`` as a view of a cached, growing tensor (no launch per step), one per stream: the tensor is created
    by, and when it grows freed to the pool of, the stream that asked — handing one stream's to another would need record_stream
    bookkeeping for nothing."""
    key = (dev.type, dev.index, torch.cuda.current_stream(dev).stream_id if dev.type == "cuda" else 0)
    t = _IOTA.get(key)
    if t is None or t.shape[0] < n:
        t = _IOTA[key] = torch.arange(max(n, 1 << 16), dtype=torch.int64, device=dev)
    return t[:n]


def _param_means(pc):
    """(mean(_anchor_feat), mean(get_scaling), mean(_offset)) over ALL anchors as one float32 [3] tensor (csrc/rate.hip
    k_param_means: one pass; the torch expression is three reductions + an exp pass)."""
    zo = getattr(pc, "_zown", None)
    if zo is not None and zo.means is not None:      # z-range ownership: the owners' means (a replica's own rows are partly stale)
        return zo.means

    from . import _lib
    f, sc, o = pc._anchor_feat, getattr(pc, "_scaling", None), pc._offset
    if sc is None or not (f.is_cuda and f.dtype == sc.dtype == o.dtype == torch.float32 and f.is_contiguous() and sc.is_contiguous() and o.is_contiguous()):
        with torch.no_grad():
            return torch.stack([f.mean(), pc.get_scaling.mean(), o.mean()]).float()
    L = _lib.lib()
    scratch = torch.empty(int(L.gsvc_param_means_scratch_floats()), device=f.device, dtype=torch.float32)
    out = torch.empty(3, device=f.device, dtype=torch.float32)
    _lib.check(L.gsvc_param_means(_lib.ptr(f), f.numel(), _lib.ptr(sc), sc.numel(), 0 if pc.decoded_version else 1, _lib.ptr(o), o.numel(),
                                  _lib.ptr(scratch), _lib.ptr(out), _lib.current_stream(f.device)), "gsvc_param_means")
    return out


def _rate_many(pc, seg, feat, grid_scaling, grid_offsets, offset_masks, Q_feat, Q_scaling, Q_offsets, ec, ec_row=None, sel=None):
    """Sampled rate of the R renders of a batch (reference guassian.py:73-132 per render): 5 % of the visible anchors that have
    a live offset, bits of their features / scalings / offsets under the entropy context, four normalised means per render.
    ``ec_row``: row -> row of ``ec`` when the entropy context was evaluated once per distinct anchor.  ``sel``: the sample's
    rows when a StepPlan drew it ahead of the step (same rule: Bernoulli(SAMPLE_RATE) and mask_anchor).

    Written as a few wide tensor operations: the three attribute groups side by side ([n_sel, 3] steps, [3, n_sel] clamp
    bounds, [R, 3] sums) instead of one chain of small launches per group and per render — this function used to issue
    ~120 kernels forward and as many backward to produce sixteen numbers."""
    from .entropy_models import CLAMP_STEPS, _GaussianBits
    K, R, dev = pc.n_offsets, seg.R, feat.device
    fused = (feat.is_cuda and R <= 16 and all(isinstance(q, torch.Tensor) and q.numel() == feat.shape[0] for q in (Q_feat, Q_scaling, Q_offsets))
             and not switches.NO_FUSED_RATE)
    compact = hasattr(ec, "rows")        # SampledEntropyContext: mean / scale exist for the sampled rows only, row i = i-th sampled row
    if fused and sel is not None and offset_masks.dtype == torch.float32:
        # everything on the device in a dozen launches: the parameter means, the sampled bits summed per render (k_rate_sample),
        # and their normalisation with the keep rates and sample sizes (k_rate_normalise)
        sel_ec = sel if ec_row is None else ec_row.index_select(0, sel)
        if compact:
            ec, sel_ctx = ec.rows(sel_ec), _iota(dev, sel.shape[0])
        else:
            sel_ctx = None if ec_row is None else sel_ec
        S = _SampledRate.apply(feat, grid_scaling, grid_offsets, offset_masks, Q_feat, Q_scaling, Q_offsets, ec.mean_feat, ec.scale_feat,
                               ec.mean_scaling, ec.scale_scaling, ec.mean_offsets, ec.scale_offsets, sel,
                               sel_ctx, _param_means(pc), seg.bounds, K)
        out, total = _RateNorm.apply(S, offset_masks, sel, seg.bounds, (float(feat.shape[1]), float(grid_scaling.shape[1]), float(3 * K)), K)
        packs = [RatePack(bit_per_param=out[r, 0], bit_per_feat_param=out[r, 1], bit_per_scaling_param=out[r, 2],
                          bit_per_offsets_param=out[r, 3]) for r in range(R)]
        packs[0].bit_per_param_sum = total        # sum over the renders: the fitting loss takes this one scalar (one backward node)
        return packs
    with torch.no_grad():
        live = (torch.sum(offset_masks, dim=1)[:, 0] > 0)
        kr = seg.sums(live) / seg.counts_t.clamp_min(1)                  # keep rate per render
        if sel is None:
            chosen = (torch.rand_like(feat[:, 0]) <= SAMPLE_RATE) & live
            sel = chosen.nonzero(as_tuple=False).squeeze(1)
        sel_seg = seg.seg_id.index_select(0, sel)
        # selected rows per render: the selection is sorted by row, so the counts are differences of its positions of the bounds
        edges = torch.searchsorted(sel, seg.bounds_t)
        n_sel = (edges[1:] - edges[:-1]).to(torch.float32)
    sel_ec = sel if ec_row is None else ec_row.index_select(0, sel)
    sel_ctx = None if ec_row is None else sel_ec
    if compact:
        ec = ec.rows(sel_ec)
        sel_ec = sel_ctx = _iota(dev, sel.shape[0])
    if fused:
        # fused path (csrc/rate.hip k_rate_sample): gathers, per-render clamp bounds, offset mask and per-render sums in one launch
        xm = _param_means(pc)
        S = _SampledRate.apply(feat, grid_scaling, grid_offsets, offset_masks, Q_feat, Q_scaling, Q_offsets, ec.mean_feat, ec.scale_feat,
                               ec.mean_scaling, ec.scale_scaling, ec.mean_offsets, ec.scale_offsets, sel,
                               sel_ctx, xm, seg.bounds, K)
        dims = host_values([float(feat.shape[1]), float(grid_scaling.shape[1]), float(3 * K)], dev)
        N = n_sel.unsqueeze(1) * dims
        per = S / N * kr.unsqueeze(1)
        tot = S.sum(dim=1) / N.sum(dim=1) * kr
        return [RatePack(bit_per_param=tot[r], bit_per_feat_param=per[r, 0], bit_per_scaling_param=per[r, 1],
                         bit_per_offsets_param=per[r, 2]) for r in range(R)]
    # steps of the chosen rows, the three groups side by side; clamp bounds = x_mean -+ 15000 * (mean step of the row's render)
    Q3 = torch.cat([Q_feat.reshape(-1, 1), Q_scaling.reshape(-1, 1), Q_offsets.reshape(-1, 1)], dim=1).index_select(0, sel)
    Q3T = Q3.t().contiguous()                                                     # [3, n_sel]: one contiguous row per group
    with torch.no_grad():
        qm = torch.zeros(R, 3, device=dev).index_add_(0, sel_seg, Q3) / n_sel.clamp_min(1).unsqueeze(1)
        qmT = qm.index_select(0, sel_seg).t()                                     # [3, n_sel]
        xm = torch.stack([pc._anchor_feat.mean(), pc.get_scaling.mean(), pc._offset.mean()]).unsqueeze(1)
        lo, hi = (xm - CLAMP_STEPS * qmT).contiguous(), (xm + CLAMP_STEPS * qmT).contiguous()
    xs = (feat, grid_scaling, grid_offsets.view(-1, 3 * K))
    means = (ec.mean_feat, ec.mean_scaling, ec.mean_offsets)
    scales = (ec.scale_feat, ec.scale_scaling, ec.scale_offsets)
    bits = [_GaussianBits.apply(xs[g].index_select(0, sel), means[g].index_select(0, sel_ec), scales[g].index_select(0, sel_ec),
                                Q3T[g], 0.0, lo[g], hi[g], True) for g in range(3)]
    bits[2] = bits[2] * offset_masks.index_select(0, sel).repeat(1, 1, 3).view(-1, 3 * K)
    S = torch.zeros(R, 3, device=dev).index_add_(0, sel_seg, torch.stack([b.sum(dim=1) for b in bits], dim=1))   # [R, 3] bit sums
    dims = host_values([float(bits[0].shape[1]), float(bits[1].shape[1]), float(bits[2].shape[1])], dev)
    N = n_sel.unsqueeze(1) * dims                                                 # coded elements per render and group
    per = S / N * kr.unsqueeze(1)
    tot = S.sum(dim=1) / N.sum(dim=1) * kr
    return [RatePack(bit_per_param=tot[r], bit_per_feat_param=per[r, 0], bit_per_scaling_param=per[r, 1],
                     bit_per_offsets_param=per[r, 2]) for r in range(R)]


_RATE_STREAM = {}
SMALL_WORK_MIN_ROWS = 150_000      # visible anchor rows of a step from which the small-work stream is used: below, the step is
                                   # host-bound (configs[3]: 58 k rows) and the extra events / stream switches only cost host time
# ... unless the Trainer has MEASURED which side bounds its steps (round 5): the row count says nothing about the rasterizer's load, and
# late in a fit a 100 k-anchor model (80 k rows) spends 7 ms per step on the GPU — its Gaussians cover 57 tiles each — while the host
# needs 5.  The host only ever blocks at the step's two waits (the plan's counts, the renders' counters): what it blocks there per step
# is the GPU's lead over it.  None = no measurement yet (the row count decides); set by gsvc_amd.train.Trainer.
class StepContext:
    """The measurement belongs to the Trainer that made it (as module globals two Trainers of one process clobbered
    each other's): ``gpu_bound_hint`` (None = no measurement yet: the row count decides) and the seconds the host spent blocked in the
    step's waits since its Trainer last read them.  A Trainer makes its own context current for the duration of each of its steps
    (``step_context``); callers without a Trainer (render loops, tests) see the neutral default."""
    __slots__ = ("gpu_bound_hint", "host_blocked_s", "flips")

    def __init__(self):
        self.gpu_bound_hint, self.host_blocked_s, self.flips = None, 0.0, 0


_DEFAULT_CONTEXT = StepContext()
step_context = _DEFAULT_CONTEXT


def gpu_bound(rows) -> bool:
    """Should this step spend host time (events, stream switches, separate launches) to save GPU time?"""
    # (the measurement only ever ADDS steps to the GPU-bound side: a caller that synchronises every step — float(loss) — never lets the
    # host block in the step's own waits, and must not lose what the row count alone already grants)
    return rows >= SMALL_WORK_MIN_ROWS or bool(step_context.gpu_bound_hint)


def _blocked_wait(event):
    import time
    t0 = time.perf_counter()
    event.synchronize()
    step_context.host_blocked_s += time.perf_counter() - t0


def small_work_stream(dev):
    """The stream on which a fitting step issues the small launch-bound pieces that do not read the rasterizer's output (the sampled
    rate here, the regularisers / optical-flow term / table bits in gsvc_amd/train.py), so that they run under the compositing
    kernels instead of in front of or behind them."""
    rs = _RATE_STREAM.get(dev.index)
    if rs is None:
        rs = _RATE_STREAM[dev.index] = torch.cuda.Stream(device=dev)
    return rs


class two_stream_backward:
    """Scope of a backward whose graph has nodes on the small-work stream: a parameter that both that stream's nodes and the step's
    stream's nodes use gets its gradient from two streams — intended (the engine orders them), so the engine's warning about it
    is switched off for the duration of THIS backward only (it was switched off process-wide at first use)."""

    _depth = 0          # nested scopes (a backward inside a backward's hook): only the outermost one switches the warning back on

    def __enter__(self):
        self._set = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if self._set is not None:
            two_stream_backward._depth += 1
            self._set(False)

    def __exit__(self, *exc):
        if self._set is not None:
            two_stream_backward._depth -= 1
            # torch offers a setter only (no getter to save the caller's choice): a caller who has switched the warning off for the
            # whole process says so once with GSVC_KEEP_STREAM_WARNING_OFF=1 and this scope then leaves it off
            if two_stream_backward._depth == 0 and not os.environ.get("GSVC_KEEP_STREAM_WARNING_OFF"):
                self._set(True)


def record_on(stream, *objs):
    """``record_stream(stream)`` on every CUDA tensor in ``objs`` (tensors, lists / tuples of them, objects whose attributes hold
    them; None is skipped).  A tensor allocated on the step's stream and READ on the small-work stream must tell the caching
    allocator so: autograd releases saved tensors as soon as a node's backward is enqueued, and a block returned to the step's
    pool while the other stream's kernel is still queued could be handed to — and overwritten by — the next main-stream kernel
    (ADVICE round 4).  The allocator then holds the block back until the work queued on ``stream`` at the time of the free is done."""
    seen = set()

    def walk(o, depth):
        if o is None or id(o) in seen:
            return
        seen.add(id(o))
        if isinstance(o, torch.Tensor):
            if o.is_cuda:
                o.record_stream(stream)
        elif isinstance(o, (list, tuple)):
            for v in o:
                walk(v, depth)
        elif isinstance(o, dict):
            for v in o.values():
                walk(v, depth)
        elif depth > 0 and hasattr(o, "__dict__") and not isinstance(o, torch.nn.Module):
            for v in vars(o).values():
                walk(v, depth - 1)
    for o in objs:
        walk(o, 1)


def finish_deferred_rate(gss_list):
    """The sampled rate of a batched TRAINING_ENTROPY generation pass whose caller asked for it late (``defer_rate``): issued now —
    the caller has queued the rasterizer's launches — on a stream of its own that waits only for what the rate reads (the event
    recorded behind the noise quantisation).  The current stream waits for the result; the rate's autograd nodes belong to that
    stream, so their backward (the first nodes of the step's backward) runs next to the rasterizer's backward instead of in
    front of it.  Fills the renders' ``bit_per_*`` and the batch's ``bit_per_param_sum``; no-op when nothing was deferred."""
    batch = getattr(gss_list[0], "batch", None) if gss_list else None
    pending = getattr(batch, "deferred_rate", None)
    if pending is None:
        return
    rate, ready, reads = pending
    batch.deferred_rate = None
    dev = gss_list[0].xyz.device
    main = torch.cuda.current_stream(dev)
    rs = small_work_stream(dev)
    rs.wait_event(ready)
    record_on(rs, *reads)                    # allocated on this stream, read (and saved for a backward that runs) on that one
    with torch.cuda.stream(rs):
        packs = rate()
    main.wait_stream(rs)
    total = getattr(packs[0], "bit_per_param_sum", None)
    for t in [total] + [v for p in packs for v in (p.bit_per_param, p.bit_per_feat_param, p.bit_per_scaling_param, p.bit_per_offsets_param)]:
        if isinstance(t, torch.Tensor):
            t.record_stream(main)            # allocated on the rate stream, read by the loss on this one
    for gs, p in zip(gss_list, packs):
        gs.bit_per_param, gs.bit_per_feat_param = p.bit_per_param, p.bit_per_feat_param
        gs.bit_per_scaling_param, gs.bit_per_offsets_param = p.bit_per_scaling_param, p.bit_per_offsets_param
    batch.bit_per_param_sum = total


def _entropy_context_distinct(pc, anchor_all, vis, plan=None, sampled=False):
    """Entropy context of the batch's rows, evaluated once per DISTINCT anchor.

    The context (hash-grid lookup + the three EntropyParamsNets, reference scene/gaussian_model.py:1569-1597) is a
    function of the anchor position only, and the renders of one step — two adjacent frames x two opposite views —
    see almost the same anchors: the concatenated rows name each anchor ~4 times.  Evaluating the grid and the nets
    on the distinct anchors and gathering rows from that (the per-row step sizes) or composing indices (the 5 % rate
    sample) is the same arithmetic on a quarter of the rows; the gradients of the duplicates meet in the gathers'
    backward.  Returns (context over the distinct anchors, row -> distinct index), or (context, None) when nothing
    repeats.  ``sampled``: the training form (gsvc_amd.model.SampledEntropyContext: quantisation steps on every row, the
    priors' mean / scale deferred to the rows of the rate sample)."""
    calc = pc.calc_entropy_context_sampled if sampled else pc.calc_entropy_context
    A = anchor_all.shape[0]
    if vis.numel() == 0:
        return calc(anchor_all.index_select(0, vis)), None
    if plan is not None:           # the distinct list and the anchor -> row map were computed with the visibility test
        if plan.distinct.shape[0] == vis.shape[0]:
            return calc(anchor_all.index_select(0, vis)), None
        return calc(anchor_all.index_select(0, plan.distinct)), plan.pos.index_select(0, vis)
    present = torch.zeros(A, dtype=torch.bool, device=vis.device)
    present[vis] = True
    distinct = present.nonzero(as_tuple=False).squeeze(1)
    if distinct.shape[0] == vis.shape[0]:
        return calc(anchor_all.index_select(0, vis)), None
    pos = torch.cumsum(present, dim=0) - 1
    return calc(anchor_all.index_select(0, distinct)), pos.index_select(0, vis)


def _embed_rows(pc, frames, anchor, seg):
    """pe = [embed_time(cam z of the row's render) | embed(anchor z - cam z)] for the concatenated rows (reference
    guassian.py:225-230).  Detached CUDA anchors with the standard embedders: one kernel (csrc/generate.hip)."""
    et, ez = getattr(pc.embed_time_fn, "__self__", None), getattr(pc.embed_fn, "__self__", None)
    cams = [float(f.cam_pos[-1]) for f in frames]
    if (anchor.is_cuda and not anchor.requires_grad and seg.R <= 16 and et is not None and ez is not None
            and all(getattr(e, "include_input", False) and getattr(e, "input_dims", 0) == 1 for e in (et, ez))
            and et.freq_bands.numel() == ez.freq_bands.numel() <= 24
            and all(torch.equal(e.freq_bands, 2.0 ** torch.arange(e.freq_bands.numel(), dtype=e.freq_bands.dtype)) for e in (et, ez))):
        import ctypes as C
        from . import _lib
        F = int(ez.freq_bands.numel())
        pe = torch.empty(seg.rows, 2 * (2 * F + 1), device=anchor.device, dtype=torch.float32)
        bounds = (C.c_int64 * (seg.R + 1))(*seg.bounds)
        cz = (C.c_float * seg.R)(*cams)
        _lib.check(_lib.lib().gsvc_embed_pe(_lib.ptr(anchor.contiguous()), bounds, cz, seg.R, F, _lib.ptr(pe),
                                            _lib.current_stream(anchor.device)), "gsvc_embed_pe")
        return pe
    cam_z = host_values(cams, anchor.device, anchor.dtype)
    cam_z_row = cam_z.index_select(0, seg.seg_id).unsqueeze(1)
    ob_view = anchor[:, 2:] - cam_z_row
    return torch.cat([pc.embed_time_fn(cam_z_row), pc.embed_fn(ob_view)], dim=1)


class _ViewRows(torch.autograd.Function):
    """Rows of (frame, distinct anchor) -> rows of (view, visible anchor): what the generators computed once per frame, handed to
    the frame's two opposite views.  ``row_of[r]`` = frame row of view row r; ``src_a[j]`` / ``src_b[j]`` = the view rows behind
    frame row j (-1: that view does not see the anchor) — the maps of _film_rows.  The backward adds the two views' gradients
    (a gather per frame row, no atomics: csrc/generate.hip gsvc_pair_rows_sum)."""

    @staticmethod
    def forward(ctx, t, row_of, src_a, src_b):
        ctx.save_for_backward(src_a, src_b)
        ctx.rows_u = t.shape[0]
        return t.index_select(0, row_of)

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        src_a, src_b = ctx.saved_tensors
        g = g.contiguous()
        C = g[0].numel() if g.shape[0] else 1
        out = torch.empty((ctx.rows_u,) + tuple(g.shape[1:]), device=g.device, dtype=torch.float32)
        _lib.check(_lib.lib().gsvc_pair_rows_sum(_lib.ptr(g), _lib.ptr(src_a), _lib.ptr(src_b), ctx.rows_u, C, _lib.ptr(out),
                                                 _lib.current_stream(g.device)), "gsvc_pair_rows_sum")
        return out, None, None, None


def _film_rows(pc, frames, plan, vis, seg, anchor_all):
    """Rows of the generators' FiLM networks when the step's views come in opposite pairs (frame f seen from both sides: same
    camera z, so the same condition for the same anchor — reference frame_cube/frame.py:18-43, guassian.py:225-230): one row per
    (frame, distinct visible anchor) instead of one per (view, anchor).  Returns (cond_film, row_of, src_a, src_b) for
    gsvc_amd.mlp.generate_all, or None (no plan, an odd number of views, views that are not such pairs, or GSVC_NO_FILM_SHARE)."""
    R = seg.R
    if (plan is None or R % 2 != 0 or plan.distinct is None or switches.NO_FILM_SHARE
            or any(float(frames[2 * i].cam_pos[-1]) != float(frames[2 * i + 1].cam_pos[-1]) for i in range(R // 2))):
        return None
    dev = vis.device
    D, A = int(plan.distinct.shape[0]), plan._A
    if D == 0 or seg.rows == 0:
        return None
    F = R // 2
    with torch.no_grad():
        # FiLM row of chain row (view r, anchor a) = (r // 2) * D + position of a in the distinct list; the chain rows behind
        # FiLM row (f, a): the row of a in views 2 f and 2 f + 1 (scan of the flattened view masks - 1), -1 if unseen
        Mflat, c = plan.ranks
        if vis.is_cuda and R <= 16 and Mflat.dtype == torch.bool and c.dtype == torch.int64 and seg.rows < 2 ** 31 and F * D < 2 ** 31:
            import ctypes as C
            from . import _lib
            maps = torch.empty(seg.rows + 2 * F * D, dtype=torch.int32, device=dev)
            row_of, src_a, src_b = maps[:seg.rows], maps[seg.rows:seg.rows + F * D], maps[seg.rows + F * D:]
            _lib.check(_lib.lib().gsvc_film_row_maps(_lib.ptr(vis.contiguous()), (C.c_int64 * (R + 1))(*seg.bounds), R, _lib.ptr(plan.pos.contiguous()),
                                                 D, A, _lib.ptr(Mflat.contiguous()), _lib.ptr(c.contiguous()), _lib.ptr(plan.distinct.contiguous()),
                                                 _lib.ptr(row_of), C.c_void_p(src_a.data_ptr()), C.c_void_p(src_b.data_ptr()),
                                                 _lib.current_stream(dev)), "gsvc_film_row_maps")
        else:
            frame_of_row = torch.div(seg.seg_id, 2, rounding_mode="floor")
            row_of = (plan.pos.index_select(0, vis) + frame_of_row * D).to(torch.int32)
            base = (torch.arange(R, device=dev, dtype=torch.int64) * A).view(R, 1) + plan.distinct.view(1, D)      # [R, D] flat positions
            rows = torch.where(Mflat.index_select(0, base.view(-1)), c.index_select(0, base.view(-1)) - 1,
                               torch.full((), -1, device=dev, dtype=c.dtype)).view(F, 2, D).to(torch.int32)
            src_a, src_b = rows[:, 0, :].reshape(-1).contiguous(), rows[:, 1, :].reshape(-1).contiguous()
        seg_f = _Segments([D] * F, dev)
        cond_film = _embed_rows(pc, [frames[2 * i] for i in range(F)], anchor_all.index_select(0, plan.distinct).repeat(F, 1), seg_f)
    return cond_film, row_of, src_a, src_b


class _LazyTrunks(dict):
    """``generator_trunks`` evaluated on first access: a decoder loop whose generation takes the whole-network chain kernels
    never reads them (three GEMM passes over ALL anchors saved per call)."""

    def __init__(self, pc):
        super().__init__()
        self._pc = pc

    def __missing__(self, key):
        with torch.no_grad():
            for name in ("get_opacity_mlp", "get_color_mlp", "get_cov_mlp"):
                dict.__setitem__(self, name, getattr(self._pc, name).trunk(self._pc._anchor_feat))
        return dict.__getitem__(self, key)


def generator_trunks(pc):
    """Frame-independent half of the three generator MLPs for ALL anchors: ``linear2(GELU(linear1(anchor_feat)))``
    (reference scene/gaussian_model.py:168-196 evaluates it per render).  Valid while the parameters do not change and
    the generation mode leaves the features as stored (decoding): the decoder loop computes it once per video — on first use."""
    return _LazyTrunks(pc)


def generate_neural_gaussians_many(frames, pc, visible_masks, mode=GenerateMode.TRAINING_FULL_PRECISION, dense=False,
                                   anchors=None, trunks=None, plan=None, defer_rate=False):
    """`generate_neural_gaussians` for R renders at once; returns a list of R GeneratedGaussians.

    ``dense=True`` skips the "opacity > 0" compaction: every visible anchor contributes all K Gaussians, ``mask``
    still marks the ones with opacity > 0, and the rasterizer culls the rest itself (csrc/raster_fwd.hip K1) — the
    image, the gradients and every masked statistic are the same, but no tensor shape depends on device data, so
    the step runs without a host synchronisation between the visibility test and the optimizer.
    ``concatenated_all`` is not materialised in that form (``world_xyz`` carries what the optical-flow loss reads).
    ``anchors``: precomputed ``pc.get_anchor`` without autograd (anchor positions then receive no gradient)."""
    R = len(frames)
    K = pc.n_offsets
    with region('gen.visible_index'):
        vis_list = plan.resolve().vis_list if plan is not None else [_as_index(m) for m in visible_masks]
    dev = vis_list[0].device
    seg = _Segments([v.shape[0] for v in vis_list], dev)
    with region('gen.gather'):
        vis = plan.vis_all if (plan is not None and plan.vis_all is not None) else torch.cat(vis_list)
        anchor_all = pc.get_anchor if anchors is None else anchors
        anchor = anchor_all.index_select(0, vis)
        ranks = plan.ranks if plan is not None else None
        # a fitting step gathers (offsets, scaling, masks) behind the generators' forward, in every phase: see _gather_rows
        # (data parallel too since round 3: the reducer launches its collectives in an order the ranks agree on)
        late_rows = (mode in (GenerateMode.TRAINING_ENTROPY, GenerateMode.TRAINING_FULL_PRECISION, GenerateMode.TRAINING_QUANTIZED,
                              GenerateMode.TRAININ_STE_ENTROPY)
                     and trunks is None and torch.is_grad_enabled() and not switches.NO_LATE_ROWS)
        # FULL_PRECISION / STE_ENTROPY draw nothing per render, and the two opposite views of a frame share the camera position:
        # their generators see the same row for the same anchor (reference guassian.py:225-273 evaluates them once per view, with
        # equal results).  The features, the conditioning and the four networks then run once per (frame, distinct visible anchor)
        # — the rows the FiLM networks already use (_film_rows: the two sides of a frame see all but ~0.2 % of the same anchors,
        # adjacent frames ~94 %) —; _ViewRows hands the outputs to the views' rows and adds the two views' gradients.
        share = None
        if (late_rows and mode in (GenerateMode.TRAINING_FULL_PRECISION, GenerateMode.TRAININ_STE_ENTROPY) and vis.is_cuda
                and plan is not None and plan.distinct is not None and plan.distinct.shape[0] != vis.shape[0] and not switches.NO_VIEW_SHARE):
            share = _film_rows(pc, frames, plan, vis, seg, anchor_all)
    from . import mlp as _mlp
    # few rows: the step's time is the host's (SMALL_WORK_MIN_ROWS) -> the MLP layer launches prefer fewer, multi-product launches
    _mlp.host_bound_step = bool(vis.is_cuda and torch.is_grad_enabled() and not gpu_bound(seg.rows) and "GSVC_MANY_MIN_ROWS" not in os.environ)
    gens = [getattr(pc, n) for n in ("get_opacity_mlp", "get_color_mlp", "get_cov_mlp")]
    deform_mods = list(pc.get_deform_mlp) if isinstance(pc.get_deform_mlp, torch.nn.Sequential) else []
    deform_linears = deform_mods[0::2]
    seg_u = feat_u = pe_u = None
    if share is not None:
        pe_u, row_of, src_a, src_b = share
        row_of = row_of.long()
        F_, D_ = R // 2, int(plan.distinct.shape[0])
        with region('gen.gather'):
            (feat_u,) = _gather_rows(pc, plan.distinct.repeat(F_), None, parts="feat")
        if (all(hasattr(g, "film") and hasattr(g, "out_linear") for g in gens)
                and all(isinstance(m, torch.nn.Linear) for m in deform_linears) and all(isinstance(m, torch.nn.GELU) for m in deform_mods[1::2])
                and _mlp.chain_usable(feat_u, pe_u, gens, deform_linears)):
            seg_u = _Segments([D_] * F_, dev)
        else:
            share = feat_u = pe_u = None          # not the production widths: per view, layer by layer
    with region('gen.gather'):
        if seg_u is not None:
            feat = grid_offsets = grid_scaling = offset_masks = None
        elif late_rows:
            (feat,) = _gather_rows(pc, vis, ranks, parts="feat")
            grid_offsets = grid_scaling = offset_masks = None
        else:
            feat, grid_offsets, grid_scaling, offset_masks = _gather_rows(pc, vis, ranks)
    rates = [RatePack() for _ in range(R)]
    deferred = []
    Q_feat, Q_scaling, Q_offsets = BASE_Q_FEAT, BASE_Q_SCALING, BASE_Q_OFFSETS
    time_sub = 0
    # the conditioning input and the generators' FiLM networks depend on the anchors' z only: issued FIRST, their twelve large
    # GEMMs keep the GPU busy while the host queues the entropy context's many small launches (a step starts host-bound)
    with region('gen.embed'):
        pe = _embed_rows(pc, frames, anchor, seg) if seg_u is None else None
    films = {}
    # the three generators + mlp_deform as whole-network chain kernels (gsvc_amd.mlp.generate_all) when the widths are the
    # production ones: then nothing is issued ahead (8 forward launches in all)
    # (decoding hands in the cached feature-only half of the generators: with the production widths the forward-only chain kernels,
    # which read the features once and keep a row block in registers through all layers, are faster per frame and are preferred)
    chain = ((trunks is None or (not torch.is_grad_enabled() and not switches.NO_DECODE_CHAIN))
             and all(hasattr(g, "film") and hasattr(g, "out_linear") for g in gens)
             and all(isinstance(m, torch.nn.Linear) for m in deform_linears) and all(isinstance(m, torch.nn.GELU) for m in deform_mods[1::2])
             and (seg_u is not None or _mlp.chain_usable(feat, pe, gens, deform_linears)))
    if chain:
        trunks = None
    if trunks is None and not chain:
        with region('gen.film'):
            for name in ("get_opacity_mlp", "get_color_mlp", "get_cov_mlp"):
                net = getattr(pc, name)
                films[name] = net.film_nets(pe) if hasattr(net, "film_nets") else None

    rows_work = None       # the phase's work on (offsets, scaling, masks) rows: behind the generators when they are gathered late
    if mode in (GenerateMode.TRAINING_FULL_PRECISION, GenerateMode.DECODING_AS_IS):
        pass
    elif mode == GenerateMode.TRAINING_QUANTIZED:
        feat = _seg_noise_quant(feat, Q_feat, seg)

        def rows_work():      # the draws keep their order (features, scalings, offsets): nothing in between draws
            nonlocal grid_offsets, grid_scaling
            grid_scaling = _seg_noise_quant(grid_scaling, Q_scaling, seg)
            grid_offsets = _seg_noise_quant(grid_offsets, Q_offsets, seg)
    elif mode == GenerateMode.TRAINING_ENTROPY:
        # the priors' mean / scale are read at the rate sample's rows only (reference guassian.py:99-113): their networks run on
        # those rows (SampledEntropyContext); GSVC_CTX_ALL_ROWS=1 keeps the all-rows form (A/B timing, the equivalence test)
        sampled_ctx = feat.is_cuda and not switches.CTX_ALL_ROWS
        with region('gen.entropy_context'):
            ec, ec_row = _entropy_context_distinct(pc, anchor_all, vis, plan, sampled=sampled_ctx)
        with region('gen.noise_quant'):
            if sampled_ctx and all(isinstance(q, (int, float)) for q in (Q_feat, Q_scaling, Q_offsets)) and all(a.dtype == torch.float32 for a in ec.q_raw):
                Q_feat, Q_scaling, Q_offsets = _QRows.apply(ec.q_raw[0], ec.q_raw[1], ec.q_raw[2], ec_row, Q_feat, Q_scaling, Q_offsets, True)
            elif (feat.is_cuda and all(isinstance(q, (int, float)) for q in (Q_feat, Q_scaling, Q_offsets))
                    and all(a.dtype == torch.float32 and a.numel() == ec.Q_feat_adj.numel() for a in (ec.Q_feat_adj, ec.Q_scaling_adj, ec.Q_offsets_adj))):
                Q_feat, Q_scaling, Q_offsets = _QRows.apply(ec.Q_feat_adj, ec.Q_scaling_adj, ec.Q_offsets_adj, ec_row, Q_feat, Q_scaling, Q_offsets)
            else:
                rows_of = (lambda t: t) if ec_row is None else (lambda t: t.index_select(0, ec_row))  # noqa: E731
                Q_feat, Q_scaling, Q_offsets = (Q_feat * rows_of(ec.Q_feat_adj), Q_scaling * rows_of(ec.Q_scaling_adj),
                                                Q_offsets * rows_of(ec.Q_offsets_adj))
            feat = _seg_noise_quant(feat, Q_feat, seg)

        def rows_work():
            # the same noise draws in the same order (features, scalings, offsets) wherever this runs: nothing between the
            # features' quantisation and this draws from the generator
            nonlocal grid_offsets, grid_scaling, rates
            with region('gen.noise_quant'):
                grid_scaling = _seg_noise_quant(grid_scaling, Q_scaling, seg)
                grid_offsets = _seg_noise_quant(grid_offsets, Q_offsets.unsqueeze(1), seg)
            rates = rate_now_or_later()
    elif mode == GenerateMode.TRAININ_STE_ENTROPY:
        ec, ec_row = _entropy_context_distinct(pc, anchor_all, vis, plan, sampled=vis.is_cuda and not switches.CTX_ALL_ROWS)
        rows_of = (lambda t: t) if ec_row is None else (lambda t: t.index_select(0, ec_row))  # noqa: E731
        Q_feat, Q_scaling, Q_offsets = (Q_feat * rows_of(ec.Q_feat_adj).detach(), Q_scaling * rows_of(ec.Q_scaling_adj).detach(),
                                        Q_offsets * rows_of(ec.Q_offsets_adj).detach())
        # the three parameters' means in one pass (gsvc_param_means: what the rate's clamp reads too) instead of three reductions
        # and an exp pass over every anchor; they only place the +-15 000-step bounds
        pm = _param_means(pc) if (vis.is_cuda and not switches.NO_FUSED_STE) else None
        mean_feat = pm[0:1] if pm is not None else pc._anchor_feat.mean()
        if seg_u is None:
            feat = _seg_ste(feat, Q_feat, seg, mean_feat)
        else:
            # the step of (frame, anchor) is the anchor's; the clamp's integer bounds (mean / mean step -+ 15000, truncated) come from
            # the frame's rows instead of each view's (they differ by an anchor in 500): the same integers unless the centre sits
            # within 1e-3 of an integer, and the clamp only acts 15000 steps from the mean
            feat_u = _seg_ste(feat_u, (BASE_Q_FEAT * ec.Q_feat_adj.detach()).repeat(R // 2, 1), seg_u, mean_feat)
            feat = feat_u.index_select(0, row_of)      # (detached: the rate reads it per view row)

        def rows_work():
            nonlocal grid_offsets, grid_scaling, rates
            grid_scaling = _seg_ste(grid_scaling, Q_scaling, seg, pm[1:2] if pm is not None else pc.get_scaling.mean())
            grid_offsets = _seg_ste(grid_offsets, Q_offsets.unsqueeze(1), seg, pm[2:3] if pm is not None else pc._offset.mean())
            rates = rate_now_or_later()
    else:
        raise ValueError(f"Unknown mode {mode}")

    def rate_now_or_later():
        """The sampled rate of the two entropy phases (runs once the row tensors are in their final form)."""
        def rate():
            with region('gen.rate'):
                return _rate_many(pc, seg, feat, grid_scaling, grid_offsets, offset_masks, Q_feat, Q_scaling, Q_offsets, ec, ec_row,
                                  sel=plan.sel if plan is not None else None)
        if (defer_rate and dense and late_rows and plan is not None and plan.sel is not None and vis.is_cuda
                and gpu_bound(seg.rows) and not switches.NO_RATE_OVERLAP):
            # the sampled rate — three small networks on ~10 k rows and a dozen reductions, launch-bound — is issued by the caller
            # BEHIND the rasterizer's launches, on its own stream (finish_deferred_rate): it runs under the compositing kernels
            # forward and, its autograd nodes living on that stream, under the rasterizer's backward
            reads = [feat, grid_scaling, grid_offsets, offset_masks, Q_feat, Q_scaling, Q_offsets, ec_row, plan.sel, ec]
            ready = torch.cuda.current_stream(vis.device).record_event()
            if switches.RATE_EARLY:
                # issued NOW (still on its own stream, so it runs beside whatever follows): the rate's autograd nodes are then
                # OLDER than the rasterizer's, and the engine — which runs younger nodes first — launches the rasterizer's backward
                # before the rate's ~40 small launches instead of behind them
                rs = small_work_stream(vis.device)
                rs.wait_event(ready)
                record_on(rs, *reads)
                with torch.cuda.stream(rs):
                    packs = rate()
                deferred.append((lambda: packs, ready, ()))
                return rates
            deferred.append((rate, ready, reads))
            return rates
        return rate()

    def rows_now():
        nonlocal grid_offsets, grid_scaling, offset_masks
        if late_rows:
            grid_offsets, grid_scaling, offset_masks = _gather_rows(pc, vis, ranks, parts="rows")
        if rows_work is not None:
            rows_work()
    if not late_rows:
        rows_now()

    rows = seg.rows
    with region('gen.mlps'):
        if trunks is not None:      # decoding: the feature-only half of the generators was evaluated once for all anchors
            if mode not in (GenerateMode.DECODING_AS_IS, GenerateMode.TRAINING_FULL_PRECISION):
                raise ValueError("generator trunks are only valid when the features are used as stored")
            op_raw = pc.get_opacity_mlp.head(trunks["get_opacity_mlp"].index_select(0, vis), pe)
            color = pc.get_color_mlp.head(trunks["get_color_mlp"].index_select(0, vis), pe).reshape(rows * K, 3)
            scale_rot = pc.get_cov_mlp.head(trunks["get_cov_mlp"].index_select(0, vis), pe).reshape(rows * K, 7)
        elif chain:
            if seg_u is not None:
                op_raw, color, scale_rot, neural_offset = (_ViewRows.apply(o, row_of, src_a, src_b)
                                                           for o in _mlp.generate_all(gens, deform_linears, feat_u, pe_u, film=None))
            else:
                op_raw, color, scale_rot, neural_offset = _mlp.generate_all(gens, deform_linears, feat, pe, film=_film_rows(pc, frames, plan, vis, seg, anchor_all))
            color, scale_rot, neural_offset = color.reshape(rows * K, 3), scale_rot.reshape(rows * K, 7), neural_offset.reshape(rows * K, 3)
        else:
            gen = lambda name: (getattr(pc, name)(feat, pe, film=films[name]) if films.get(name) is not None  # noqa: E731
                                else getattr(pc, name)(feat, pe))
            op_raw = gen("get_opacity_mlp")
            color = gen("get_color_mlp").reshape(rows * K, 3)
            scale_rot = gen("get_cov_mlp").reshape(rows * K, 7)
        if not chain:
            neural_offset = pc.get_deform_mlp(torch.cat([feat, pe], dim=1)).reshape(rows * K, 3)
    if late_rows:
        rows_now()
    if dense:
        # opacity mask, sigmoid scaling, normalised rotation, world position, bound clamp: one kernel (csrc/generate.hip)
        neural_opacity, mask, scaling, rot, world, xyz = _GenTail.apply(
            op_raw.reshape(-1), offset_masks.reshape(-1), grid_offsets.reshape(-1, 3), neural_offset, scale_rot, grid_scaling,
            anchor, K, pc.bound_min_host, pc.bound_max_host)
        # per-render pieces by split (one cat in backward per tensor, instead of a zero-fill + copy + add per slice)
        sizes = [c * K for c in seg.counts]
        parts = [t.split(sizes, dim=0) for t in (xyz, color, neural_opacity, scaling, rot, world)]
        from types import SimpleNamespace
        batch = SimpleNamespace(scaling=scaling, neural_opacity=neural_opacity, mask=mask, seg_offsets=[b * K for b in seg.bounds], vis=vis,
                                xyz=xyz, color=color, rot=rot, world=world,      # the un-split tensors: rasterize_many works on their row ranges
                                bit_per_param_sum=getattr(rates[0], "bit_per_param_sum", None),
                                deferred_rate=deferred[0] if deferred else None,
                                small_work=bool(vis.is_cuda and gpu_bound(seg.rows) and not switches.NO_RATE_OVERLAP))
        out = []
        for r, gs in enumerate(seg.slices(K)):
            out.append(GeneratedGaussians(
                xyz=parts[0][r], color=parts[1][r], opacity=parts[2][r], scaling=parts[3][r], rot=parts[4][r],
                neural_opacity=parts[2][r], visable_mask=visible_masks[r], mask=mask[gs],
                bit_per_param=rates[r].bit_per_param, bit_per_feat_param=rates[r].bit_per_feat_param,
                bit_per_scaling_param=rates[r].bit_per_scaling_param, bit_per_offsets_param=rates[r].bit_per_offsets_param,
                concatenated_all=None, time_sub=time_sub, visible_index=vis_list[r], world_xyz=parts[5][r], batch=batch))
        return out
    neural_opacity = op_raw.reshape(-1, 1) * offset_masks.view(-1, 1)
    mask = (neural_opacity > 0.0).view(-1)
    offsets = grid_offsets.view(-1, 3) + neural_offset
    per_anchor = torch.cat([grid_scaling, anchor], dim=-1)
    concatenated_all = torch.cat([per_anchor.repeat_interleave(K, dim=0), color, scale_rot, offsets], dim=-1)
    alive_idx = mask.nonzero(as_tuple=False).squeeze(1)
    alive = concatenated_all.index_select(0, alive_idx)
    scaling_rep, anchor_rep = alive[:, 0:6], alive[:, 6:9]
    color_a, scale_rot_a, offsets_a = alive[:, 9:12], alive[:, 12:19], alive[:, 19:22]
    scaling = scaling_rep[:, 3:] * torch.sigmoid(scale_rot_a[:, :3])
    rot = pc.rotation_activation(scale_rot_a[:, 3:7])
    xyz = torch.clamp(anchor_rep + offsets_a * scaling_rep[:, :3], pc.x_bound_min, pc.x_bound_max)
    opacity = neural_opacity.index_select(0, alive_idx)
    # split the compacted Gaussians back into the R renders (one host read of R+1 offsets)
    edges = host_values([b * K for b in seg.bounds], dev)
    cut = torch.searchsorted(alive_idx, edges).tolist()
    out = []
    for r, (rs, gs) in enumerate(zip(seg.slices(), seg.slices(K))):
        a = slice(cut[r], cut[r + 1])
        out.append(GeneratedGaussians(
            xyz=xyz[a], color=color_a[a], opacity=opacity[a], scaling=scaling[a], rot=rot[a],
            neural_opacity=neural_opacity[gs], visable_mask=visible_masks[r], mask=mask[gs],
            bit_per_param=rates[r].bit_per_param, bit_per_feat_param=rates[r].bit_per_feat_param,
            bit_per_scaling_param=rates[r].bit_per_scaling_param, bit_per_offsets_param=rates[r].bit_per_offsets_param,
            concatenated_all=concatenated_all[gs], time_sub=time_sub))
    return out
