"""The fitting step: 4 renders (2 adjacent frames x 2 opposite views), distortion + regularisers + optical-flow
consistency + lambda * rate, backward, densification statistics, Adam.

Same loss and order of operations as the step body of reference pipeline/train.py:325-462,559-581; logging,
evaluation, tensorboard and the codec calls around it are out of scope.  Differences in mechanism: no per-step
`.item()` / synchronize (the reference syncs ~10x per step for logging), the hash-table bit count stays on the
device, and under torch.distributed each rank steps on its own frame pair and gradients are averaged with one
collective (gsvc_amd/dist.py).  Anchor growing/pruning runs at the reference's cadence (gsvc_amd/densify.py).
"""
from __future__ import annotations

import os
import random
from dataclasses import dataclass

import contextlib

import torch

from . import dist as gdist
from . import switches
from .generate import GenerateMode, region
from .loss_utils import calc_optical_loss, render_regs, ssim_l1, ssim_l1_pair
from .optim import FusedAdam
from .ortho_gaussian_renderer import plan_views, render, render_many
from .rasterizer import resolve_deferred
from .train_util import TrainingController


EARLY_PLAN_MIN_ROWS = 150_000      # visible anchor rows of a step's views from which Trainer._early_tail is used


@dataclass
class StepOutput:
    """What a step hands back — all of it graph-free: the loss and images are detached, and so are the tensors inside ``renders``
    (their backward has run; ``_release_graph``).  Differentiate through ``render()`` / ``render_many()`` directly instead."""
    loss: torch.Tensor
    image1: torch.Tensor
    image2: torch.Tensor
    renders: tuple
    active_gaussians: torch.Tensor  # sum over the 4 renders of Gaussians with radius > 0 (device scalar)
    frame_idx: int


def _release_graph(renders):
    """The step's backward has run: what the caller keeps of the step (the renders, their generated Gaussians, the batch) must not
    keep the autograd graph alive.  A caller that holds the previous step's output while the next step runs — every training loop
    does: ``out = trainer.step(i)`` — otherwise keeps that graph's AccumulateGrad nodes, the engine reuses them with the streams
    they were created on, and the extra waits it then inserts undo the overlap of the small-work stream (measured: 6.6 -> 7.1 ms)."""
    T = torch.Tensor
    seen = set()

    def strip(obj):
        d = getattr(obj, "__dict__", None)
        if d is None or id(obj) in seen:
            return
        seen.add(id(obj))
        for k, v in d.items():
            if type(v) is T and v.grad_fn is not None:
                d[k] = v.detach()
    for r in renders:
        strip(r)
        g = r.generated_gaussians
        if g is not None:
            strip(g)
            strip(getattr(g, "batch", None))


def hash_grid_bits(pc):
    """Bernoulli code length of the binarised hash tables (reference pipeline/train.py:456:
    ``get_binary_vxl_size((get_encoding_params() + 1) / 2)``).  The count of ones is taken table by table from the binarised
    tables the grid lookups already made ({-1, +1}: ones = (sum + n) / 2) instead of from a 14 MB concatenation of them."""
    tables = grid_tables(pc)
    if not (pc.ste_binary and all(hasattr(g, "embeddings") for g in tables)):
        return get_binary_vxl_size_device((pc.get_encoding_params() + 1) / 2)
    from .encodings import CountBits, binarized_tables
    embs = binarized_tables(tables)
    cache0 = getattr(tables[0], "step_cache", None) or {}
    if "counts_all" in cache0 and cache0["counts_all"][1] == tuple(id(g) for g in tables):
        # every table was binarised by one launch that also counted its +1 entries: the term and its gradient are written
        # on that count vector (two one-thread launches; the tables receive it inside their straight-through backward)
        return CountBits.apply(cache0["counts_all"][0], sum(e.numel() for e in embs))
    counts = [(getattr(g, "step_cache", None) or {}).get("ones") for g in tables]
    return _TableBits.apply(torch.cat(counts) if all(c is not None for c in counts) else None, *embs)


class _TableBits(torch.autograd.Function):
    """bits = n1 (-log2 p) + n0 (-log2 (1 - p)) + 32 over the {-1, +1} tables, p = n1 / n clamped to [1e-6, 1 - 1e-6]
    (reference utils/encodings.py:34-51).  d bits / d n1 = log2((1 - p) / p) — the terms through p cancel exactly where p is not
    clamped and vanish where it is — and n1 = (sum of the entries + n) / 2, so every table entry receives the same number:
    the backward is one scalar expression and an expand instead of a dozen one-element kernels each way."""

    @staticmethod
    def forward(ctx, counts, *tables):
        total = sum(t.numel() for t in tables)
        # counts: the +1 entries of every table, counted by the kernel that binarised it (else: from the tables' sums)
        ones = counts.sum() if counts is not None else (torch.stack([t.sum() for t in tables]).sum() + total) / 2
        p = torch.clamp(ones / total, min=1e-6, max=1 - 1e-6)
        ctx.save_for_backward(p)
        ctx.shapes = [t.shape for t in tables]
        return ones * (-torch.log2(p)) + (total - ones) * (-torch.log2(1 - p)) + 32

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        coef = g * 0.5 * torch.log2((1 - p) / p)
        return (None,) + tuple(coef.expand(sh) for sh in ctx.shapes)


def grid_tables(pc):
    enc = pc.encoding_xyz
    return [enc.encoding_xyz, enc.encoding_xy, enc.encoding_xz, enc.encoding_yz] if pc.use_2D else [enc]


def get_binary_vxl_size_device(binary_vxl):
    total = binary_vxl.numel()
    ones = binary_vxl.sum()
    p = torch.clamp(ones / total, min=1e-6, max=1 - 1e-6)
    return ones * (-torch.log2(p)) + (total - ones) * (-torch.log2(1 - p)) + 32


def _mean_over_selected(values, r):
    """Mean over the Gaussians a render generated (opacity > 0): all rows of a compacted result, the rows marked
    by ``selection_mask`` of an un-compacted one."""
    if not r.dense:
        return values.mean()
    w = r.selection_mask.to(values.dtype)
    return (values * w).sum() / w.sum()


class Trainer:
    def __init__(self, gaussians, dataset, opt, pipe, model_params, seed: int = 0, batched: bool = True, prefetch: bool = True,
                 shard_optimizer=None):
        if shard_optimizer is not None:      # removed in round 4 (the index-range sharded Adam): accepted and ignored for one release
            import warnings
            warnings.warn("Trainer(shard_optimizer=...) is ignored: the optimizer is replicated (gsvc_amd/dist.py)", DeprecationWarning,
                          stacklevel=2)
        self.batched = batched
        # prefetch: the next step's frame pair is drawn, its visibility test run and every data-dependent index list of its
        # generation pass queued at the END of a step (gsvc_amd.generate.StepPlan): the next step then starts with one wait
        # for nine counts instead of six device round trips with an idle GPU
        self.prefetch = prefetch and batched and not switches.NO_PREFETCH      # env: A/B timing only
        self._plan = self._plan_idx = self._plan_mode = None
        self.pc, self.dataset, self.opt, self.pipe, self.mp = gaussians, dataset, opt, pipe, model_params
        if gaussians._anchor.is_cuda:
            from .hostbind import bind_to_device
            bind_to_device(gaussians._anchor.device)      # the step is host-bound at small sizes: stay on the GPU's NUMA node
        self.controller = TrainingController(opt)
        self.controller.step()  # iterations are 1-based
        self.rng = random.Random(seed + gdist.rank())
        self.lo, self.hi = gdist.frame_shard(dataset.len_z_frames)
        bg = [1, 1, 1] if model_params.white_background else [0, 0, 0]
        self.background = torch.tensor(bg, dtype=torch.float32)  # host tensor: the kernels take bg by value
        # anchors are trained with learning rate 0 in GSVC (position_lr_init = position_lr_final = 0): their gradient
        # changes nothing, so the batched step does not compute it unless a non-zero rate is configured
        self.anchor_grad = bool(getattr(opt, "position_lr_init", 0.0) or getattr(opt, "position_lr_final", 0.0))
        gaussians.anchor_static = not self.anchor_grad      # the quantised anchors may be cached between steps (prefilter_geometry)
        self.reducer = gdist.GradReducer()
        gdist.plan_group()      # created HERE, where every rank stands at the same point (creating a group is itself collective)
        # GSVC_DP_ZOWN=1: the per-anchor tensors are OWNED by z-range (gsvc_amd.dist.ZRangeOwnership) — gradients of the halo rows go
        # to their owners, updated rows come back — instead of being summed on every replica
        self._zown = None
        if gdist.zrange_enabled() and not (batched and not self.anchor_grad):
            import sys
            sys.stderr.write("gsvc_amd.train: GSVC_DP_ZOWN=1 ignored (needs the batched step and anchors with learning rate 0: an anchor that moves "
                             "could change its owner): replicated exchange\n")
        if gdist.zrange_enabled() and batched and not self.anchor_grad:
            self._zown = gdist.ZRangeOwnership(dataset.len_z_frames, dataset.scale, model_params.threshold)
            assert abs(dataset[self.lo].z - (self.lo - dataset.len_z_frames / 2) / dataset.scale) < 1e-6, "frame z convention"
        gaussians._zown = self._zown      # gsvc_amd.generate._param_means reads the owners' means from it
        self._mask_reg_weight = 0.0
        self._ovf_handle = None
        # which side bounds the steps, measured (gsvc_amd.generate.StepContext.gpu_bound_hint): the host's blocked time per step, smoothed
        self._blocked_ema = None
        from . import generate as _gen
        self._ctx = _gen.StepContext()      # this trainer's own measurement (made current for the duration of each of its steps)

    def close(self):
        """Give the calling thread its CPU affinity back (``__init__`` narrowed it to the GPU's NUMA node: gsvc_amd/hostbind.py;
        GSVC_NO_CPU_BIND=1 never narrows it)."""
        if self.pc._anchor.is_cuda:
            from .hostbind import unbind
            unbind(self.pc._anchor.device)
        from . import generate as _gen
        if _gen.step_context is self._ctx:
            _gen.step_context = _gen._DEFAULT_CONTEXT

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _two_views(self, frame, mode, retain_grad):
        f = render(frame, self.pc, self.pipe, self.background, retain_grad=retain_grad, mode=mode)
        frame.view_matrix, frame.view_matrix_s = frame.view_matrix_s, frame.view_matrix
        b = render(frame, self.pc, self.pipe, self.background, retain_grad=retain_grad, mode=mode)
        image = (f.rendered_image + torch.flip(b.rendered_image, dims=(-1,))) / 2
        return f, b, image

    def _update_bound(self):
        """The host blocks only at the step's two waits, and what it blocks there is how far the GPU is behind it: over ~0.4 ms per
        step the GPU bounds the step and host-side work that shortens the GPU's (the small-work stream, the early plan, separate
        layer launches) pays; under ~0.1 ms the host does and it does not.  In between the last decision stands (the measures
        themselves move the balance).  The first steps (allocator growth, first launches) are not counted."""
        from . import generate as _gen
        ctx = self._ctx
        _gen.step_context = ctx          # whatever this step's generation / waits read and add to is this trainer's
        b = ctx.host_blocked_s
        ctx.host_blocked_s = 0.0
        self._steps_seen = getattr(self, "_steps_seen", 0) + 1
        if switches.NO_ADAPTIVE_BOUND or switches.DETERMINISTIC:
            # (deterministic mode: a wall-clock measurement must not pick between the separate and the multi-product layer launches —
            # they round differently — so the row count alone decides, the same way in every run)
            ctx.gpu_bound_hint = None
            return
        if self._steps_seen <= 8:
            return
        self._blocked_ema = b if self._blocked_ema is None else 0.8 * self._blocked_ema + 0.2 * b
        self._ema_n = getattr(self, "_ema_n", 0) + 1
        if self._steps_seen < 16 or self._ema_n < 8:
            # (the average needs a few steps of its own: a fit that leaves the deterministic mode, or follows a synchronisation, starts
            # it with ONE step's wait — bench.py's first timed steps after its frozen pretrain were decided by a 0.008 ms sample and ran
            # host-bound for 57 steps; until then the row count decides)
            return
        before = ctx.gpu_bound_hint
        if self._blocked_ema > 4e-4:
            ctx.gpu_bound_hint = True
        elif self._blocked_ema < 1e-4:
            ctx.gpu_bound_hint = False
        if ctx.gpu_bound_hint != before:
            # a wall-clock measurement changes which launches the step takes from here on: on record, so that runs can be compared
            ctx.flips += 1
            self.bound_log = getattr(self, "bound_log", [])[-31:] + [(self._steps_seen, bool(ctx.gpu_bound_hint), 1e3 * self._blocked_ema)]

    def step(self, iteration: int, frame_idx: int | None = None) -> StepOutput:
        self._update_bound()
        if frame_idx is None and self._plan_idx is not None:
            frame_idx = self._plan_idx              # drawn (from the same generator, in the same order) at the end of the last step
        self._early = None
        out = self._step(iteration, frame_idx)
        if out is None:      # a rasterizer instance buffer overflowed (capacity now raised): repeat the step
            self.repeated_steps = getattr(self, "repeated_steps", 0) + 1
            dropped = None
            if self._early is not None:
                # the guarded early update saw the overflow word and wrote nothing: only its step counts moved.  The plan it
                # queued is for the NEXT frame pair; the repeat draws its own lists.
                self.pc.optimizer.rewind(self._early[0])
                dropped = self._early[3]          # its count exchange is in flight: the tensors stay alive until the repeat is done
                self._early = None
            elif self.prefetch and self.pc._anchor.is_cuda and gdist.active():
                # Whether the early tail ran is a rank-LOCAL fact (its row threshold and reducer.complete() depend on the rank's
                # own views), and it queued a plan = one collective on the plan group.  A rank that did not run it answers with a
                # matching no-op exchange, so that every rank has issued exactly ONE plan-group collective per attempted step:
                # otherwise the repeat's end-of-step plan on this rank would pair with the other rank's dropped one and every
                # later distinct_cap would be one step out of phase between the ranks.
                dropped = gdist.plan_group_noop(self.pc._anchor.device)
            # the repeat draws its own index lists on EVERY rank (no plan -> dense gradient exchange everywhere): a rank that kept
            # its plan would exchange rows while a rank that dropped its early plan all-reduces
            self._plan = None
            self.pc.optimizer.zero_grad(set_to_none=True)
            out = self._step(iteration, frame_idx if frame_idx is not None else self._last_idx, early=False)     # the same frame pair again
            if out is None:
                raise RuntimeError("rasterizer instance buffer overflowed twice in a row")
            if dropped is not None and getattr(dropped, "_gmax_work", None) is not None:
                dropped._gmax_work.wait()
            del dropped
        self.controller.step()
        if self._early is not None:          # the next step's plan was queued from inside the backward (_early_tail)
            _, self._plan_idx, self._plan_mode, self._plan = self._early
            self._early = None
            return out
        self._plan = self._plan_idx = None
        if self.prefetch and self.pc._anchor.is_cuda:
            with torch.no_grad(), region('step.plan'):
                self._plan_idx = self.rng.randint(self.lo, max(self.lo, self.hi - 1))
                self._prefetch_frames(self._plan_idx)
                self._plan_mode = self.controller.render_mode
                self._plan = plan_views(self._views(self._plan_idx), self.pc, self.pipe, self.background, self._plan_mode)
        return out

    def sync_replicas(self, moments: bool = True):
        """Under z-range ownership (GSVC_DP_ZOWN=1) a replica is only fresh inside its own block + halo: call this (on every rank)
        before anything reads the whole model — evaluation, estimate_final_bits, stream encoding, a checkpoint."""
        if self._zown is not None:
            self._zown.sync_full(self.pc, moments=moments)
        self.pc._zown = None      # outside a step the whole-tensor means are the model's own again (gsvc_amd.generate._param_means)

    def _zown_names(self, mode):
        # TRAININ_STE_ENTROPY renders from detached attributes: only _mask receives a gradient (and only it moves)
        return ("_mask",) if mode == GenerateMode.TRAININ_STE_ENTROPY else gdist.PER_ANCHOR

    def _prefetch_frames(self, idx):
        """A dataset that keeps its pictures in host memory (gsvc_amd.frame.HostResidentCube) starts uploading the next step's pair
        now, on its copy stream (reference pipeline/train.py:407-408 uploads them inside the step)."""
        pf = getattr(self.dataset, "prefetch", None)
        if pf is not None:
            pf(idx)

    def _add_mask_reg(self):
        """Sparse data-parallel exchange: the mask regulariser 5e-4 * mean(sigmoid(_mask)) touches EVERY row of ``_mask`` with the
        same number on every rank (the parameters are replicas), so it stays out of the exchanged gradient — its loss term is taken
        of a detached copy — and its gradient, a closed form, is added on every rank after the exchange."""
        if self._mask_reg_weight:
            pc = self.pc
            with torch.no_grad():
                sg = torch.sigmoid(pc._mask)
                reg = sg * (1.0 - sg) * (self._mask_reg_weight / pc._mask.numel())
                if pc._mask.grad is None:
                    pc._mask.grad = reg
                else:
                    pc._mask.grad.add_(reg)
            self._mask_reg_weight = 0.0

    def _early_tail(self, renders, params):
        """Called from inside the backward, the moment the gradients of ``params`` — ``_scaling`` and ``_mask``, those of the two
        that this phase's step gives a gradient — are complete (the last
        contribution is the late row gather's backward: gsvc_amd.generate._gather_rows): these two tensors are all the next
        step's visibility test and rate sample read that this step changes, so they are updated NOW and the next step's plan
        is queued behind them — its counts reach the host ~2 ms before the GPU finishes this step, and the host, which runs
        ahead of the GPU, never has to let the queue drain at a step boundary.  (With the plan queued behind the optimizer the
        host waited ~1.4 ms per step for it and the GPU then idled until the next step's first kernels arrived.)
        The update is guarded by the rasterizer's overflow words: a step that must be repeated changes nothing."""
        from .train_util import render_mode_at
        pc = self.pc
        with torch.no_grad():
            guards = [r.raster_state.binning[4:8].view(torch.int32) for r in renders]
            if gdist.active():
                # data parallel: the two gradients are final once their collectives (launched by the reducer's hooks, which run
                # before this one) have completed; a step in which they have not gone out yet (no agreed order in the very first
                # one) takes the ordinary end-of-step path.  "Some rank overflowed" guards the update as well.
                if not self.reducer.complete(params):
                    return
                self._add_mask_reg()
                guards = [self._ovf_handle[2]]        # the MAX over the ranks of the four local words (this rank's included)
            pc.optimizer.step(only=params, guards=guards)
            self.early_steps = getattr(self, "early_steps", 0) + 1
            idx = self.rng.randint(self.lo, max(self.lo, self.hi - 1))
            self._prefetch_frames(idx)
            mode = render_mode_at(self.controller.current_iteration + 1, self.opt)
            plan = plan_views(self._views(idx), pc, self.pipe, self.background, mode) if mode is not None else None
            self._early = (params, idx, mode, plan)

    def _sparse_dp(self, plan):
        """Row-sparse gradient exchange of the per-anchor tensors: needs the step plan (the rank's distinct visible anchors and the
        largest such count over the ranks) and the replicated Adam; GSVC_DP_SPARSE=0 keeps the dense all-reduce."""
        if not (plan is not None and gdist.active() and self.reducer.enabled and not self.anchor_grad and self._zown is None
                and getattr(plan, "distinct_cap", None) is not None and switches.DP_SPARSE != "0"):
            return False
        # every rank receives the other ranks' row lists (all-gather, padded to the largest): worth it while those rows are fewer
        # than what a ring all-reduce of the dense tensors moves (2 (W - 1) / W of the anchors) — two ranks always, eight ranks
        # only when a rank sees less than a quarter of the anchors.  The same decision on every rank (cap is their maximum).
        return switches.DP_SPARSE == "1" or gdist.sparse_rows_pay(gdist.world_size(), int(self.pc._anchor.shape[0]),
                                                                                  plan.distinct_cap)

    @staticmethod
    def _gpu_bound(rows):
        from . import generate as _gen
        return rows >= EARLY_PLAN_MIN_ROWS or bool(_gen.step_context.gpu_bound_hint)

    def _views(self, frame_idx):
        """The step's four views: (frame, frame seen from the opposite side) of the two adjacent frames."""
        import copy
        views = []
        for fr in (self.dataset[frame_idx], self.dataset[frame_idx + 1]):
            back = copy.copy(fr)
            back.view_matrix, back.view_matrix_s = fr.view_matrix_s, fr.view_matrix
            views += [fr, back]
        return views

    def _adjust_anchor(self, iteration):
        """Densify / prune (reference pipeline/train.py:567-569).  Under data parallelism every rank must take the same
        decisions: the statistics are summed over ranks first and the random thinning uses a per-iteration seed."""
        opt = self.opt
        if self._zown is not None:
            self._zown.sync_full(self.pc)      # the decisions read every row: whole replicas first (once per update_interval steps)
        gdist.adjust_anchor_replicated(self.pc, iteration, check_interval=opt.update_interval, success_threshold=opt.success_threshold,
                                       grad_threshold=opt.densify_grad_threshold, min_opacity=opt.min_opacity)
        if self._zown is not None:
            self._zown.ensure(self.pc)         # new anchors, new row lists (existing anchors keep their owner: z does not move)

    def _step(self, iteration: int, frame_idx: int | None = None, early: bool = True):
        for g in grid_tables(self.pc):
            g.step_cache = {}           # one binarisation of each hash table per step (gsvc_amd.encodings.GridEncoder.embeddings)
        try:
            return self._step_body(iteration, frame_idx, early)
        finally:
            for g in grid_tables(self.pc):
                g.step_cache = None

    def _step_body(self, iteration: int, frame_idx: int | None = None, early: bool = True):
        opt, pc = self.opt, self.pc
        dev = pc.device
        pc.update_learning_rate(iteration)
        if frame_idx is None:
            frame_idx = self.rng.randint(self.lo, max(self.lo, self.hi - 1))
        self._last_idx = frame_idx
        frame1, frame2 = self.dataset[frame_idx], self.dataset[frame_idx + 1]
        mode = self.controller.render_mode
        retain_grad = opt.update_until > iteration >= 0

        zown = self._zown if (self._zown is not None and self.reducer.enabled) else None
        pc._zown = zown
        if zown is not None:
            zown.update_means(pc)
        if self.batched:
            # one generation pass for the 4 views (frame1 f/b, frame2 f/b), then 4 rasterizations
            plan = self._plan if (self._plan is not None and self._plan_idx == frame_idx and self._plan_mode == mode and
                                  self._plan.matches(pc)) else None
            views = plan.frames if plan is not None else self._views(frame_idx)
            frame1, frame2 = views[0], views[2]
            with region('step.render_many'):
                r1f, r1b, r2f, r2b = render_many(views, self.pc, self.pipe, self.background, retain_grad=retain_grad, mode=mode,
                                                 dense=True, anchor_grad=self.anchor_grad, plan=plan)
            image1 = image2 = None         # the two-view frames are formed inside the SSIM kernels below (ssim_l1_pair)
            # replicas: "did any rank's instance buffer overflow" is reduced right behind the forward kernels, so that
            # the end-of-step check does not have to wait for the backward (overflow word = second int32 of a binning blob)
            ovf_handle = gdist.any_rank_start([r.raster_state.binning[4:8].view(torch.int32) for r in (r1f, r1b, r2f, r2b)])
            self._ovf_handle = ovf_handle
        else:
            r1f, r1b, image1 = self._two_views(frame1, mode, retain_grad)
            r2f, r2b, image2 = self._two_views(frame2, mode, retain_grad)
        renders = (r1f, r1b, r2f, r2b)
        ready = getattr(self.dataset, "ready", None)
        if ready is not None:          # pictures uploaded on a copy stream (HostResidentCube): their first readers come now
            ready(frame_idx)
            ready(frame_idx + 1)
        gt1 = frame1.image.to(dev).permute(0, 2, 1)
        gt2 = frame2.image.to(dev).permute(0, 2, 1)
        if image1 is None:
            ssim1, l1_1, image1 = ssim_l1_pair(r1f.rendered_image, r1b.rendered_image, gt1.contiguous())
            ssim2, l1_2, image2 = ssim_l1_pair(r2f.rendered_image, r2b.rendered_image, gt2.contiguous())
        else:
            ssim1, l1_1 = ssim_l1(image1, gt1.contiguous())
            ssim2, l1_2 = ssim_l1(image2, gt2.contiguous())
        # the loss is a weighted sum of scalar terms: they are collected and combined with one stack + dot (instead of
        # one tiny kernel per +, *, and their backward nodes)
        terms, weights, const = [l1_1, l1_2, ssim1, ssim2], [1.0 - opt.lambda_dssim] * 2 + [-opt.lambda_dssim] * 2, 2.0 * opt.lambda_dssim
        batch = getattr(r1f.generated_gaussians, "batch", None) if self.batched else None
        # the terms below read the generation pass's tensors and parameters, not the images: a dozen small launches each way, issued
        # on the small-work stream behind the generation pass's event so that they run under the compositing kernels (their backward
        # too: autograd runs a node on its forward's stream) instead of behind the image losses
        side = None
        if batch is not None and getattr(batch, "generated_event", None) is not None and not switches.NO_RATE_OVERLAP:
            from .generate import small_work_stream
            side = small_work_stream(dev)
            side.wait_event(batch.generated_event)
        n_image_terms = len(terms)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            if batch is not None:
                regs = render_regs(batch.scaling, batch.neural_opacity, batch.mask, batch.seg_offsets)
                terms += [regs[0], regs[1]]
            else:
                terms += [sum(_mean_over_selected(r.scaling.prod(dim=1), r) for r in renders),
                          sum((1 - r.neural_opacity).mean() for r in renders)]
            weights += [opt.scaling_reg, opt.opacity_reg]
            if opt.optical_lambda != 0:
                flow = self.dataset.get_optical_flow(frame_idx)
                terms.append(calc_optical_loss(r1f, r1b, r2f, r2b, flow, self.dataset.x_min, self.dataset.y_min,
                                               self.dataset.scale, self.dataset.width, self.dataset.height, pc.n_offsets))
                weights.append(opt.optical_lambda)
            if self.controller.entropy_constrained:
                assert all(r.entropy_constrained for r in renders)
                denom = pc._anchor.shape[0] * (pc.feat_dim + 6 + 3 * pc.n_offsets)
                # sparse data-parallel exchange (below): the regulariser's dense gradient is added after the exchange (_add_mask_reg)
                sparse_dp = self._sparse_dp(plan if self.batched else None) or zown is not None
                self._mask_reg_weight = 5e-4 if sparse_dp else 0.0
                rate_sum = getattr(batch, "bit_per_param_sum", None)
                # the renders' rates enter with one weight: their sum, when the batched generation already formed it, is one term
                terms += ([rate_sum] if rate_sum is not None else [r.bit_per_param for r in renders]) + \
                         [hash_grid_bits(pc), (zown.mask_sigmoid_mean if (zown is not None and zown.mask_sigmoid_mean is not None) else
                                               torch.mean(torch.sigmoid(pc._mask.detach() if sparse_dp else pc._mask)))]
                weights += [opt.lmbda] * (1 if rate_sum is not None else 4) + [opt.lmbda / denom, 5e-4]
        if side is not None:
            # what the terms above read of the step's stream's allocations (and saved for their backward on the small-work stream)
            from .generate import record_on
            record_on(side, batch.scaling, batch.neural_opacity, batch.mask, getattr(batch, "world", None), getattr(batch, "vis", None),
                      [g.step_cache for g in grid_tables(pc)], flow if opt.optical_lambda != 0 else None)
            main_stream = torch.cuda.current_stream(dev)
            main_stream.wait_stream(side)
            for t in terms[n_image_terms:]:
                if t.is_cuda:
                    t.record_stream(main_stream)       # allocated on the small-work stream, combined on this one
        key = tuple(weights)
        if getattr(self, "_w_key", None) != key:      # the weights change only with the training phase
            from .generate import host_values
            self._w_key, self._w = key, host_values(weights, dev, torch.float32)
        w = self._w
        loss = torch.dot(torch.stack([t.reshape(()) for t in terms]), w) + const
        owned = {id(getattr(pc, n)) for n in gdist.PER_ANCHOR} if zown is not None else ()
        self.reducer.arm([p for g in pc.optimizer.param_groups for p in g["params"] if id(p) not in owned], phase=mode)
        use_rows = bool(self.batched and self._sparse_dp(plan))
        if use_rows:
            # the per-anchor gradients are non-zero only in the rows of this rank's distinct visible anchors: exchanged as rows
            self.reducer.set_sparse(plan.distinct, plan.distinct_cap, [pc._offset, pc._mask, pc._anchor_feat, pc._scaling])
        else:
            self.reducer.set_sparse(None, 0, [])
        if zown is not None and gdist.rank() == 0 and getattr(self, "_logged_exchange", None) != "zown":
            import sys
            zown.ensure(pc)
            self._logged_exchange = "zown"
            sent, got = zown.halo_rows()
            sys.stderr.write(f"gsvc_amd.train: per-anchor tensors owned by z-range: rank 0 owns {int(zown.own_idx.shape[0])} of {int(pc._anchor.shape[0])} "
                             f"anchors, sends {sent} halo rows of gradients and returns {got} rows of parameters per step, {gdist.world_size()} ranks\n")
        elif zown is None and gdist.active() and gdist.rank() == 0 and getattr(self, "_logged_exchange", None) != use_rows and plan is not None:
            import sys
            self._logged_exchange = use_rows
            sys.stderr.write(f"gsvc_amd.train: per-anchor gradient exchange = {'rows of the distinct visible anchors (cap ' + str(plan.distinct_cap) + ')' if use_rows else 'dense all-reduce'}"
                             f" at {int(pc._anchor.shape[0])} anchors, {gdist.world_size()} ranks\n")
        handles = []
        if (early and self.batched and self.prefetch and pc._anchor.is_cuda
                and (not gdist.active() or (self.reducer.enabled and self.reducer._order is not None and zown is None))
                and not self.anchor_grad      # a trained anchor tensor moves behind the early plan's visibility test
                and mode is not None and iteration < opt.iterations and not self.controller.gaussian_adjust_anchor
                and isinstance(pc.optimizer, FusedAdam) and not switches.NO_EARLY_PLAN
                and pc._scaling.requires_grad and pc._mask.requires_grad
                # it pays when the GPU, not the host, bounds the step (the hook's work costs ~1 ms of host time more on the
                # autograd thread than at the end of the step: a 6 k-row step went 8.3 -> 9.7 ms, the 200 k-row step 11.95 -> 11.3)
                and (switches.EARLY_PLAN or self._gpu_bound(sum(int(r.visible_index.shape[0]) for r in
                                                                                (x.generated_gaussians for x in renders))))):
            # STE_ENTROPY renders from detached attributes (reference guassian.py:205-207): _scaling receives no gradient there,
            # is not changed by the step, and its hook would never run
            waited = [pc._mask] if mode == GenerateMode.TRAININ_STE_ENTROPY else [pc._scaling, pc._mask]
            pending = [len(waited)]

            def arrived(_p):
                pending[0] -= 1
                if pending[0] == 0:
                    self._early_tail(renders, waited)
            handles = [p.register_post_accumulate_grad_hook(arrived) for p in waited]
        try:
            from .generate import two_stream_backward
            from .mlp import wgrad_overlap
            # a GPU-bound step (the one that uses the small-work stream) also sends the chain networks' weight gradients to their own
            # stream; a host-bound one only pays for the events and the allocator's stream records (configs[3]: +0.2 ms of host time)
            gpu_bound_step = side is not None or getattr(batch, "small_work", False)
            with region('step.backward'), (two_stream_backward() if gpu_bound_step else contextlib.nullcontext()), \
                    (wgrad_overlap(dev) if gpu_bound_step else contextlib.nullcontext()):
                loss.backward()
        finally:
            for h in handles:
                h.remove()
        if dev.type == "cuda" and (side is not None or getattr(batch, "small_work", False)):
            # backward() returns with every node's kernels queued on the node's forward stream.  A parameter that only the small-work
            # stream touched (the priors' networks: used by the deferred rate alone) has its gradient accumulated THERE, and nothing in
            # the engine orders that before what this stream does next — the optimizer, the reducer, the statistics.  Without this
            # wait Adam occasionally read those gradients half written (seen as run-to-run drift of the rate model, not as a crash).
            from .generate import small_work_stream
            torch.cuda.current_stream(dev).wait_stream(small_work_stream(dev))
        done = getattr(self.dataset, "step_done", None)
        if done is not None:
            done()      # every reader of the ground-truth slots (image losses, optical term — forward and backward, this stream and the
                        # small-work stream, which this one has just waited for) is queued: the slots' next upload may start behind here
        with region('step.reducer_finish'):
            self.reducer.finish()
            if zown is not None:
                zown.exchange_grads(pc, self._zown_names(mode))
            self._add_mask_reg()

        if self.batched:
            # the only host synchronisation of the step after the visibility test: the 4 renders' instance counters
            with region('step.resolve_counters'):
                _, overflowed = resolve_deferred([r.raster_state for r in renders])
            if gdist.any_rank_finish(ovf_handle, overflowed):      # replicas repeat the step together (their collectives must pair up)
                return None
            for r in renders:
                r.num_rendered = r.raster_state.counters()[0]
        with torch.no_grad():
            if self.controller.gaussian_statis:
                if self.batched:
                    with region('step.statis'):
                        pc.training_statis_many(renders)
                else:
                    for r in renders:
                        pc.training_statis(r)
            adjusted = False
            if self.controller.gaussian_adjust_anchor:
                self._adjust_anchor(iteration)
                adjusted = True
            if self.controller.clean_denorm:
                pc.opacity_accum = pc.offset_gradient_accum = pc.offset_denom = None
            if iteration < opt.iterations:
                with region('step.optimizer'):
                    pc.optimizer.step()
                    pc.optimizer.zero_grad(set_to_none=True)
                    if zown is not None:
                        zown.refresh_params(pc, self._zown_names(mode))
        active = sum(r.active_gaussains for r in renders)
        _release_graph(renders)      # at every size: a StepOutput never carries the step's autograd graph (see _release_graph)
        return StepOutput(loss=loss.detach(), image1=image1.detach(), image2=image2.detach(), renders=renders,
                          active_gaussians=active, frame_idx=frame_idx)
