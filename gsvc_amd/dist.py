"""Multi-GPU data parallelism for the fitting loop: frames of one video shard across the GPUs of a node,
parameters are replicated, gradients are summed over the ranks with RCCL over xGMI: the hash tables and MLPs by all-reduce, the
per-anchor tensors as (row index, row) lists of each rank's distinct visible anchors while that moves fewer bytes (GradReducer).

The reference is single-GPU (no torch.distributed / NCCL anywhere); this layer is new.
One process per GPU (torchrun / torch.distributed.run), backend "nccl" (= RCCL on ROCm) on GPUs and "gloo" in
the CPU tests.  Each step samples an independent adjacent-frame pair (reference pipeline/train.py:336-343), so
rank r draws its pairs from its own contiguous block of frames (its z-slab working set stays local) and the
only exchange is the gradient of the shared parameters: per-anchor tensors (96 floats/anchor), the four hash
tables and the MLPs — one flat bucket, one collective (54 MB at 100k anchors; ring all-reduce over 7 xGMI links
is per-link bound, so one large message beats many small ones).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK (no-op for a single process).
    Returns (rank, world_size, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
        log_ranks()
    return rank, world, local


def log_ranks():
    """One stderr line on rank 0 naming the communicator the step's collectives will run on ("nccl" IS RCCL on ROCm), so that a
    scaling record can confirm how many ranks really took part."""
    if active() and rank() == 0:
        import sys
        b = dist.get_backend()
        sys.stderr.write(f"gsvc_amd.dist: {'RCCL' if b == 'nccl' else b} ranks = {world_size()} (backend {b}; one rank per GPU, frames sharded "
                         f"by contiguous blocks, gradients: per-anchor tensors as rows or dense by sparse_rows_pay, tables + MLPs dense)\n")
        sys.stderr.flush()


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def active() -> bool:
    """Is the data-parallel machinery on?  More than one rank — or GSVC_DP_FORCE=1 with an initialised process group of ONE rank
    (test knob: every collective of the step then runs, as the identity, on a real one-rank RCCL communicator of a single-GPU
    box: device / dtype / contiguity rules of the backend, stream ordering, group creation)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("GSVC_DP_FORCE") == "1"


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


_plan_group = None


def plan_group():
    """A process group of its own for the step plan's count exchange (gsvc_amd.generate.StepPlan): a plan is built once per step
    on every rank, but at a rank-dependent moment relative to the gradient collectives (from inside the backward when the early
    tail runs, after it otherwise) — on the default group that would break the one order every rank must issue collectives in;
    a communicator of its own only needs ITS sequence to agree."""
    global _plan_group
    if not active():
        return None
    if _plan_group is None:
        _plan_group = dist.new_group()
    return _plan_group


class _NoopPlan:
    def __init__(self, t, work):
        self._gmax, self._gmax_work = t, work


def plan_group_noop(device):
    """One collective on the plan group that carries nothing (MAX of a zero count, the shape a StepPlan exchanges): issued by a
    rank that has no plan to drop in a step every rank repeats, so that the ranks' plan-group sequences stay paired
    (Trainer.step).  Returns a holder whose ``_gmax_work`` the caller waits for."""
    if not active():
        return None
    t = torch.zeros(1, dtype=torch.int64, device=device)
    return _NoopPlan(t, dist.all_reduce(t, op=dist.ReduceOp.MAX, group=plan_group(), async_op=True))


def frame_shard(num_frames: int, rank_: int | None = None, world: int | None = None):
    """Contiguous block [lo, hi) of first-frame indices a rank samples pairs (i, i+1) from.  The blocks
    partition [0, num_frames-1) — every adjacent pair belongs to exactly one rank."""
    r = rank() if rank_ is None else rank_
    w = world_size() if world is None else world
    pairs = num_frames - 1
    if w > pairs:
        raise ValueError(f"frame_shard: {w} ranks but only {pairs} adjacent-frame pairs in a {num_frames}-frame video; "
                         f"use at most {pairs} ranks (every rank needs at least one pair of its own)")
    base, extra = divmod(pairs, w)
    lo = r * base + min(r, extra)
    hi = lo + base + (1 if r < extra else 0)
    return lo, hi


def allreduce_gradients(params, average: bool = True):
    """Sum (or mean) the .grad of every parameter that has one, across ranks, with ONE collective on a flat
    bucket.  Every rank must hold gradients for the same parameters (true for a given GenerateMode)."""
    w = world_size()
    grads = [p.grad for p in params if p.grad is not None]
    if not active() or not grads:
        return 0
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat.div_(w)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
    return flat.numel()


class GradReducer:
    """Gradient all-reduce that overlaps the backward pass: every large parameter's collective is launched from a
    post-accumulate-grad hook once its gradient is final, small parameters (the MLP weights) are reduced in one flat bucket at
    the end.  Collectives must be issued in the same order on every rank.  The hooks fire in the order the ranks' backward
    graphs produce the gradients — the same graph everywhere in a given mode, but the fused / layer-by-layer MLP paths are chosen
    by the number of visible rows, which differs from rank to rank — so the launch order is FIXED instead: the first step
    launches nothing from its hooks (everything goes out in ``finish``, in parameter-list order) and adopts rank 0's observed
    order of completion; afterwards a parameter's collective goes out when it AND every parameter before it in that order
    are ready (what DistributedDataParallel does with its buckets).  On one rank it does nothing."""

    SMALL = 1 << 18     # elements; below this a tensor joins the flat bucket

    def __init__(self, average: bool = True):
        self.average = average
        self.enabled = True         # False: no exchange at all (bench.py measures the step without it; replicas diverge)
        self._hooked = {}           # id(param) -> (param, handle of the hook)
        self._pending = []          # (work, grad) of the collectives in flight
        self._params = []
        self._armed = False
        self._order = None          # agreed launch order: indices into the list of hooked parameters (in parameter-list order)
        self._order_key = None      # (number, sizes) of the hooked parameters the order was agreed for
        self._hook_list = []        # this step's hooked parameters in parameter-list order
        self._ready = {}            # index in _hook_list -> parameter whose gradient is final
        self._seen = []             # indices in the order their hooks fired (this step)
        self._next = 0
        self._sparse = None         # (idx [cap] padded with 0, n, cap, {id(param)}): rows the per-anchor gradients are non-zero in
        self._sparse_idx_all = None
        self.bytes_sent = 0         # gradient payload this rank handed to collectives in the last step (bench.py reports it)

    def set_sparse(self, idx, cap, params):
        """This step's per-anchor gradients (``params``) are non-zero only in rows ``idx`` (the distinct visible anchors of the
        rank's views, sorted; ``cap`` = the largest such count over the ranks, agreed with the step plan): they are exchanged as
        (row index, row) lists by all-gather + local scatter-add instead of a dense all-reduce — 53 k of 245 k rows in the
        BASELINE configs[2] step.  ``idx=None``: dense.  Every rank must make the same choice in the same step (it follows from
        the step plan, whose availability does not depend on data)."""
        self._sparse_idx_all = None
        if idx is None or not active():
            self._sparse = None
            return
        n = int(idx.shape[0])
        pad = torch.zeros(cap, dtype=torch.int64, device=idx.device)
        pad[:n] = idx
        self._sparse = (pad, n, int(cap), {id(p) for p in params})

    def arm(self, params, phase=None):
        """Call before backward with the step's parameters (new Parameter objects, e.g. after densification, get hooks;
        hooks of parameters that are gone are dropped).  ``phase``: anything that changes WHICH parameters receive a gradient
        (the generation mode: TRAININ_STE_ENTROPY renders from detached attributes, so _scaling / _offset / _anchor_feat get
        none) — part of the order's key, so that a new phase agrees on a new order in its first step; an order carried over
        from a phase where such a parameter came early would stall every launch behind a hook that never fires."""
        self._params = [p for p in params if p.requires_grad]
        if not active() or not self.enabled:
            self._armed = False
            return
        live = {id(p) for p in self._params}
        for k in [k for k in self._hooked if k not in live]:
            self._hooked.pop(k)[1].remove()
        for p in self._params:
            if id(p) not in self._hooked and p.numel() >= self.SMALL:
                self._hooked[id(p)] = (p, p.register_post_accumulate_grad_hook(self._on_grad))
        self._hook_list = [p for p in self._params if id(p) in self._hooked]
        key = (tuple(p.numel() for p in self._hook_list), phase)
        if key != self._order_key:          # another set of large parameters (densification changes their sizes) or phase: agree again
            self._order, self._order_key = None, key
        self._index = {id(p): i for i, p in enumerate(self._hook_list)}
        self._pending, self._ready, self._seen, self._next = [], {}, [], 0
        self._sparse_pending = []
        self.bytes_sent = 0
        self._armed = True

    def _launch(self, p):
        if self._sparse is not None and id(p) in self._sparse[3]:
            self._launch_sparse(p)
        else:
            self._pending.append((dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, async_op=True), p.grad))
            self.bytes_sent += 4 * p.numel()

    def _launch_sparse(self, p):
        idx, n, cap, _ = self._sparse
        W = world_size()
        if self._sparse_idx_all is None:          # the ranks' index lists: gathered once per step, shared by the per-anchor tensors
            self._sparse_idx_all = [torch.empty_like(idx) for _ in range(W)]
            self._sparse_idx_work = dist.all_gather(self._sparse_idx_all, idx, async_op=True)
            self.bytes_sent += 8 * cap
        A = p.shape[0]
        g2 = p.grad.reshape(A, -1)
        rows = torch.zeros(cap, g2.shape[1], device=p.device, dtype=p.dtype)
        rows[:n] = g2.index_select(0, idx[:n])
        got = [torch.empty_like(rows) for _ in range(W)]
        work = dist.all_gather(got, rows, async_op=True)
        self._sparse_pending.append((work, p, got))
        self.bytes_sent += 4 * rows.numel()

    def complete(self, params):
        """Wait for the collectives of these parameters NOW (they must have been launched: returns False otherwise and touches
        nothing), leave their gradients summed AND averaged, and take them out of what ``finish`` handles.  The early tail of a
        data-parallel step (Trainer._early_tail) updates two tensors from inside the backward."""
        want = {id(p) for p in params}
        dense = [(w, g) for w, g in self._pending if any(g is p.grad for p in params)]
        sparse = [e for e in self._sparse_pending if id(e[1]) in want]
        if len(dense) + len(sparse) != len(params):
            return False
        for w, g in dense:
            w.wait()
            if self.average:
                g.div_(float(world_size()))
        self._pending = [e for e in self._pending if not any(e[1] is g for _, g in dense)]
        for e in sparse:
            self._finish_sparse(e)
            if self.average:
                e[1].grad.div_(float(world_size()))
        self._sparse_pending = [e for e in self._sparse_pending if id(e[1]) not in want]
        return True

    def _finish_sparse(self, entry):
        work, p, got = entry
        self._sparse_idx_work.wait()
        work.wait()
        # the SAME fp32 summation order on every rank (((0 + g_0) + g_1) + ... + g_{W-1}): this rank's own rows are cleared and
        # come back through its gathered list like everybody else's.  Keeping them in place and adding the others in rank order
        # gave rank 2 of three (g2 + g0) + g1 where ranks 0 and 1 had (g0 + g1) + g2: replicas one ulp apart, which voxel
        # rounding in adjust_anchor can turn into different anchor counts.  Padding rows are zeros added to row 0.
        idx, n, _, _ = self._sparse
        g2 = p.grad.reshape(p.shape[0], -1)
        g2.index_fill_(0, idx[:n], 0.0)
        for r, rows in enumerate(got):
            g2.index_add_(0, self._sparse_idx_all[r], rows)

    def _launch_ready(self):
        while self._order is not None and self._next < len(self._order) and self._order[self._next] in self._ready:
            self._launch(self._ready[self._order[self._next]])
            self._next += 1

    def _on_grad(self, p):
        if self._armed and p.grad is not None and id(p) in self._index:
            i = self._index[id(p)]
            self._ready[i] = p
            self._seen.append(i)
            self._launch_ready()

    def finish(self):
        """Call after backward: launches what the hooks could not (first step: everything), reduces the small parameters, waits for
        everything, averages.  Returns the number of gradient elements reduced."""
        w = world_size()
        if not active() or not self.enabled:
            return 0
        self._armed = False
        # large parameters not launched from their hooks, in the agreed order (or parameter-list order while there is none);
        # one whose hook did not fire (hooks armed late) still takes part if it has a gradient
        for i, p in enumerate(self._hook_list):
            if i not in self._ready and p.grad is not None:
                self._ready[i] = p
        rest = self._order[self._next:] if self._order is not None else range(len(self._hook_list))
        for i in rest:
            if i in self._ready:
                self._launch(self._ready[i])
        done = {id(g) for _, g in self._pending} | {id(p.grad) for _, p, _ in self._sparse_pending}
        small = [p.grad for p in self._params if p.grad is not None and id(p.grad) not in done]
        n = sum(g.numel() for _, g in self._pending)
        if small:
            flat = torch.cat([g.reshape(-1) for g in small])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            off = 0
            for g in small:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
            n += off
        if self._order is None and self._hook_list:
            # agree on rank 0's order of completion (parameters whose hook did not fire go last, in list order)
            mine = self._seen + [i for i in range(len(self._hook_list)) if i not in self._seen]
            t = torch.tensor(mine, dtype=torch.int64, device=small[0].device if small else self._hook_list[0].device)
            dist.broadcast(t, src=0)
            self._order = [int(v) for v in t.tolist()]
        for work, _ in self._pending:
            work.wait()
        for e in self._sparse_pending:
            self._finish_sparse(e)
        if self.average:
            torch._foreach_div_([g for _, g in self._pending] + [p.grad for _, p, _ in self._sparse_pending] + small, float(w))
        self.bytes_sent += 4 * sum(g.numel() for g in small)
        self._pending, self._sparse_pending = [], []
        return n


def sparse_rows_pay(world: int, anchors: int, cap: int) -> bool:
    """The row-sparse exchange of the per-anchor gradients (GradReducer.set_sparse) against the dense all-reduce: every rank
    receives the other W - 1 ranks' row lists, each padded to ``cap`` = the largest distinct-visible-anchor count over the ranks,
    while a ring all-reduce of the dense tensors moves 2 (W - 1) / W of the ``anchors`` rows per rank.  Rows pay while
    (W - 1) cap < 2 (W - 1) A / W, i.e. cap < 2 A / W: always at two ranks, at eight only when a rank sees less than a quarter
    of the anchors.  A pure function of numbers every rank holds identically (cap is their agreed maximum)."""
    return world > 1 and (world - 1) * int(cap) < 2 * (world - 1) * int(anchors) // world


def adjust_anchor_replicated(pc, iteration: int, **kw):
    """``pc.adjust_anchor(**kw)`` with every rank taking the same decisions: the densification statistics are summed over the
    ranks first (each rank only saw its own frames), the random thinning of anchor_growing draws from a per-iteration seed, and
    afterwards only rank 0 carries the surviving accumulator rows into the next interval — the others restart from zero, so that
    the next sum is (old global + every rank's new observations) and not world_size copies of the survivors."""
    if not active():
        return pc.adjust_anchor(**kw)
    allreduce_statistics(pc)
    devices = [pc.device] if pc.device.type == "cuda" else []
    with torch.random.fork_rng(devices=devices):
        torch.manual_seed(977 + iteration)
        pc.adjust_anchor(**kw)
    keep_statistics_on_rank0(pc)


def allreduce_statistics(pc):
    """Sum the densification accumulators across ranks (each rank only sees its own frames)."""
    if not active():
        return
    for name in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"):
        t = getattr(pc, name, None)
        if isinstance(t, torch.Tensor) and t.numel():
            dist.all_reduce(t, op=dist.ReduceOp.SUM)


def keep_statistics_on_rank0(pc):
    """After an all-reduced adjust_anchor: ranks other than 0 zero their densification accumulators (see Trainer._adjust_anchor)."""
    if not active() or rank() == 0:
        return
    for name in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"):
        t = getattr(pc, name, None)
        if isinstance(t, torch.Tensor) and t.numel():
            t.zero_()


def any_rank(flag: bool, device) -> bool:
    """True on every rank if `flag` is True on at least one (a one-element MAX all-reduce; identity on one rank).
    Used for decisions all replicas must take together, e.g. repeating a step whose rasterizer buffer overflowed."""
    if not active():
        return bool(flag)
    t = torch.tensor([1.0 if flag else 0.0], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item() > 0)


def any_rank_start(flags_dev):
    """Non-blocking form of any_rank for flags that already live on the device (the rasterizers' overflow words): the
    MAX all-reduce and the copy of its result to the host are queued NOW — behind the kernels that produce the flags, not
    behind whatever is launched afterwards — and any_rank_finish() waits for that copy only.  None on a single rank."""
    if not active():
        return None
    t = torch.stack([f.reshape(-1)[0] for f in flags_dev]).max().to(torch.float32).reshape(1)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if not t.is_cuda:
        return (t, None, t)
    host = torch.empty(1, dtype=torch.float32, pin_memory=True)
    host.copy_(t, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return (host, ev, t)      # t: the reduced flag on the device (non-zero bits = some rank overflowed): a guard word for Adam


def any_rank_finish(handle, local_flag: bool) -> bool:
    if handle is None:
        return bool(local_flag)
    host, ev = handle[0], handle[1]
    if ev is not None:
        ev.synchronize()
    return bool(host.item() > 0) or bool(local_flag)


def broadcast_parameters(module, src: int = 0):
    """Make every rank start from rank `src`'s parameters and buffers."""
    if not active():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


# ------------------------------------------------------------------------------------------------ z-range ownership
# SURVEY.md section 8(e) "Collective": "reduce-scatter by z-sorted anchor range + all-gather of updated params (each rank owns the
# optimiser state of a z-range)".  Built as a HALO EXCHANGE, the shape xGMI's point-to-point links favour: the video's z axis is cut
# between the ranks' frame blocks, every anchor belongs to the rank whose block holds its z (anchor positions have learning rate 0:
# the owner of an existing anchor never changes), and a rank only ever READS the anchors inside its block widened by the slab
# half-width on both sides.  Per step and rank:
#   1. gradient rows of the halo — the anchors it can see but does not own — go to their owners (static row lists, one
#      all_to_all_single = grouped sends / receives between neighbours; no index lists travel, no counts are agreed);
#   2. the owner adds what arrived to its own rows in rank order (the same fp32 order as the replicated row exchange), averages,
#      adds the mask regulariser's closed form and runs Adam: owned rows hold the one true value of the model;
#   3. the owner sends the updated rows back along the same lists: every rank's block + halo is fresh before its next step.
# Rows outside a rank's block + halo go stale there; nothing of the step reads them (the slab test is exact on the anchor's z) except
# whole-tensor reductions, which become owner partial sums + one small all-reduce (param_means).  sync_full() makes every replica
# whole again: before densification (every update_interval steps), before evaluation / encoding / checkpoints.
# At configs[3] (100 k anchors, 8 ranks x 75 frames, slab +-48 frames) a rank sends 2 x 7.3 k halo rows of gradients and receives
# as many rows of parameters: 11 MB per step against 41 MB of row lists or 67 MB of ring all-reduce.
PER_ANCHOR = ("_anchor_feat", "_offset", "_scaling", "_mask")


def zrange_enabled() -> bool:
    """GSVC_DP_ZOWN=1 under data parallelism (off by default: RCCL has not measured it yet)."""
    return active() and os.environ.get("GSVC_DP_ZOWN") == "1"


def _all_to_all_rows(out, inp, out_splits, in_splits):
    """Rows to / from every peer by count (RCCL: one group of point-to-point sends and receives).  gloo has no CUDA all_to_all:
    the tests' shared-GPU ranks stage through the host."""
    if inp.is_cuda and dist.get_backend() == "gloo":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host, inp.cpu(), out_splits, in_splits)
        out.copy_(host)
    else:
        dist.all_to_all_single(out, inp, out_splits, in_splits)
    return out


class ZRangeOwnership:
    def __init__(self, num_frames: int, scale: float, threshold: float):
        self.W, self.r = world_size(), rank()
        T = int(num_frames)
        shards = [frame_shard(T, q, self.W) for q in range(self.W)]

        def z_of(t):          # reference frame_cube/frame.py:156-190: z = (id - T/2) / scale
            return (t - T / 2) / scale
        margin = 1e-4 * threshold + 1e-6          # the slab test runs in fp32 on view-space z: the read range errs on the wide side
        # rank q renders frames lo .. hi (pairs (i, i + 1), i in [lo, hi)): it can see anchors within the slab of any of them
        self.read = [(z_of(lo) - threshold - margin, z_of(hi) + threshold + margin) for lo, hi in shards]
        self.cuts = [z_of(shards[q][0] - 0.5) for q in range(1, self.W)]      # owner(z) = number of cuts <= z
        self._key = None
        self.bytes_sent = 0          # payload this rank handed to the two exchanges in the last step
        self.means = self.mask_sigmoid_mean = None

    def ensure(self, pc):
        a = pc._anchor
        key = (id(a), a.data_ptr(), int(a.shape[0]), a._version)
        if key != self._key:
            self.build(a)
            self._key = key

    def build(self, anchor):
        """Row lists from the anchors' z — identical arithmetic on identical replicas: every rank derives every list it shares
        with a peer by itself (rows ascending), nothing is exchanged."""
        W, r = self.W, self.r
        z = anchor.detach()[:, 2].float().contiguous()
        dev, A = z.device, int(z.shape[0])
        owner = (torch.bucketize(z, torch.tensor(self.cuts, dtype=z.dtype, device=dev), right=True) if self.cuts
                 else torch.zeros(A, dtype=torch.int64, device=dev))
        inr = [(z >= lo) & (z <= hi) for lo, hi in self.read]
        own = owner == r
        self.A, self.own_mask = A, own
        self.own_idx = torch.nonzero(own).squeeze(1)
        empty = torch.zeros(0, dtype=torch.int64, device=dev)
        self.send_idx = [torch.nonzero((owner == q) & inr[r]).squeeze(1) if q != r else empty for q in range(W)]
        self.recv_idx = [torch.nonzero(own & inr[q]).squeeze(1) if q != r else empty for q in range(W)]
        t = torch.zeros(A, dtype=torch.bool, device=dev)
        for idx in self.recv_idx:
            t[idx] = True
        self.touched = torch.nonzero(t).squeeze(1)          # owned rows some peer can see
        # one dummy row in the rank's own slot keeps every message non-empty (a one-rank RCCL communicator still runs the call)
        dummy = torch.zeros(1, dtype=torch.int64, device=dev)
        self.in_splits = [int(self.send_idx[q].shape[0]) if q != r else 1 for q in range(W)]
        self.out_splits = [int(self.recv_idx[q].shape[0]) if q != r else 1 for q in range(W)]
        self.send_all = torch.cat([self.send_idx[q] if q != r else dummy for q in range(W)])
        self.recv_all = torch.cat([self.recv_idx[q] if q != r else dummy for q in range(W)])

    def halo_rows(self):
        """(rows sent as gradients / received as parameters, rows received as gradients / sent as parameters) per step."""
        return sum(self.in_splits) - 1, sum(self.out_splits) - 1

    @staticmethod
    def _tensors(pc, names):
        ts = [getattr(pc, n) for n in names]
        assert all(p.is_contiguous() for p in ts)
        return ts

    def exchange_grads(self, pc, names=PER_ANCHOR):
        """Step 1 + 2 above.  Afterwards the OWNED rows of every ``names`` gradient hold the mean over the ranks; other rows hold
        local leftovers that the parameter refresh makes irrelevant."""
        self.ensure(pc)
        W, r, A = self.W, self.r, self.A
        ts = self._tensors(pc, names)
        for p in ts:
            if p.grad is None:          # a rank whose views touched nothing still takes part
                p.grad = torch.zeros_like(p)
            elif not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()
        g2 = [p.grad.view(A, -1) for p in ts]
        widths = [int(g.shape[1]) for g in g2]
        if os.environ.get("GSVC_DP_ZOWN_CHECK"):          # diagnostics: a gradient outside block + halo would be dropped silently
            seen = torch.zeros(A, dtype=torch.bool, device=g2[0].device)
            seen[self.own_idx] = True
            seen[self.send_all] = True
            for n, g in zip(names, g2):
                bad = (g != 0).any(dim=1) & ~seen
                if bool(bad.any()):
                    raise RuntimeError(f"ZRangeOwnership: {int(bad.sum())} rows of {n}.grad lie outside this rank's block + halo")
        inp = torch.cat([g.index_select(0, self.send_all) for g in g2], dim=1)
        out = torch.empty(int(self.recv_all.shape[0]), sum(widths), dtype=inp.dtype, device=inp.device)
        _all_to_all_rows(out, inp, self.out_splits, self.in_splits)
        self.bytes_sent = 4 * (inp.shape[0] - 1) * inp.shape[1]
        U, c0 = self.touched, 0
        for g, w in zip(g2, widths):
            if U.numel():
                # ((0 + g_0) + g_1) + ... + g_{W-1}: the order GradReducer._finish_sparse uses, this rank's own rows in its place
                mine = g.index_select(0, U)
                g.index_fill_(0, U, 0.0)
                off = 0
                for q in range(W):
                    n = self.out_splits[q]
                    if q == r:
                        g.index_add_(0, U, mine)
                    elif n:
                        g.index_add_(0, self.recv_idx[q], out[off:off + n, c0:c0 + w])
                    off += n
            c0 += w
            g.div_(float(W))

    def refresh_params(self, pc, names=PER_ANCHOR):
        """Step 3: the owners' updated rows travel back along the same lists."""
        self.ensure(pc)
        W, r, A = self.W, self.r, self.A
        ts = self._tensors(pc, names)
        with torch.no_grad():
            p2 = [p.view(A, -1) for p in ts]
            widths = [int(p.shape[1]) for p in p2]
            inp = torch.cat([p.index_select(0, self.recv_all) for p in p2], dim=1)
            out = torch.empty(int(self.send_all.shape[0]), sum(widths), dtype=inp.dtype, device=inp.device)
            _all_to_all_rows(out, inp, self.in_splits, self.out_splits)
            self.bytes_sent += 4 * (inp.shape[0] - 1) * inp.shape[1]
            c0 = 0
            for p, w in zip(p2, widths):
                off = 0
                for q in range(W):
                    n = self.in_splits[q]
                    if q != r and n:
                        p.index_copy_(0, self.send_idx[q], out[off:off + n, c0:c0 + w])
                    off += n
                c0 += w

    def sync_full(self, pc, moments: bool = True):
        """Every replica whole again: each row from its owner (zeros elsewhere, summed: x + 0 + ... + 0 is exact).  With
        ``moments`` the Adam moments too (a checkpoint must hold the owners')."""
        self.ensure(pc)
        keep = self.own_mask.unsqueeze(1)
        with torch.no_grad():
            for n in PER_ANCHOR:
                p = getattr(pc, n)
                tensors = [p]
                st = pc.optimizer.state.get(p, {}) if (moments and getattr(pc, "optimizer", None) is not None) else {}
                tensors += [st[k] for k in ("exp_avg", "exp_avg_sq") if isinstance(st.get(k), torch.Tensor)]
                for t in tensors:
                    v = t.view(self.A, -1)
                    v.masked_fill_(~keep, 0.0)
                    dist.all_reduce(v, op=dist.ReduceOp.SUM)

    def update_means(self, pc):
        """[mean(_anchor_feat), mean(get_scaling), mean(_offset)] — the clamp centres of the rate model and of the straight-through
        quantiser (gsvc_amd.generate._param_means) — from the owners' partial sums: a replica's own whole-tensor mean would read
        stale rows.  Once per step, before the forward (every rank the same number of collectives)."""
        self.ensure(pc)
        with torch.no_grad():
            own = self.own_idx
            parts = [pc._anchor_feat.detach().index_select(0, own), pc.get_scaling.detach().index_select(0, own),
                     pc._offset.detach().index_select(0, own)]
            # (the fourth partial: sigmoid(_mask) of the owned rows — the VALUE of the mask regulariser 5e-4 * mean(sigmoid(_mask)) over
            # ALL anchors, which a replica would otherwise take over its partly stale rows: the reported loss is then exact)
            s = torch.stack([t.sum(dtype=torch.float32) for t in parts] +
                            [torch.sigmoid(pc._mask.detach().index_select(0, own)).sum(dtype=torch.float32)])
            dist.all_reduce(s, op=dist.ReduceOp.SUM)
            n = torch.tensor([float(pc._anchor_feat.numel()), float(pc.get_scaling.numel()), float(pc._offset.numel()),
                              float(pc._mask.numel())], device=s.device)
            m = s / n
            self.means, self.mask_sigmoid_mean = m[:3], m[3]
        return self.means
