"""world_size-2 gloo tests of the multi-GPU layer (gsvc_amd/dist.py): frame sharding, the one-bucket
gradient all-reduce and the densification-statistics reduction.  Runs on CPU."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from gsvc_amd import dist as gd
    r, w, _ = gd.init_from_env("gloo")
    assert (r, w) == (rank, world) and gd.rank() == rank and gd.world_size() == world
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    extra = torch.nn.Parameter(torch.zeros(4))          # a parameter that gets no gradient on any rank
    if rank == 1:
        for p in model.parameters():
            p.data.add_(1.0)
    gd.broadcast_parameters(model, src=0)
    x = torch.full((2, 5), float(rank + 1))
    model(x).sum().backward()
    local = [p.grad.clone() for p in model.parameters()]
    n = gd.allreduce_gradients(list(model.parameters()) + [extra])
    assert n == sum(p.numel() for p in model.parameters()) and extra.grad is None
    gathered = [[torch.zeros_like(g) for _ in range(world)] for g in local]
    for g, out in zip(local, gathered):
        dist.all_gather(out, g)
    for p, out in zip(model.parameters(), gathered):
        assert torch.allclose(p.grad, sum(out) / world, atol=1e-6)
    pc = type("PC", (), {})()
    pc.opacity_accum = torch.full((3, 1), float(rank + 1))
    pc.anchor_demon = torch.ones(3, 1)
    pc.offset_gradient_accum = torch.full((6, 1), 0.5 * (rank + 1))
    pc.offset_denom = torch.ones(6, 1)
    gd.allreduce_statistics(pc)
    assert torch.all(pc.opacity_accum == 3.0) and torch.all(pc.anchor_demon == 2.0) and torch.all(pc.offset_gradient_accum == 1.5)
    # two densification intervals: what the ranks hold after the second reduction must equal a single process that saw every
    # rank's observations — not world_size copies of the rows that survived the first adjust_anchor
    def observe(step):        # this rank's new observations of interval `step`
        return torch.arange(1.0, 4.0).view(3, 1) * (rank + 1) * step
    def adjust(t):            # stand-in for adjust_anchor: rows above the threshold are consumed (zeroed), the rest survive
        t[t > 5.0] = 0.0
    pc2 = type("PC", (), {})()
    for name in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"):
        setattr(pc2, name, torch.zeros(3, 1))
    single = torch.zeros(3, 1)
    for step in (1, 2):
        for name in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"):
            getattr(pc2, name).add_(observe(step))
        single += sum(torch.arange(1.0, 4.0).view(3, 1) * (r_ + 1) * step for r_ in range(world))
        gd.allreduce_statistics(pc2)
        assert torch.equal(pc2.opacity_accum, single), (step, pc2.opacity_accum.tolist(), single.tolist())
        adjust(single)
        for name in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"):
            adjust(getattr(pc2, name))
        gd.keep_statistics_on_rank0(pc2)
        assert torch.equal(pc2.offset_denom, single if rank == 0 else torch.zeros(3, 1))
    # overlapped reducer: large tensors from hooks during backward, small ones in a flat bucket; a second step re-arms it
    red = gd.GradReducer()
    red.SMALL = 16
    big = torch.nn.Parameter(torch.ones(8, 4))
    tiny = torch.nn.Parameter(torch.ones(3))
    unused = torch.nn.Parameter(torch.ones(2))
    for it in range(2):
        for p in (big, tiny, unused):
            p.grad = None
        red.arm([big, tiny, unused])
        ((big * float(rank + 1)).sum() + (tiny * float(10 * (rank + 1))).sum()).backward()
        n = red.finish()
        assert n == big.numel() + tiny.numel() and unused.grad is None
        assert torch.allclose(big.grad, torch.full_like(big, 1.5)) and torch.allclose(tiny.grad, torch.full_like(tiny, 15.0))
    big2 = torch.nn.Parameter(torch.ones(8, 4))          # a replaced parameter (densification) gets its own hook
    red.arm([big2, tiny])
    big2.grad, tiny.grad = None, None
    ((big2 * float(rank + 1)).sum() + tiny.sum()).backward()
    red.finish()
    assert torch.allclose(big2.grad, torch.full_like(big2, 1.5)) and len(red._hooked) == 1
    # the ranks' backward graphs may finish the large gradients in DIFFERENT orders (the fused / layer-by-layer MLP paths are
    # chosen by the number of visible rows): the collectives still go out in one agreed order (rank 0's of the first step)
    red2 = gd.GradReducer()
    red2.SMALL = 16
    pa, pb, pc3 = (torch.nn.Parameter(torch.ones(8, 4) * k) for k in (1.0, 2.0, 3.0))
    for it in range(3):
        for p in (pa, pb, pc3):
            p.grad = None
        red2.arm([pa, pb, pc3])
        x = torch.ones(1, requires_grad=True)
        # rank 0's graph completes pa, pb, pc3 in that order, rank 1's in the opposite order (later-created nodes run first)
        chain = (pc3, pb, pa) if rank == 0 else (pa, pb, pc3)
        loss = x.sum() * 0
        for k, p in enumerate(chain):
            loss = loss + (p * float(rank + 1 + k)).sum() * (x * 0 + 1).sum()
        loss.backward()
        red2.finish()
        fired = [("a", "b", "c")[i] for i in red2._seen]
        assert fired == (["a", "b", "c"] if rank == 0 else ["c", "b", "a"]), fired
        assert red2._order == [0, 1, 2]                    # rank 0's order, on both ranks
        want = {id(pa): 0.5 * ((1 + 2) + (2 + 0)), id(pb): 0.5 * ((1 + 1) + (2 + 1)), id(pc3): 0.5 * ((1 + 0) + (2 + 2))}
        for p in (pa, pb, pc3):
            assert torch.allclose(p.grad, torch.full_like(p, want[id(p)])), (rank, it)
    # collective yes/no decisions: blocking form and the early (start / finish) form
    assert gd.any_rank(rank == 1, torch.device("cpu")) is True and gd.any_rank(False, torch.device("cpu")) is False
    h = gd.any_rank_start([torch.tensor([0, 0]), torch.tensor([1 if rank == 0 else 0, 7])])
    assert gd.any_rank_finish(h, False) is True
    h = gd.any_rank_start([torch.tensor([0]), torch.tensor([0])])
    assert gd.any_rank_finish(h, False) is False and gd.any_rank_finish(h, True) is True
    q.put((rank, gd.frame_shard(600), [float(p.data.sum()) for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, w0), (r1, s1, w1) = res
    assert s0 == (0, 300) and s1 == (300, 599)       # blocks partition the 599 adjacent pairs
    assert w0 == w1                                   # broadcast made the replicas identical


def test_frame_shard_partitions():
    from gsvc_amd.dist import frame_shard
    with pytest.raises(ValueError, match="adjacent-frame pairs"):
        frame_shard(4, 0, 8)              # more ranks than pairs: refused instead of handing out another rank's frames
    for T, W in ((600, 8), (300, 8), (17, 4), (9, 8), (2, 1)):
        blocks = [frame_shard(T, r, W) for r in range(W)]
        assert blocks[0][0] == 0 and blocks[-1][1] == T - 1
        for a, b in zip(blocks, blocks[1:]):
            assert a[1] == b[0]
        sizes = [hi - lo for lo, hi in blocks]
        assert max(sizes) - min(sizes) <= 1
    from gsvc_amd import dist as gd
    assert gd.world_size() == 1 and gd.rank() == 0 and gd.allreduce_gradients([]) == 0


def _many_rank_worker(rank, world, port, q):
    """What two ranks cannot show (a + b is commutative): one summation order on every rank of the row-sparse exchange, the
    agreed collective order with every rank finishing its gradients in another order, the sparse / dense decision on both sides
    of cap = 2 A / W, and densification from unevenly sharded statistics."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import numpy as np
    from gsvc_amd import dist as gd
    gd.init_from_env("gloo")
    W = world

    def same_on_all_ranks(t, what):
        got = [torch.zeros_like(t) for _ in range(W)]
        dist.all_gather(got, t.contiguous())
        for r in range(1, W):
            assert torch.equal(got[0], got[r]), (what, "rank", r, float((got[0] - got[r]).abs().max()))
        return got

    # ---- 1. row-sparse exchange: mean of the ranks' gradients, the SAME BITS on every rank, replicas stay identical under Adam
    A = 257
    shapes = {"offset": (A, 2, 3), "mask": (A, 2, 1), "anchor_feat": (A, 5), "scaling": (A, 6)}
    torch.manual_seed(5)
    ps = {k: torch.nn.Parameter(torch.randn(*sh)) for k, sh in shapes.items()}
    mlp = torch.nn.Parameter(torch.randn(4, 3))
    opt = torch.optim.Adam([{"params": [p], "lr": 0.01} for p in list(ps.values()) + [mlp]], eps=1e-15)
    red = gd.GradReducer()
    red.SMALL = 16
    for step in range(1, 4):
        g = torch.Generator().manual_seed(1000 * step + rank)
        n = int(torch.randint(20, 120, (1,), generator=g))                 # this rank's distinct visible anchors: its own count
        idx = torch.randperm(A, generator=g)[:n].sort().values
        capt = torch.tensor([n])
        dist.all_reduce(capt, op=dist.ReduceOp.MAX)
        cap = int(capt)
        coef = {k: torch.zeros(*sh) for k, sh in shapes.items()}
        for k in shapes:                                                   # values spread over many binades: fp32 sums depend on the order
            v = torch.randn(n, *shapes[k][1:], generator=g) * torch.exp2(torch.randint(-12, 12, (n,), generator=g).float()).view(
                n, *([1] * (len(shapes[k]) - 1)))
            coef[k][idx] = v
        cm = torch.randn(4, 3, generator=g)
        opt.zero_grad(set_to_none=True)
        red.arm(list(ps.values()) + [mlp])
        red.set_sparse(idx, cap, list(ps.values()))
        (sum((ps[k] * coef[k]).sum() for k in shapes) + (mlp * cm).sum()).backward()
        red.finish()
        assert red._sparse is not None
        for k in shapes:
            locals_ = [torch.zeros_like(coef[k]) for _ in range(W)]
            dist.all_gather(locals_, coef[k])
            want = torch.stack(locals_).double().sum(0) / W
            scale = float(want.abs().max())
            assert float((ps[k].grad.double() - want).abs().max()) <= 1e-6 * scale, (step, k)
            same_on_all_ranks(ps[k].grad, f"sparse grad {k} step {step}")
        opt.step()
        for k in shapes:
            same_on_all_ranks(ps[k].data, f"parameter {k} after step {step}")
        same_on_all_ranks(mlp.data, "mlp")

    # ---- 2. one agreed launch order although every rank completes its gradients in its own order
    red2 = gd.GradReducer()
    red2.SMALL = 16
    big = [torch.nn.Parameter(torch.ones(8, 4) * (k + 1)) for k in range(5)]
    for it in range(3):
        for p in big:
            p.grad = None
        red2.arm(big)
        x = torch.ones(1, requires_grad=True)
        perm = torch.randperm(5, generator=torch.Generator().manual_seed(rank)).tolist()     # later-created nodes run first
        loss = x.sum() * 0
        for k in perm:
            loss = loss + (big[k] * float(rank + 1 + k)).sum() * (x * 0 + 1).sum()
        loss.backward()
        red2.finish()
        assert red2._seen == perm[::-1], (rank, red2._seen, perm)
        order = torch.tensor(red2._order)
        same_on_all_ranks(order, "agreed order")
        for k, p in enumerate(big):
            want = sum(r + 1 + k for r in range(W)) / W
            assert torch.allclose(p.grad, torch.full_like(p, want)), (rank, it, k)

    # ---- 3. rows or dense: the same answer on every rank, on both sides of cap = 2 A / W
    A3 = 100_800                                                          # divisible by 3 and by 8: the edge is an integer
    edge = 2 * A3 // W
    mine = torch.tensor([edge - 1 - 17 * rank])                            # every rank sees another count; the largest decides
    for shift, want in ((0, True), (1, False), (5000, False), (-5000, True)):
        t = mine + shift
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ans = gd.sparse_rows_pay(W, A3, int(t))
        assert ans is want, (W, int(t), edge, ans)
        same_on_all_ranks(torch.tensor([int(ans)]), "sparse / dense decision")
    assert gd.sparse_rows_pay(1, A3, 1) is False and gd.sparse_rows_pay(2, A3, A3 - 1) is True

    # ---- 4. densification from statistics that live unevenly on the ranks: same anchors everywhere, and the anchors a single
    # process gets from the summed statistics under the same seed
    from test_densify_cpu import T, _model
    g = np.load(os.path.join(ROOT, "tests", "golden", "densify.npz"))
    kw = dict(check_interval=100, success_threshold=0.8, grad_threshold=0.0005, min_opacity=0.005)

    def build():
        pc = _model(g)
        pc.update_learning_rate(2000)
        for nm in ("_offset", "_mask", "_anchor_feat", "_scaling"):
            getattr(pc, nm).grad = T(g["grad::" + nm])
        pc.optimizer.step()
        return pc
    pc = build()
    full = {nm: T(g["stat::" + nm]) for nm in ("offset_denom", "offset_gradient_accum", "anchor_demon", "opacity_accum")}
    for nm, t in full.items():
        rows = t.shape[0]
        cuts = [0] + sorted(torch.randperm(rows - 1, generator=torch.Generator().manual_seed(3))[:W - 1].add(1).tolist()) + [rows]
        part = torch.zeros_like(t)                                         # rank r observed rows [cuts[r], cuts[r + 1]): uneven blocks
        part[cuts[rank]:cuts[rank + 1]] = t[cuts[rank]:cuts[rank + 1]]
        setattr(pc, nm, part)
    gd.adjust_anchor_replicated(pc, 700, **kw)
    assert pc._anchor.shape[0] != g["in::_anchor"].shape[0]
    for nm in ("_anchor", "_offset", "_mask", "_anchor_feat", "_scaling"):
        same_on_all_ranks(getattr(pc, nm).data, "after adjust_anchor: " + nm)
    st = pc.optimizer.state[pc._anchor_feat]
    same_on_all_ranks(st["exp_avg_sq"], "Adam moments after adjust_anchor")
    single = build()
    for nm, t in full.items():
        setattr(single, nm, t.clone())
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(977 + 700)
        single.adjust_anchor(**kw)
    for nm in ("_anchor", "_offset", "_anchor_feat"):
        assert torch.equal(getattr(pc, nm).data, getattr(single, nm).data), nm
    for nm in full:        # survivors are carried by rank 0 alone: the next interval's sum counts them once
        mine_ = getattr(pc, nm)
        assert torch.equal(mine_, getattr(single, nm)) if rank == 0 else not mine_.any(), nm
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


@pytest.mark.parametrize("world", [3, 8])
def test_many_rank_gloo(world):
    port = 29500 + (os.getpid() + 31 * world) % 400
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_many_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5)[0] for _ in range(world)) == list(range(world))


# ------------------------------------------------------------------------------------------------ z-range ownership
class _FakeModel:
    """The per-anchor tensors of a GaussianModel (shapes scaled down) with a plain Adam: what gsvc_amd.dist.ZRangeOwnership touches."""

    def __init__(self, A, z_lim, seed=0):
        g = torch.Generator().manual_seed(seed)
        P = torch.nn.Parameter
        anchor = torch.rand(A, 3, generator=g) * 2 - 1
        anchor[:, 2] *= z_lim
        self._anchor = P(anchor, requires_grad=False)
        self._anchor_feat = P(torch.randn(A, 5, generator=g))
        self._offset = P(torch.randn(A, 2, 3, generator=g))
        self._scaling = P(torch.randn(A, 6, generator=g) * 0.1)
        self._mask = P(torch.randn(A, 2, 1, generator=g))
        self.optimizer = torch.optim.Adam([self._anchor_feat, self._offset, self._scaling, self._mask], lr=1e-2, eps=1e-15)

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)


def _zown_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      GSVC_DP_ZOWN="1")
    from gsvc_amd import dist as gd
    gd.init_from_env("gloo")
    assert gd.zrange_enabled()
    T, scale = 64, 32.0
    thr = (10.0 if world > 2 else 6.0) / scale          # eight ranks: blocks of 7-8 frames, a halo that reaches two neighbours a side
    A = 600
    names = gd.PER_ANCHOR
    own, ref = _FakeModel(A, 1.1 * T / 2 / scale), _FakeModel(A, 1.1 * T / 2 / scale)      # ref: the replicated form, every rank the whole sum
    zo = gd.ZRangeOwnership(T, scale, thr)
    zo.ensure(own)
    z = own._anchor.detach()[:, 2]
    lo, hi = gd.frame_shard(T)
    in_read = (z >= zo.read[rank][0]) & (z <= zo.read[rank][1])
    sent, got = zo.halo_rows()
    tot = torch.tensor([float(sent), float(got)])
    dist.all_reduce(tot)
    assert tot[0] == tot[1] and (world == 1 or tot[0] > 0)          # what the ranks send is what the owners receive
    owners = torch.zeros(A)
    owners[zo.own_idx] = 1.0
    dist.all_reduce(owners)
    assert torch.all(owners == 1.0)                                  # every anchor has exactly one owner
    rng = torch.Generator().manual_seed(100 + rank)
    for step in range(1, 7):
        f = int(torch.randint(lo, max(lo + 1, hi), (1,), generator=rng))
        zf = [(t - T / 2) / scale for t in (f, f + 1)]
        vis = ((z - zf[0]).abs() <= thr) | ((z - zf[1]).abs() <= thr)
        assert torch.all(in_read[vis])                               # a rank only sees rows of its block + halo
        vis &= torch.rand(A, generator=rng) < 0.7
        only_mask = step == 5                                        # the STE phase's shape: one tensor has a gradient
        use = ("_mask",) if only_mask else names
        mean_ref = torch.stack([ref._anchor_feat.mean(), ref.get_scaling.mean(), ref._offset.mean()]).detach()
        m = zo.update_means(own)
        assert torch.allclose(m, mean_ref, rtol=1e-5, atol=1e-7), (m, mean_ref)
        # the mask regulariser's VALUE over all anchors from the owners' partial sums (a replica's own mean would read stale rows)
        assert torch.allclose(zo.mask_sigmoid_mean, torch.sigmoid(ref._mask.detach()).mean(), rtol=1e-5, atol=1e-7)
        ms = [torch.zeros_like(m) for _ in range(world)]
        dist.all_gather(ms, m)
        assert all(torch.equal(ms[0], x) for x in ms)                # the same bits on every rank
        for n in names:
            getattr(own, n).grad = None
            getattr(ref, n).grad = None
        for n in use:
            p = getattr(own, n)
            g = torch.randn(p.shape, generator=rng) * vis.view(-1, *([1] * (p.dim() - 1)))
            if rank == 1 and step == 3 and n == "_offset":
                g = None                                              # a rank whose views touched nothing of this tensor
            p.grad = g
            mine = g if g is not None else torch.zeros_like(p)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            total = torch.zeros_like(mine)
            for t in every:                                           # ((0 + g_0) + g_1) + ...: GradReducer._finish_sparse's order
                total = total + t
            getattr(ref, n).grad = total / float(world)
        zo.exchange_grads(own, use)
        for n in use:
            assert torch.equal(getattr(own, n).grad[zo.own_idx], getattr(ref, n).grad[zo.own_idx]), (step, n)
        own.optimizer.step()
        ref.optimizer.step()
        zo.refresh_params(own, use)
        for n in names:                                               # block + halo: the owners' values, bit for bit
            assert torch.equal(getattr(own, n).data[in_read], getattr(ref, n).data[in_read]), (step, n)
    stale = sum(int((getattr(own, n).data != getattr(ref, n).data).any()) for n in names)
    if world > 2:
        assert stale > 0                                              # rows outside block + halo did go stale ...
    zo.sync_full(own)
    for n in names:                                                   # ... and come back whole, moments included
        assert torch.equal(getattr(own, n).data, getattr(ref, n).data), n
        so, sr = own.optimizer.state[getattr(own, n)], ref.optimizer.state[getattr(ref, n)]
        assert torch.equal(so["exp_avg"], sr["exp_avg"]) and torch.equal(so["exp_avg_sq"], sr["exp_avg_sq"]), n
    # densification replaces the anchor tensor: the lists follow (existing anchors keep their owner)
    old_owner = zo.own_mask.clone()
    own._anchor = torch.nn.Parameter(torch.cat([own._anchor.data, own._anchor.data[:50] * 0.5]), requires_grad=False)
    zo.ensure(own)
    assert zo.A == A + 50 and torch.equal(zo.own_mask[:A], old_owner)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_z_range_ownership_equals_the_replicated_exchange(world):
    """gsvc_amd.dist.ZRangeOwnership (SURVEY 8e "Collective"; GSVC_DP_ZOWN=1): gradients of the halo rows to their owners, Adam on
    the owned rows, updated rows back — against the replicated form in which every rank adds every rank's rows in rank order and
    updates everything: the owners' gradients, the parameters of a rank's block + halo after every step, and the whole model with
    its Adam moments after sync_full are the same bits; the clamp centres' means are one number on every rank."""
    port = 29500 + (os.getpid() + 57 * world) % 400
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_zown_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5)[0] for _ in range(world)) == list(range(world))
