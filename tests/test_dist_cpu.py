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


def _sharded_worker(rank, world, port, q):
    """ShardedAnchorAdam (reduce-scatter over anchor ranges + sharded Adam + all-gather) against a single-process torch Adam
    on the rank-averaged gradients: same parameters on every rank after every step, same moments when gathered."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from gsvc_amd import dist as gd
    gd.init_from_env("gloo")
    A = 7                                             # not a multiple of the world size: the last range is short
    shapes = {"offset": (A, 2, 3), "mask": (A, 2, 1), "anchor_feat": (A, 5), "scaling": (A, 6), "mlp": (4, 3)}
    lrs = {"offset": 0.01, "mask": 0.02, "anchor_feat": 0.0075, "scaling": 0.007, "mlp": 0.005}

    def build():
        torch.manual_seed(1)
        ps = {k: torch.nn.Parameter(torch.randn(*sh)) for k, sh in shapes.items()}
        opt = torch.optim.Adam([{"params": [ps[k]], "lr": lrs[k], "name": k} for k in shapes], lr=0.0, eps=1e-15)
        return ps, opt

    def coeff(k, r, step):                            # d loss / d p on rank r at `step`
        g = torch.Generator().manual_seed(100 * step + 10 * r + len(k))
        return torch.randn(*shapes[k], generator=g)

    ps, opt = build()
    ref_ps, ref_opt = build()
    sh = gd.ShardedAnchorAdam(opt)
    red = gd.GradReducer(sharded=sh)
    assert sorted(n for n in shapes if sh.owns(ps[n])) == ["anchor_feat", "mask", "offset", "scaling"]
    for step in range(1, 5):
        red.arm(list(ps.values()))
        sum((ps[k] * coeff(k, rank, step)).sum() for k in shapes).backward()
        red.finish()
        assert torch.allclose(ps["mlp"].grad, sum(coeff("mlp", r, step) for r in range(world)) / world, atol=1e-6)
        if step == 3:                                 # state round trip through the wrapped optimizer (what anchor growing edits)
            sh.gather_state()
            for k in ("offset", "scaling"):
                assert torch.allclose(opt.state[ps[k]]["exp_avg"], ref_opt.state[ref_ps[k]]["exp_avg"], atol=1e-7)
                assert torch.allclose(opt.state[ps[k]]["exp_avg_sq"], ref_opt.state[ref_ps[k]]["exp_avg_sq"], atol=1e-9)
            sh.adopt_state()
            assert ps["offset"] not in opt.state
        sh.step()
        assert all(ps[k].grad is None for k in ("offset", "mask", "anchor_feat", "scaling"))
        opt.step()
        opt.zero_grad(set_to_none=True)
        for k in shapes:
            ref_ps[k].grad = sum(coeff(k, r, step) for r in range(world)) / world
        ref_opt.step()
        for k in shapes:
            assert torch.allclose(ps[k], ref_ps[k], atol=2e-7), (step, k, (ps[k] - ref_ps[k]).abs().max().item())
            other = [torch.zeros_like(ps[k]) for _ in range(world)]
            dist.all_gather(other, ps[k].data)
            assert torch.equal(other[0], other[1]), (step, k)       # replicas stay identical
    # a step whose parameters were replaced after the backward completes its collectives and updates nothing
    before = {k: ps[k].detach().clone() for k in shapes}
    red.arm(list(ps.values()))
    sum((ps[k] * coeff(k, rank, 9)).sum() for k in shapes).backward()
    red.finish()
    sh.step(skip_update=True)
    assert all(torch.equal(ps[k], before[k]) for k in ("offset", "mask", "anchor_feat", "scaling"))
    # ... which is also what Trainer does with a step whose rasterizer buffers overflowed (its gradients are invalid) before it
    # repeats the step: the repeat's hooks must start FRESH reduce-scatters, and its update must come from the repeat's
    # gradients alone (a collective left in flight made start() skip the parameter and Adam run on the stale shards)
    opt.zero_grad(set_to_none=True)
    red.arm(list(ps.values()))
    sum((ps[k] * coeff(k, rank, 10)).sum() for k in shapes).backward()
    red.finish()
    sh.step()
    opt.step()
    opt.zero_grad(set_to_none=True)
    for k in shapes:
        ref_ps[k].grad = sum(coeff(k, r, 10) for r in range(world)) / world
    ref_opt.step()
    for k in shapes:
        assert torch.allclose(ps[k], ref_ps[k], atol=2e-7), ("after the dropped step", k, (ps[k] - ref_ps[k]).abs().max().item())
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def test_sharded_anchor_adam_two_rank_gloo():
    world, port = 2, 29500 + (os.getpid() + 7) % 400
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5)[0] for _ in range(world)) == [0, 1]


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
