"""Whole-network autograd functions (gsvc_amd/mlp.py: every MLP of reference scene/gaussian_model.py:150-232 as one
chain of GEMM launches with the activations on the epilogue) against a plain PyTorch fp32 statement of the same networks:
outputs and the gradient of every parameter and input.  fp32 products and sums on both sides; summation order and the
GEMM-side erf differ -> 2e-5 of the tensor's scale forward, 1e-3 of the gradient's scale backward (sums over ~6000 rows).
"""
import os

import pytest
import torch
import torch.nn.functional as F

from gsvc_amd import switches

pytestmark = pytest.mark.gpu


def _close(a, b, tol):
    scale = max(1e-6, b.abs().max().item())
    err = (a - b).abs().max().item()
    assert err <= tol * scale, (err, scale)


def _torch_generator(net, feat, cond):
    h = F.linear(F.gelu(F.linear(feat, net.linear1.weight, net.linear1.bias)), net.linear2.weight, net.linear2.bias)
    f = net.film
    gamma = F.linear(torch.relu(F.linear(cond, f.fc_gamma0.weight, f.fc_gamma0.bias)), f.fc_gamma1.weight, f.fc_gamma1.bias)
    beta = F.linear(torch.relu(F.linear(cond, f.fc_beta0.weight, f.fc_beta0.bias)), f.fc_beta1.weight, f.fc_beta1.bias)
    return net.out_act(F.linear(gamma * h + beta, net.out_linear.weight, net.out_linear.bias))


@pytest.mark.parametrize("out_dim,act", [(10, "tanh"), (70, None), (30, "sigmoid")])
@pytest.mark.parametrize("M", [6001, 4096])
def test_generator_chain_matches_torch(out_dim, act, M):
    """GeneratorNet at the production sizes (feature 50 -> 100 -> 100, condition 66 -> 66 -> 100, out 10 / 70 / 30)."""
    from gsvc_amd.model import GeneratorNet
    torch.manual_seed(out_dim + M)
    out_act = {"tanh": torch.nn.Tanh(), "sigmoid": torch.nn.Sigmoid(), None: None}[act]
    net = GeneratorNet(50, out_dim, 100, 66, out_act=out_act).cuda()
    feat = (torch.randn(M, 50, device="cuda") * 2).requires_grad_(True)
    cond = torch.randn(M, 66, device="cuda").requires_grad_(True)
    g = torch.randn(M, out_dim, device="cuda")
    with torch.no_grad():
        # a ReLU pre-activation within rounding of 0 may land on either side of the kink in the two summation orders (its
        # derivative jumps): rows that have one take no part in the gradient comparison
        f = net.film
        pre = torch.cat([F.linear(cond, f.fc_gamma0.weight, f.fc_gamma0.bias), F.linear(cond, f.fc_beta0.weight, f.fc_beta0.bias)], 1)
        g[(pre.abs() < 1e-4).any(dim=1)] = 0
    y = net(feat, cond)
    assert y.grad_fn is not None and "Generator" in type(y.grad_fn).__name__      # the fused function ran
    (y * g).sum().backward()
    got = [feat.grad.clone(), cond.grad.clone()] + [p.grad.clone() for p in net.parameters()]
    feat.grad = cond.grad = None
    net.zero_grad()
    ref = _torch_generator(net, feat, cond)
    _close(y.detach(), ref.detach(), 2e-5)
    (ref * g).sum().backward()
    want = [feat.grad, cond.grad] + [p.grad for p in net.parameters()]
    for a, b in zip(got, want):
        _close(a, b, 1e-3)


@pytest.mark.parametrize("dims", [(116, 100, 100, 100, 100, 30), (192, 150, 100), (192, 50, 1), (192, 100, 100, 12), (66, 7)])
def test_gelu_sequential_chain_matches_torch(dims):
    """mlp_deform and the sub-networks of the EntropyParamsNets (Linear -> GELU -> ... -> Linear)."""
    from gsvc_amd.model import GeluSequential, Linear
    torch.manual_seed(sum(dims))
    mods = []
    for i in range(len(dims) - 1):
        mods.append(Linear(dims[i], dims[i + 1]))
        if i + 2 < len(dims):
            mods.append(torch.nn.GELU())
    net = GeluSequential(*mods).cuda()
    M = 5003
    x = torch.randn(M, dims[0], device="cuda").requires_grad_(True)
    g = torch.randn(M, dims[-1], device="cuda")
    y = net(x)
    assert "SeqGelu" in type(y.grad_fn).__name__
    (y * g).sum().backward()
    got = [x.grad.clone()] + [p.grad.clone() for p in net.parameters()]
    x.grad = None
    net.zero_grad()
    h = x
    for m in net:
        h = F.linear(h, m.weight, m.bias) if isinstance(m, torch.nn.Linear) else F.gelu(h)
    _close(y.detach(), h.detach(), 2e-5)
    (h * g).sum().backward()
    for a, b in zip(got, [x.grad] + [p.grad for p in net.parameters()]):
        _close(a, b, 1e-3)
    # no gradient wanted for the input: the first layer's dX product is skipped
    net.zero_grad()
    y2 = net(x.detach())
    (y2 * g).sum().backward()
    _close(net[0].weight.grad, got[1], 1e-6)


def test_chain_falls_back_for_small_batches_and_cpu():
    from gsvc_amd.model import GeneratorNet
    net = GeneratorNet(50, 10, 100, 66, out_act=torch.nn.Tanh())
    y = net(torch.randn(8, 50), torch.randn(8, 66))
    assert "Generator" not in type(y.grad_fn).__name__
    net = net.cuda()
    y = net(torch.randn(8, 50, device="cuda"), torch.randn(8, 66, device="cuda"))
    assert "Generator" not in type(y.grad_fn).__name__


def test_fused_positional_embedding_matches_the_embedders():
    """csrc/generate.hip k_embed_pe against the module-level expression (reference guassian.py:225-230 + utils/time_util.py)."""
    from types import SimpleNamespace
    from gsvc_amd.generate import _Segments, _embed_rows
    from gsvc_amd.time_util import get_embedder
    dev = torch.device("cuda")
    et, _ = get_embedder(16, 1)
    ez, _ = get_embedder(16, 1)
    pc = SimpleNamespace(embed_time_fn=et, embed_fn=ez)
    counts = [5000, 0, 4097, 3]
    seg = _Segments(counts, dev)
    torch.manual_seed(1)
    anchor = (torch.rand(sum(counts), 3, device=dev) - 0.5) * 0.4
    cams = [-0.031, 0.0, 0.0125, 0.2]
    frames = [SimpleNamespace(cam_pos=torch.tensor([0.0, 0.0, c])) for c in cams]
    got = _embed_rows(pc, frames, anchor, seg)
    cam_row = torch.tensor(cams, device=dev).index_select(0, seg.seg_id).unsqueeze(1)
    want = torch.cat([et(cam_row), ez(anchor[:, 2:] - cam_row)], dim=1)
    assert got.shape == want.shape == (sum(counts), 66)
    # same float32 arguments into sin / cos on both sides
    assert (got - want).abs().max().item() <= 1e-6
    # anchors that carry a gradient keep the differentiable path
    a2 = anchor.clone().requires_grad_(True)
    assert _embed_rows(pc, frames, a2, seg).requires_grad


def test_fused_row_gather_and_context_tail_match_the_tensor_paths(monkeypatch):
    """k_gather_rows (per-anchor parameters of the visible rows + getter activations, scatter-add backward) and k_ctx_post
    (split / clamp / exp tail of the entropy networks) against the index_select / clamp / exp expressions they replace."""
    from types import SimpleNamespace
    from gsvc_amd.generate import _gather_rows
    from gsvc_amd.model import _CtxPost
    dev = torch.device("cuda")
    torch.manual_seed(3)
    A, K, F = 5000, 10, 50
    vis = torch.cat([torch.randperm(A, device=dev)[:3000], torch.randperm(A, device=dev)[:2500]])      # anchors repeat across renders

    def model(decoded):
        torch.manual_seed(5)
        P = lambda *s, sc=1.0: torch.nn.Parameter(torch.randn(*s, device=dev) * sc)  # noqa: E731
        return SimpleNamespace(_anchor_feat=P(A, F), _offset=P(A, K, 3), _scaling=P(A, 6, sc=0.5), _mask=P(A, K, 1, sc=4.0),
                               decoded_version=decoded, scaling_activation=torch.exp)

    w = [torch.randn(vis.shape[0], *s, device=dev) for s in ((F,), (K, 3), (6,), (K, 1))]
    for decoded in (False, True):
        res = []
        for fused in (True, False):
            if fused:
                monkeypatch.delenv("GSVC_NO_FUSED_GATHER", raising=False)
            else:
                monkeypatch.setenv("GSVC_NO_FUSED_GATHER", "1")
            pc = model(decoded)
            outs = _gather_rows(pc, vis)
            sum((o * ww).sum() for o, ww in zip(outs, w)).backward()
            res.append(([o.detach() for o in outs], [pc._anchor_feat.grad, pc._offset.grad, pc._scaling.grad, pc._mask.grad]))
        for a, b in zip(res[0][0], res[1][0]):
            assert torch.equal(a, b)                       # same float operations in the same order
        for a, b in zip(res[0][1], res[1][1]):
            assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item())     # atomics reorder sums of <= 2 terms
    # rows that are a step plan's R ascending lists: the ranked backward (no atomics, every element written, view order)
    monkeypatch.delenv("GSVC_NO_FUSED_GATHER", raising=False)
    M = torch.rand(3, A, device=dev) < 0.4
    M[:, 17] = False                                              # an anchor no view holds gets exact zeros
    vis3 = torch.cat([m.nonzero().squeeze(1) for m in M])
    ranks = (M.view(-1), torch.cumsum(M.view(-1), dim=0))
    w3 = [torch.randn(vis3.shape[0], *s, device=dev) for s in ((F,), (K, 3), (6,), (K, 1))]
    for decoded in (False, True):
        res = []
        for rk in (ranks, ranks, None):
            pc = model(decoded)
            outs = _gather_rows(pc, vis3, rk)
            sum((o * ww).sum() for o, ww in zip(outs, w3)).backward()
            res.append([pc._anchor_feat.grad, pc._offset.grad, pc._scaling.grad, pc._mask.grad])
        for a, b, c in zip(*res):
            assert torch.equal(a, b)                               # deterministic
            assert (a - c).abs().max().item() <= 1e-5 * max(1.0, c.abs().max().item())
            assert not a[17].any()
    # context tail
    n, C = 4097, 30
    params = torch.randn(n, 2 * C, device=dev, requires_grad=True)
    q = (torch.randn(n, 1, device=dev) * 8).requires_grad_(True)          # some |q| > 10: the clamp is active
    with torch.no_grad():
        params[::7, C:] = -0.5                                            # scales below the 1e-9 floor
    gm, gs, ga = torch.randn(n, C, device=dev), torch.randn(n, C, device=dev), torch.randn(n, 1, device=dev)
    mean, scale, adj = _CtxPost.apply(params, q)
    ((mean * gm).sum() + (scale * gs).sum() + (adj * ga).sum()).backward()
    got = (mean.detach(), scale.detach(), adj.detach(), params.grad.clone(), q.grad.clone())
    params.grad = q.grad = None
    m2, s2 = params.split([C, C], dim=1)
    s2c, a2 = torch.clamp(s2, 1e-9), torch.exp(torch.clamp(q, min=-10, max=10))
    ((m2 * gm).sum() + (s2c * gs).sum() + (a2 * ga).sum()).backward()
    for a, b in zip(got, (m2.detach(), s2c.detach(), a2.detach(), params.grad, q.grad)):
        assert (a - b).abs().max().item() <= 1e-6 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("M,share", [(4096, False), (6001, False), (20011, False), (6001, True), (20011, True)])
def test_whole_network_chain_kernels_match_torch(M, share):
    """gsvc_generator_* / gsvc_deform_* (csrc/mlp_chain.hip: a 16-row block's activations stay in registers from the network's
    input to its output) through gsvc_amd.mlp.generate_all — the three GeneratorNets (out 10 tanh / 30 sigmoid / 70) and
    mlp_deform (116 -> 100 x4 -> 30) on the same (feature, condition) rows — against plain PyTorch fp32: the four outputs, the
    accumulated feature gradient and every weight / bias gradient (reference scene/gaussian_model.py:150-196, 468-489).
    M = 4096: whole 16-row blocks; 6001, 20011: ragged last block, odd row counts (8-byte aligned matrix bases)."""
    from gsvc_amd import mlp
    from gsvc_amd.model import GeluSequential, GeneratorNet, Linear
    torch.manual_seed(M)
    gens = [GeneratorNet(50, 10, 100, 66, out_act=torch.nn.Tanh()).cuda(), GeneratorNet(50, 30, 100, 66, out_act=torch.nn.Sigmoid()).cuda(),
            GeneratorNet(50, 70, 100, 66).cuda()]
    deform = GeluSequential(Linear(116, 100), torch.nn.GELU(), Linear(100, 100), torch.nn.GELU(), Linear(100, 100), torch.nn.GELU(),
                            Linear(100, 100), torch.nn.GELU(), Linear(100, 30)).cuda()
    lin = list(deform)[0::2]
    feat = (torch.randn(M, 50, device="cuda") * 2).requires_grad_(True)
    cond = torch.randn(M, 66, device="cuda")
    film = None
    if share:
        # shared FiLM rows (gsvc_film_rows): the rows are two "views" of overlapping anchor sets; a FiLM row = one anchor of the
        # union, its condition shared by the (up to) two chain rows that name it
        Mf = int(0.6 * M)
        cond_film = torch.randn(Mf, 66, device="cuda")
        gen = torch.Generator(device="cuda").manual_seed(M)
        na = M // 2
        in_a = torch.randperm(Mf, device="cuda", generator=gen)[:na].sort().values           # view a sees these FiLM rows
        in_b = torch.randperm(Mf, device="cuda", generator=gen)[:M - na].sort().values       # view b these
        row_of = torch.cat([in_a, in_b]).to(torch.int32)
        src_a = torch.full((Mf,), -1, dtype=torch.int32, device="cuda")
        src_b = torch.full((Mf,), -1, dtype=torch.int32, device="cuda")
        src_a[in_a] = torch.arange(na, dtype=torch.int32, device="cuda")
        src_b[in_b] = torch.arange(na, M, dtype=torch.int32, device="cuda")
        cond = cond_film.index_select(0, row_of.long())
        film = (cond_film, row_of, src_a, src_b)
    gs = [torch.randn(M, n, device="cuda") for n in (10, 30, 70, 30)]
    with torch.no_grad():
        # rows with a FiLM ReLU pre-activation within rounding of its kink take no part in the gradient comparison
        bad = torch.zeros(M, dtype=torch.bool, device="cuda")
        for net in gens:
            f = net.film
            pre = torch.cat([F.linear(cond, f.fc_gamma0.weight, f.fc_gamma0.bias), F.linear(cond, f.fc_beta0.weight, f.fc_beta0.bias)], 1)
            bad |= (pre.abs() < 1e-4).any(dim=1)
        for g in gs:
            g[bad] = 0
    assert mlp.chain_usable(feat, cond, gens, lin)
    outs = mlp.generate_all(gens, lin, feat, cond, film=film)
    assert "GenerateAll" in type(outs[0].grad_fn).__name__
    sum((o * g).sum() for o, g in zip(outs, gs)).backward()
    params = [p for net in gens for p in net.parameters()] + list(deform.parameters())
    got = [feat.grad.clone()] + [p.grad.clone() for p in params]
    feat.grad = None
    for p in params:
        p.grad = None
    refs = [_torch_generator(net, feat, cond) for net in gens]
    x = torch.cat([feat, cond], dim=1)
    for i, l in enumerate(lin):
        x = F.linear(x, l.weight, l.bias)
        if i + 1 < len(lin):
            x = F.gelu(x)
    refs.append(x)
    for o, r in zip(outs, refs):
        _close(o.detach(), r.detach(), 2e-5)
    sum((r * g).sum() for r, g in zip(refs, gs)).backward()
    want = [feat.grad] + [p.grad for p in params]
    for a, b in zip(got, want):
        _close(a, b, 1e-3)
    # a second backward through a fresh forward gives bit-identical gradients (no atomics anywhere)
    feat.grad = None
    for p in params:
        p.grad = None
    outs2 = mlp.generate_all(gens, lin, feat, cond, film=film)
    sum((o * g).sum() for o, g in zip(outs2, gs)).backward()
    for a, b in zip(got, [feat.grad] + [p.grad for p in params]):
        assert torch.equal(a, b)
    # forward only (decoder / evaluation): the same kernels without the stores a backward reads — the same numbers bit for bit
    with torch.no_grad():
        outs3 = mlp.generate_all(gens, lin, feat, cond, film=film)
    for o, r in zip(outs3, outs):
        assert o.grad_fn is None and torch.equal(o, r.detach())
    # widths without an instantiation are refused by the C-ABI (callers keep the layer path)
    import ctypes as C
    from gsvc_amd import _lib
    d = mlp._gen_desc(mlp._generator_params(gens[0]), 1, 10)
    d.hidden_dim = 96
    assert _lib.lib().gsvc_generator_saved_floats(C.byref(d), 16, 0) > 0
    assert _lib.lib().gsvc_generator_forward(C.byref(d), None, None, 16, None, None, None) == -3


def test_entropy_sub_networks_as_one_function_match_torch():
    """gsvc_amd.mlp.seq_gelu_many: the six sub-networks of the three EntropyParamsNets (reference scene/gaussian_model.py:198-232)
    on one [rows, 192] feature matrix — outputs, every weight / bias gradient and the input gradient that the first layers'
    products accumulate in place (GSVC_LIN_ADD) against plain PyTorch."""
    from gsvc_amd import mlp
    from gsvc_amd.model import Linear
    torch.manual_seed(5)
    M = 6001
    dims = [(192, 150, 100), (192, 50, 1), (192, 100, 100, 12), (192, 100, 1), (192, 150, 60), (192, 150, 1)]
    chains = [[Linear(a, b).cuda() for a, b in zip(d[:-1], d[1:])] for d in dims]
    x = torch.randn(M, 192, device="cuda").requires_grad_(True)
    gs = [torch.randn(M, d[-1], device="cuda") for d in dims]
    outs = mlp.seq_gelu_many(x, chains)
    assert "SeqGeluMany" in type(outs[0].grad_fn).__name__
    sum((o * g).sum() for o, g in zip(outs, gs)).backward()
    params = [p for c in chains for l in c for p in (l.weight, l.bias)]
    got = [x.grad.clone()] + [p.grad.clone() for p in params]
    x.grad = None
    for p in params:
        p.grad = None
    refs = []
    for c in chains:
        h = x
        for i, l in enumerate(c):
            h = F.linear(h, l.weight, l.bias)
            if i + 1 < len(c):
                h = F.gelu(h)
        refs.append(h)
    for o, r in zip(outs, refs):
        _close(o.detach(), r.detach(), 2e-5)
    sum((r * g).sum() for r, g in zip(refs, gs)).backward()
    for a, b in zip(got, [x.grad] + [p.grad for p in params]):
        _close(a, b, 1e-3)
    # an output nobody used contributes nothing (and its chain's weights get no gradient)
    x.grad = None
    for p in params:
        p.grad = None
    outs = mlp.seq_gelu_many(x, chains)
    (outs[0] * gs[0]).sum().backward()
    assert chains[1][0].weight.grad is None and chains[0][0].weight.grad is not None and x.grad is not None


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 4097, 52481, 65536])
def test_sum_of_products_in_one_launch_matches_the_accumulate_epilogue(M, monkeypatch):
    """gsvc_linear_accumulate_many (csrc/linear_accum.hip): Y = sum_p G_p W_p for the six first-layer input gradients of the entropy
    networks (reference scene/gaussian_model.py:198-232 read by 1569-1597) against float64 and against the product-by-product path."""
    import torch
    from gsvc_amd import mlp
    monkeypatch.setenv("GSVC_MANY_MIN_ROWS", "1")      # (the product path takes the one-launch form from 24 576 rows)
    torch.manual_seed(M)
    dev = torch.device("cuda")
    Ks = [150, 50, 100, 50, 150, 50]
    pairs = [(torch.randn(M, k, device=dev), torch.randn(k, 192, device=dev) * 0.1) for k in Ks]
    got = mlp._sum_of_products(pairs, 192)
    ref = sum(g.double() @ w.double() for g, w in pairs)
    scale = ref.abs().max().item()
    assert (got.double() - ref).abs().max().item() <= 2e-6 * scale
    os.environ["GSVC_NO_ACCUM_MANY"] = "1"
    switches.reload()
    try:
        old = mlp._sum_of_products(pairs, 192)
    finally:
        del os.environ["GSVC_NO_ACCUM_MANY"]
        switches.reload()
    assert (got - old).abs().max().item() <= 2e-6 * scale
    assert torch.equal(got, mlp._sum_of_products(pairs, 192))         # fixed order
    # a product the fused kernel does not take (odd K): the fallback answers
    odd = [(torch.randn(M, 51, device=dev), torch.randn(51, 192, device=dev)), pairs[0]]
    r2 = mlp._sum_of_products(odd, 192)
    assert (r2.double() - sum(g.double() @ w.double() for g, w in odd)).abs().max().item() <= 2e-5 * scale * 10


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 4097, 52481])
def test_first_layers_with_a_shared_input_in_one_launch(M, monkeypatch):
    """gsvc_linear_forward_shared_input (csrc/linear_accum.hip) against the layer kernel it replaces for the entropy networks' six
    first layers (same bits: the same MFMA order per output) and against float64."""
    from gsvc_amd import mlp
    monkeypatch.setenv("GSVC_MANY_MIN_ROWS", "1")      # the product path takes these launches from 24 576 rows; the kernels take any M
    torch.manual_seed(M)
    dev = torch.device("cuda")
    Ns = [150, 50, 100, 50, 150, 50]
    x = torch.randn(M, 192, device=dev)
    params, sizes = [], []
    for n in Ns:
        params += [torch.randn(n, 192, device=dev) * 0.1, torch.randn(n, device=dev), torch.randn(8, n, device=dev), torch.randn(8, device=dev)]
        sizes.append(2)
    got = mlp._first_layers_shared_input(x, sizes, params)
    assert got is not None
    for i, n in enumerate(Ns):
        w, b = params[4 * i], params[4 * i + 1]
        a_ref = torch.empty(M, n, device=dev)
        z_ref = mlp.linear_ex(x, w, b, mlp.EPI_GELU_DUAL, y2=a_ref)
        z64 = x.double() @ w.double().t() + b.double()
        assert (got[i][0].double() - z64).abs().max().item() <= 2e-6 * z64.abs().max().item()
        assert torch.equal(got[i][0], z_ref) and torch.equal(got[i][1], a_ref)
    # the module path end to end (forward values and every gradient) with and without the two fused launches
    chains = [[torch.nn.Linear(192, n).to(dev), torch.nn.Linear(n, 8).to(dev)] for n in Ns]
    gs = [torch.randn(M, 8, device=dev) for _ in Ns]

    def run():
        xx = x.clone().requires_grad_(True)
        for c in chains:
            for l in c:
                l.zero_grad()
        outs = mlp.seq_gelu_many(xx, chains)
        torch.autograd.backward(outs, gs)
        return [o.detach() for o in outs], xx.grad, [p.grad.clone() for c in chains for l in c for p in l.parameters()]
    o1, g1, p1 = run()
    os.environ["GSVC_NO_SHARED_INPUT"] = os.environ["GSVC_NO_ACCUM_MANY"] = "1"
    switches.reload()
    try:
        o0, g0, p0 = run()
    finally:
        del os.environ["GSVC_NO_SHARED_INPUT"], os.environ["GSVC_NO_ACCUM_MANY"]
        switches.reload()
    assert all(torch.equal(a, b) for a, b in zip(o1, o0))
    assert (g1 - g0).abs().max().item() <= 2e-6 * g0.abs().max().item()
    assert all(torch.equal(a, b) for a, b in zip(p1, p0))


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,Ns", [(3000, 100, [100, 30, 160]), (777, 64, [16, 48]), (20000, 192, [12, 1, 150])])
def test_shared_input_and_accumulate_entries_at_other_shapes(M, K, Ns):
    """The two multi-product entries through the C-ABI at shapes the entropy networks do not use (K < 192, N not a multiple of 16
    or of 4, a product without the GELU pair, an output narrower than 192)."""
    from gsvc_amd import _lib
    torch.manual_seed(M + K)
    dev = torch.device("cuda")
    L = _lib.lib()
    x = torch.randn(M, K, device=dev)
    Ws = [torch.randn(n, K, device=dev) * 0.2 for n in Ns]
    bs = [torch.randn(n, device=dev) for n in Ns]
    Ys = [torch.full((M, n), float("nan"), device=dev) for n in Ns]
    Y2 = [torch.full((M, n), float("nan"), device=dev) if i % 2 == 0 else None for i, n in enumerate(Ns)]
    jobs = (_lib.SharedInputJobC * len(Ns))()
    for i, n in enumerate(Ns):
        jobs[i] = _lib.SharedInputJobC(Ws[i].data_ptr(), bs[i].data_ptr() if i != 1 else None, Ys[i].data_ptr(),
                                       Y2[i].data_ptr() if Y2[i] is not None else None, n, 0)
    _lib.check(L.gsvc_linear_forward_shared_input(_lib.ptr(x), M, K, jobs, len(Ns), _lib.current_stream(dev)), "shared_input")
    for i, n in enumerate(Ns):
        ref = x.double() @ Ws[i].double().t() + (bs[i].double() if i != 1 else 0.0)
        tol = 3e-6 * max(ref.abs().max().item(), 1.0)
        assert (Ys[i].double() - ref).abs().max().item() <= tol
        if Y2[i] is not None:
            assert (Y2[i].double() - F.gelu(ref)).abs().max().item() <= 2 * tol
    # accumulate: sum of X_p W_p into an [M, N] output with N = Ns[0]
    N = Ns[0]
    Kp = [K, 50, 2]
    pairs = [(torch.randn(M, k, device=dev), torch.randn(k, N, device=dev) * 0.3) for k in Kp]
    out = torch.full((M, N), float("nan"), device=dev)
    aj = (_lib.AccumJobC * len(pairs))()
    for i, (g, w) in enumerate(pairs):
        aj[i] = _lib.AccumJobC(g.data_ptr(), w.data_ptr(), g.shape[1], 0)
    _lib.check(L.gsvc_linear_accumulate_many(aj, len(pairs), _lib.ptr(out), M, N, _lib.current_stream(dev)), "accumulate_many")
    ref = sum(g.double() @ w.double() for g, w in pairs)
    assert (out.double() - ref).abs().max().item() <= 3e-6 * max(ref.abs().max().item(), 1.0)
    # refusals: too many rows, an odd K
    assert L.gsvc_linear_accumulate_many(aj, len(pairs), _lib.ptr(out), 65537, N, _lib.current_stream(dev)) != 0
    bad = (_lib.AccumJobC * 1)(_lib.AccumJobC(pairs[0][0].data_ptr(), pairs[0][1].data_ptr(), 51, 0))
    assert L.gsvc_linear_accumulate_many(bad, 1, _lib.ptr(out), M, N, _lib.current_stream(dev)) != 0


@pytest.mark.parametrize("M,unused", [(4096, None), (53011, None), (7777, 1), (16, None)])
def test_quant_step_nets_in_one_launch_match_torch(M, unused):
    """gsvc_quant_step_nets_forward / _backward (csrc/mlp_chain.hip): the three Linear(192 -> 50) -> GELU -> Linear(50 -> 1)
    networks on the same rows against plain PyTorch fp32 modules — outputs, the input gradient and all twelve parameter
    gradients; ``unused``: an output nobody consumed (its gradient arrives as None)."""
    from gsvc_amd import mlp
    torch.manual_seed(M)
    nets = [torch.nn.Sequential(torch.nn.Linear(192, 50), torch.nn.GELU(), torch.nn.Linear(50, 1)).cuda() for _ in range(3)]
    x = torch.randn(M, 192, device="cuda", requires_grad=True)
    w = [torch.randn(M, 1, device="cuda") for _ in range(3)]
    got = mlp._QuantStepNets.apply(x, *[p for n in nets for p in (n[0].weight, n[0].bias, n[2].weight, n[2].bias)])
    sum((got[i] * w[i]).sum() for i in range(3) if i != unused).backward()
    g_x = x.grad.clone()
    g_p = [p.grad.clone() if p.grad is not None else None for n in nets for p in n.parameters()]
    x.grad = None
    for n in nets:
        n.zero_grad()
    want = [n(x) for n in nets]
    sum((want[i] * w[i]).sum() for i in range(3) if i != unused).backward()
    for i in range(3):
        assert (got[i] - want[i]).abs().max().item() <= 2e-5 * max(1.0, want[i].abs().max().item()), i
    assert (g_x - x.grad).abs().max().item() <= 1e-4 * x.grad.abs().max().item()
    k = 0
    for ni, n in enumerate(nets):
        for name, p in n.named_parameters():
            if ni == unused:
                assert g_p[k] is None or float(g_p[k].abs().max()) == 0.0, (ni, name)
            else:
                scale = p.grad.abs().max().item()
                assert (g_p[k] - p.grad).abs().max().item() <= 1e-3 * scale + 1e-12, (ni, name, (g_p[k] - p.grad).abs().max().item(), scale)
            k += 1
    if M >= 4096:
        assert mlp.quant_step_nets_usable(x.detach(), nets)
