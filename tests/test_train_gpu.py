"""GPU test of the fitting step (gsvc_amd/train.py): every GenerateMode phase steps, the loss falls on a tiny
synthetic video, densification statistics accumulate, and no parameter turns non-finite."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(anchors=6000, H=96, W=160, T=12, seed=0):
    from gsvc_amd.arguments import ModelParams, OptimizationParams, PipelineParams
    from gsvc_amd.frame import SyntheticFrameCube
    from gsvc_amd.model import GaussianModel
    from gsvc_amd.train import Trainer
    mp = ModelParams()
    mp.grid_feature_dim = 2
    opt = OptimizationParams()
    opt.lmbda = 0.004
    cube = SyntheticFrameCube(H, W, T, device="cuda")
    mp.threshold = 3.0 / cube.scale
    torch.manual_seed(seed)
    np.random.seed(seed)
    pc = GaussianModel(mp, 16, 5, 0.001, 3, 16, 4, False, n_features_per_level=2, log2_hashmap_size=10, log2_hashmap_size_2D=12,
                       resolutions_list=(18, 24, 33, 44), resolutions_list_2D=(130, 258), device="cuda")
    rng = np.random.default_rng(seed)
    lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
    pc.create_from_points(rng.uniform(lim, -lim, (anchors, 3)), spatial_lr_scale=1.0)
    pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
    return pc, cube, opt, PipelineParams(), mp, Trainer


def test_step_runs_in_every_phase_and_loss_falls():
    from gsvc_amd.generate import GenerateMode
    pc, cube, opt, pipe, mp, Trainer = _setup()
    opt.full_precision_training_total, opt.quantized_training_total = 30, 5
    opt.entropy_constrained_train_total, opt.ste_entropy_constrained_train_total = 5, 5
    opt.start_stat, opt.pause_densification = 2, 0
    pc.training_setup(opt)
    tr = Trainer(pc, cube, opt, pipe, mp)
    losses, modes = [], []
    for it in range(1, 46):
        modes.append(tr.controller.render_mode)
        out = tr.step(it, frame_idx=4)
        losses.append(float(out.loss))
        assert np.isfinite(losses[-1])
    assert modes[0] == GenerateMode.TRAINING_FULL_PRECISION and modes[31] == GenerateMode.TRAINING_QUANTIZED
    assert modes[36] == GenerateMode.TRAINING_ENTROPY and modes[41] == GenerateMode.TRAININ_STE_ENTROPY
    assert np.mean(losses[25:30]) < 0.8 * np.mean(losses[0:3])          # distortion falls in the full-precision phase
    assert out.image1.shape == (3, 96, 160) and int(out.active_gaussians) > 0
    assert all(r.entropy_constrained for r in out.renders)
    assert pc.anchor_demon.sum() > 0 and pc.offset_denom.sum() > 0 and pc.offset_gradient_accum.sum() > 0
    for n, p in pc.named_parameters():
        assert torch.isfinite(p).all(), n
    # the hash tables and the entropy nets received updates in the entropy phase
    assert pc.optimizer.state[pc.encoding_xyz.encoding_xyz.params]["step"] > 0
    assert pc.optimizer.state[pc.mlp_feature_enet.dist_net[0].weight]["step"] > 0


def test_multi_view_visibility_equals_the_per_view_filter():
    """prefilter_voxels_many (one launch, activations folded in, quantised anchors cached) = rasterizer.visible_filter on the
    activated tensors, view by view; the cache follows an in-place change of the anchors and a new bound."""
    from gsvc_amd.ortho_gaussian_renderer.preprocess import RawGeometry, prefilter_geometry, prefilter_voxels_many, raster_settings_for
    from gsvc_amd.rasterizer import GaussianRasterizer
    pc, cube, opt, pipe, mp, Trainer = _setup(anchors=5003)
    pc.training_setup(opt)
    with torch.no_grad():
        pc._rotation.copy_(torch.randn_like(pc._rotation))           # un-normalised
        pc._scaling.add_(torch.randn_like(pc._scaling) * 0.5)
    bg = torch.zeros(3, device="cuda")
    frames = [cube[i] for i in (0, 3, 4, 5, 7, 8, 9, 10, 11)]        # nine views: two launches

    def check():
        geo = prefilter_geometry(pc)
        assert isinstance(geo, RawGeometry)
        got = prefilter_voxels_many(frames, pc, pipe, bg, geometry=geo)
        seen = 0
        for f, m in zip(frames, got):
            r = GaussianRasterizer(raster_settings=raster_settings_for(f, pc, pipe, bg))
            ref = r.visible_filter(means3D=pc.get_anchor.contiguous(), scales=pc.get_scaling[:, :3].contiguous(),
                                   rotations=pc.get_rotation.contiguous()) > 0
            assert m.dtype == torch.bool and int((m != ref).sum()) <= 1          # (a radius on the boundary may round differently)
            seen += int(ref.sum())
        assert 0 < seen < len(frames) * pc._anchor.shape[0]
        return geo[0]

    a0 = check()
    pc.anchor_static = True
    a1 = check()
    assert check().data_ptr() == a1.data_ptr() and torch.equal(a0, a1)          # cached
    with torch.no_grad():
        pc._anchor.mul_(0.9)
    a2 = check()
    assert a2.data_ptr() != a1.data_ptr()
    pc.update_anchor_bound(cube.x_min * 1.5, cube.y_min * 1.5, cube.z_min * 1.5)
    assert check().data_ptr() != a2.data_ptr()


def test_averaged_image_is_view_symmetric():
    """(image_f + flip(image_b)) / 2 does not depend on which of the two views is called forward."""
    from gsvc_amd.generate import GenerateMode
    from gsvc_amd.ortho_gaussian_renderer import render
    pc, cube, opt, pipe, mp, _ = _setup(anchors=3000)
    bg = torch.zeros(3)
    fr = cube.get_dummy_frame(5)
    with torch.no_grad():
        f = render(fr, pc, pipe, bg, mode=GenerateMode.TRAINING_FULL_PRECISION).rendered_image
        fr.view_matrix, fr.view_matrix_s = fr.view_matrix_s, fr.view_matrix
        b = render(fr, pc, pipe, bg, mode=GenerateMode.TRAINING_FULL_PRECISION).rendered_image
    avg1 = (f + torch.flip(b, dims=(-1,))) / 2
    avg2 = (b + torch.flip(f, dims=(-1,))) / 2
    assert torch.allclose(avg1, torch.flip(avg2, dims=(-1,)), atol=1e-6)
    assert (f - torch.flip(b, dims=(-1,))).abs().max() < 0.5   # same Gaussians at the same pixels, reversed depth order


def test_batched_generation_equals_per_render():
    """render_many (one generation pass for the 4 views of a step) returns what 4 render() calls return in the
    deterministic mode; in the entropy mode the per-render statistics (rate, masks) agree up to the different
    noise draws (checked on the noise-free pieces: visible sets, rate of a zero-noise replay)."""
    import copy
    from gsvc_amd.generate import GenerateMode
    from gsvc_amd.ortho_gaussian_renderer import render, render_many
    pc, cube, opt, pipe, mp, _ = _setup(anchors=4000)
    bg = torch.zeros(3)
    frames = []
    for idx in (4, 5):
        fr = cube[idx]
        back = copy.copy(fr)
        back.view_matrix, back.view_matrix_s = fr.view_matrix_s, fr.view_matrix
        frames += [fr, back]
    with torch.no_grad():
        many = render_many(frames, pc, pipe, bg, mode=GenerateMode.TRAINING_FULL_PRECISION)
        single = [render(f, pc, pipe, bg, mode=GenerateMode.TRAINING_FULL_PRECISION) for f in frames]
        dense = render_many(frames, pc, pipe, bg, mode=GenerateMode.TRAINING_FULL_PRECISION, dense=True)
    # un-compacted form: same image; per-Gaussian outputs agree on the "opacity > 0" rows, the other rows are culled
    from gsvc_amd.rasterizer import resolve_deferred
    counts, overflowed = resolve_deferred([d.raster_state for d in dense])
    assert not overflowed
    for d, b, n in zip(dense, single, counts):
        assert d.dense and d.num_rendered is None and n == b.num_rendered
        assert torch.equal(d.selection_mask, b.selection_mask) and torch.equal(d.visible_index, b.visible_mask.nonzero().squeeze(1))
        assert torch.equal(d.radii[d.selection_mask], b.radii) and int(d.radii[~d.selection_mask].abs().sum()) == 0
        assert torch.allclose(d.rendered_image, b.rendered_image, atol=2e-5)
        assert torch.allclose(d.scaling[d.selection_mask], b.scaling, atol=1e-6)
        assert int(d.active_gaussains) == int(b.active_gaussains)
    for a, b in zip(many, single):
        assert torch.equal(a.visible_mask, b.visible_mask) and torch.equal(a.selection_mask, b.selection_mask)
        assert a.num_rendered == b.num_rendered and torch.equal(a.radii, b.radii)
        assert torch.allclose(a.rendered_image, b.rendered_image, atol=2e-5)
        assert torch.allclose(a.neural_opacity, b.neural_opacity, atol=1e-5)
        assert torch.allclose(a.generated_gaussians.concatenated_all, b.generated_gaussians.concatenated_all, atol=1e-5)
    # entropy mode with the noise replaced by zeros and every anchor sampled: per-render rates must agree
    import gsvc_amd.generate as G
    old_rate, old_uniform, old_rand = G.SAMPLE_RATE, torch.Tensor.uniform_, torch.rand_like
    try:
        G.SAMPLE_RATE = 2.0
        torch.Tensor.uniform_ = lambda t, *a, **k: t.zero_()
        torch.rand_like = lambda x, *a, **k: torch.zeros_like(x)
        with torch.no_grad():
            many = render_many(frames, pc, pipe, bg, mode=GenerateMode.TRAINING_ENTROPY)
            single = [render(f, pc, pipe, bg, mode=GenerateMode.TRAINING_ENTROPY) for f in frames]
    finally:
        G.SAMPLE_RATE, torch.Tensor.uniform_, torch.rand_like = old_rate, old_uniform, old_rand
    for a, b in zip(many, single):
        for nm in ("bit_per_param", "bit_per_feat_param", "bit_per_scaling_param", "bit_per_offsets_param"):
            assert abs(float(getattr(a, nm)) - float(getattr(b, nm))) < 1e-4 * max(1.0, abs(float(getattr(b, nm)))), nm
        assert torch.allclose(a.rendered_image, b.rendered_image, atol=2e-5)


@pytest.mark.parametrize("entropy", [False, True])
def test_dense_step_equals_per_render_step(entropy):
    """Trainer(batched=True) — one un-compacted generation pass, sync-free statistics and optical-flow loss, deferred
    rasterizer counters — against Trainer(batched=False) — 4 reference-style render() calls with compaction and
    boolean-mask indexing — from the same parameters: same loss, same gradients, same densification statistics.
    (Deterministic modes only: the noise modes draw per call.)"""
    res = []
    for batched in (True, False):
        pc, cube, opt, pipe, mp, Trainer = _setup(anchors=5000, seed=3)
        if entropy:      # STE entropy mode: deterministic, rate term and hash-grid context active
            opt.full_precision_training_total = opt.quantized_training_total = opt.entropy_constrained_train_total = 0
            opt.ste_entropy_constrained_train_total = 100
        else:
            opt.full_precision_training_total = 100
        opt.start_stat, opt.pause_densification, opt.iterations = 0, 0, 1      # iteration 1 == iterations: no Adam step
        pc.training_setup(opt)
        import gsvc_amd.generate as G
        old = G.SAMPLE_RATE
        G.SAMPLE_RATE = 2.0         # rate over every visible anchor (the 5 % sample is a random draw)
        try:
            out = Trainer(pc, cube, opt, pipe, mp, batched=batched).step(1, frame_idx=5)
        finally:
            G.SAMPLE_RATE = old
        grads = {n: p.grad.clone() for n, p in pc.named_parameters() if p.grad is not None}
        res.append((float(out.loss), grads, pc.opacity_accum.clone(), pc.anchor_demon.clone(), pc.offset_gradient_accum.clone(),
                    pc.offset_denom.clone(), out.image1.clone(), [r.num_rendered for r in out.renders]))
    (la, ga, oa, da, ofa, oda, ia, na), (lb, gb, ob, db, ofb, odb, ib, nb) = res
    assert na == nb and abs(la - lb) < 1e-5 * max(1.0, abs(lb))
    # with a non-zero anchor learning rate the batched step keeps the anchor gradient
    pc, cube, opt, pipe, mp, Trainer = _setup(anchors=3000, seed=3)
    opt.position_lr_init = 1e-4
    opt.full_precision_training_total, opt.iterations = 100, 1
    pc.training_setup(opt)
    Trainer(pc, cube, opt, pipe, mp, batched=True).step(1, frame_idx=5)
    assert pc._anchor.grad is not None and float(pc._anchor.grad.abs().sum()) > 0
    assert torch.allclose(ia, ib, atol=2e-5)
    assert torch.equal(da, db) and torch.equal(oda, odb)
    assert torch.allclose(oa, ob, rtol=1e-5, atol=1e-6) and torch.allclose(ofa, ofb, rtol=1e-3, atol=1e-9)
    # the batched step skips the gradient of the anchor positions (trained with learning rate 0)
    # (and, in the STE mode, leaves the quant_step networks without a gradient where the reference-style path carries zeros:
    # the steps enter detached and the batched step never builds a graph through them)
    missing = set(gb) - set(ga)
    assert "_anchor" in missing and set(ga) <= set(gb) and len(ga) > 20
    for n in missing - {"_anchor"}:
        assert "quant_step_net" in n and float(gb[n].abs().max()) == 0.0, n
    for n in ga:
        scale = gb[n].abs().max().item()
        assert (ga[n] - gb[n]).abs().max().item() <= 2e-3 * scale + 1e-12, n


def test_fitting_at_the_configs0_shape():
    """BASELINE.json configs[0]: 8 synthetic 256 x 256 frames, 5 k Gaussians (500 anchors x K = 10), lambda = 0 (the full-precision
    phase: no rate term).  The batched step equals the reference-style per-render step there too (loss, gradients, accumulators),
    and fitting lowers the loss."""
    from gsvc_amd.arguments import ModelParams, OptimizationParams, PipelineParams
    from gsvc_amd.frame import SyntheticFrameCube
    from gsvc_amd.model import GaussianModel
    from gsvc_amd.train import Trainer

    def setup():
        mp = ModelParams()
        mp.grid_feature_dim = 2
        opt = OptimizationParams()
        opt.lmbda = 0.0
        cube = SyntheticFrameCube(256, 256, 8, device="cuda")
        mp.threshold = 4.0 / cube.scale                    # the whole 8-frame video inside the slab
        torch.manual_seed(0)
        np.random.seed(0)
        pc = GaussianModel(mp, 16, 10, 0.001, 3, 16, 4, False, n_features_per_level=2, log2_hashmap_size=10, log2_hashmap_size_2D=12,
                           resolutions_list=(18, 24, 33, 44), resolutions_list_2D=(130, 258), device="cuda")
        rng = np.random.default_rng(0)
        lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
        pc.create_from_points(rng.uniform(lim, -lim, (500, 3)), spatial_lr_scale=1.0)
        pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
        opt.full_precision_training_total = 10 ** 6
        opt.start_stat, opt.pause_densification = 0, 0
        return pc, cube, opt, PipelineParams(), mp

    res = []
    for batched in (True, False):
        pc, cube, opt, pipe, mp = setup()
        opt.iterations = 1                                  # iteration 1 == iterations: no Adam step
        pc.training_setup(opt)
        out = Trainer(pc, cube, opt, pipe, mp, batched=batched).step(1, frame_idx=3)
        res.append((float(out.loss), {n: p.grad.clone() for n, p in pc.named_parameters() if p.grad is not None},
                    pc.opacity_accum.clone(), pc.offset_denom.clone(), [r.num_rendered for r in out.renders]))
    (la, ga, oa, da, na), (lb, gb, ob, db, nb) = res
    assert na == nb and sum(na) > 1000 and abs(la - lb) < 1e-5 * max(1.0, abs(lb))
    assert torch.allclose(oa, ob, rtol=1e-5, atol=1e-6) and torch.equal(da, db)
    for n in ga:
        assert (ga[n] - gb[n]).abs().max().item() <= 2e-3 * gb[n].abs().max().item() + 1e-12, n
    pc, cube, opt, pipe, mp = setup()
    pc.training_setup(opt)
    tr = Trainer(pc, cube, opt, pipe, mp)
    losses = [float(tr.step(it, frame_idx=it % 7).loss) for it in range(1, 61)]
    assert np.isfinite(losses).all() and np.mean(losses[-7:]) < 0.85 * np.mean(losses[:7])


def test_step_plan_matches_the_per_view_index_lists(monkeypatch):
    """A StepPlan (visible anchors of the four views, their union, the 5 % rate sample: gsvc_amd.generate.StepPlan, masks from
    csrc/generate.hip k_plan_masks) against the expressions it replaces, and against its own unfused form with the same draws."""
    import gsvc_amd.generate as G
    from gsvc_amd.ortho_gaussian_renderer import plan_views
    pc, cube, opt, pipe, mp, Trainer = _setup(anchors=7000, seed=5)
    pc.training_setup(opt)
    with torch.no_grad():
        pc._mask[::3] = -9.0                          # anchors without a live offset never enter the sample
    tr = Trainer(pc, cube, opt, pipe, mp)
    views = tr._views(4)
    plans = []
    for fused in (True, False):
        if fused:
            monkeypatch.delenv("GSVC_NO_FUSED_PLAN", raising=False)
        else:
            monkeypatch.setenv("GSVC_NO_FUSED_PLAN", "1")
        torch.manual_seed(11)
        plans.append(plan_views(views, pc, pipe, tr.background, G.GenerateMode.TRAINING_ENTROPY).resolve())
    a, b = plans
    assert len(a.vis_list) == 4 and sum(v.numel() for v in a.vis_list) > 1000
    for va, vb, m in zip(a.vis_list, b.vis_list, a.visible_masks):
        assert torch.equal(va, vb) and torch.equal(va, m.nonzero().squeeze(1))
    union = torch.stack(a.visible_masks).any(dim=0)
    assert torch.equal(a.distinct, b.distinct) and torch.equal(a.distinct, union.nonzero().squeeze(1))
    assert torch.equal(a.sel, b.sel) and torch.equal(a.pos, b.pos) and torch.equal(a.ranks[0], b.ranks[0]) and torch.equal(a.ranks[1], b.ranks[1])
    vis = torch.cat(a.vis_list)
    live = pc.get_mask_anchor
    assert 0 < a.sel.numel() < 0.1 * vis.numel() and bool(live[vis[a.sel]].all()) and bool((a.sel[1:] > a.sel[:-1]).all())


@pytest.mark.parametrize("phase", ["entropy", "full", "quantized", "ste"])
def test_early_plan_steps_equal_plain_steps(monkeypatch, phase):
    """Steps of every phase of the schedule with the next step's plan queued from inside the backward (late row gather, early
    guarded Adam of _scaling / _mask — of _mask alone in the STE phase, whose attributes are detached: Trainer._early_tail) against
    the same steps with everything at the end of the step: the same frames, the same random draws in the same order, the same
    parameters after four steps (sums of <= 3 gradient terms may be taken in another order)."""
    import gsvc_amd.train as T
    totals = {"full": (1000, 0, 0, 0), "quantized": (0, 1000, 0, 0), "entropy": (0, 0, 1000, 0), "ste": (0, 0, 0, 1000)}[phase]
    res = []
    for early in (True, False):
        if early:
            monkeypatch.delenv("GSVC_NO_EARLY_PLAN", raising=False)
            monkeypatch.delenv("GSVC_NO_LATE_ROWS", raising=False)
            monkeypatch.setenv("GSVC_EARLY_PLAN", "1")              # a step this small does not take the early path by itself
        else:
            monkeypatch.delenv("GSVC_EARLY_PLAN", raising=False)
            monkeypatch.setenv("GSVC_NO_EARLY_PLAN", "1")
            monkeypatch.setenv("GSVC_NO_LATE_ROWS", "1")
        pc, cube, opt, pipe, mp, Trainer = _setup(anchors=6000, seed=4)
        (opt.full_precision_training_total, opt.quantized_training_total, opt.entropy_constrained_train_total,
         opt.ste_entropy_constrained_train_total) = totals
        opt.start_stat, opt.update_until, opt.pause_densification = 0, 10 ** 9, 0
        opt.update_from = 10 ** 9                       # no densification in these steps
        pc.training_setup(opt)
        torch.manual_seed(7)
        tr = Trainer(pc, cube, opt, pipe, mp)
        calls = []
        orig = tr._early_tail
        tr._early_tail = lambda renders, params: (calls.append(1), orig(renders, params))[1]
        losses = [float(tr.step(i + 1).loss) for i in range(4)]
        assert len(calls) == (4 if early else 0)
        assert tr._plan is not None
        res.append((losses, {n: p.detach().clone() for n, p in pc.named_parameters()},
                    {n: int(pc.optimizer.state[p]["step"]) for n, p in pc.named_parameters() if p in pc.optimizer.state}))
    (la, pa, sa), (lb, pb, sb) = res
    assert sa == sb and set(sa.values()) == {4}
    for x, y in zip(la, lb):
        assert abs(x - y) <= 1e-5 * max(1.0, abs(y)), (la, lb)
    # Adam divides by the gradient's own magnitude: an element whose gradient is rounding noise may move by a whole learning
    # rate either way, so single elements are not compared — almost all of them must agree closely
    for n in pa:
        scale = max(1e-6, pb[n].abs().max().item())
        off = ((pa[n] - pb[n]).abs() > 1e-4 * scale).float().mean().item()
        assert off < 5e-3, (n, off)


def test_rate_on_its_own_stream_equals_the_rate_inside_the_generation_pass(monkeypatch):
    """TRAINING_ENTROPY steps with the sampled rate issued behind the rasterizer's launches on a stream of its own
    (gsvc_amd/generate.py finish_deferred_rate: its kernels run under the compositing kernels, forward and backward) against
    GSVC_NO_RATE_OVERLAP=1 (inside the generation pass, on the step's stream): the same draws, losses and parameters after four
    steps (as in test_early_plan_steps_equal_plain_steps: Adam turns rounding noise in a near-zero gradient into a whole step).  A missing wait between the streams would show as a different loss or as NaN."""
    res = []
    for overlap in (True, False):
        if overlap:
            monkeypatch.delenv("GSVC_NO_RATE_OVERLAP", raising=False)
        else:
            monkeypatch.setenv("GSVC_NO_RATE_OVERLAP", "1")
        monkeypatch.setenv("GSVC_EARLY_PLAN", "1")
        import gsvc_amd.generate as G
        monkeypatch.setattr(G, "SMALL_WORK_MIN_ROWS", 0)      # a step this small does not use the third stream by itself
        pc, cube, opt, pipe, mp, Trainer = _setup(anchors=6000, seed=4)
        opt.full_precision_training_total = opt.quantized_training_total = 0
        opt.entropy_constrained_train_total = 1000
        opt.start_stat, opt.update_until, opt.pause_densification, opt.update_from = 0, 10 ** 9, 0, 10 ** 9
        pc.training_setup(opt)
        torch.manual_seed(7)
        tr = Trainer(pc, cube, opt, pipe, mp)
        calls = []
        real = G.finish_deferred_rate
        import gsvc_amd.ortho_gaussian_renderer.renderer as RR
        monkeypatch.setattr(RR, "finish_deferred_rate", lambda gss: (calls.append(getattr(gss[0].batch, "deferred_rate", None) is not None), real(gss))[1])
        outs = [tr.step(i + 1) for i in range(4)]
        losses = [float(o.loss) for o in outs]
        assert all(np.isfinite(losses))
        assert all(o.renders[0].entropy_constrained and o.renders[0].bit_per_param is not None for o in outs)
        # what a caller keeps of a step does not keep its autograd graph (a kept graph undoes the overlap: train.py _release_graph)
        r0 = outs[-1].renders[0]
        assert not overlap or all(t.grad_fn is None for t in (outs[-1].loss, r0.rendered_image, r0.bit_per_param, r0.neural_opacity,
                                                              r0.generated_gaussians.xyz, r0.generated_gaussians.batch.scaling))
        # the first step has no plan (no planned sample): inline; from the second on the rate is deferred when overlap is on
        assert calls[1:] == [overlap] * 3, calls
        res.append((losses, {n: p.detach().clone() for n, p in pc.named_parameters()}))
    (la, pa), (lb, pb) = res
    for x, y in zip(la, lb):
        assert abs(x - y) <= 1e-5 * max(1.0, abs(y)), (la, lb)
    for n in pa:
        scale = max(1e-6, pb[n].abs().max().item())
        off = ((pa[n] - pb[n]).abs() > 1e-4 * scale).float().mean().item()
        assert off < 5e-3, (n, off)


def test_overflowing_step_with_early_plan_changes_nothing_before_its_repeat(monkeypatch):
    """A step whose rasterizer instance buffers overflow is repeated (gsvc_amd/train.py step()).  With the early plan its
    guarded Adam launch of _scaling / _mask has already been queued when the overflow is read back: the kernel sees the
    overflow words and writes nothing, the step counts it advanced are rewound, and the repeat updates every tensor once."""
    import gsvc_amd.rasterizer as RZ
    monkeypatch.setenv("GSVC_EARLY_PLAN", "1")
    monkeypatch.delenv("GSVC_NO_EARLY_PLAN", raising=False)
    pc, cube, opt, pipe, mp, Trainer = _setup(anchors=6000, seed=6)
    opt.full_precision_training_total = opt.quantized_training_total = 0
    opt.entropy_constrained_train_total = 1000
    opt.start_stat, opt.update_until, opt.pause_densification, opt.update_from = 0, 10 ** 9, 0, 10 ** 9
    pc.training_setup(opt)
    torch.manual_seed(3)
    tr = Trainer(pc, cube, opt, pipe, mp)
    tr.step(1)
    before = {n: p.detach().clone() for n, p in pc.named_parameters()}
    small = [4]                                     # the next four forwards (one step's renders) get a 200-instance buffer
    real = RZ.raster_forward

    def tiny(cs, *a, **k):
        if small[0] > 0 and k.get("max_instances") is None:
            small[0] -= 1
            k["max_instances"] = 200
        return real(cs, *a, **k)
    monkeypatch.setattr(RZ, "raster_forward", tiny)
    seen, early = [], []
    inner, tail = tr._step, tr._early_tail

    def spy(*a, **k):
        out = inner(*a, **k)
        if out is None:
            torch.cuda.synchronize()
            seen.append({n: p.detach().clone() for n, p in pc.named_parameters()})
        return out
    tr._step = spy
    tr._early_tail = lambda renders, params: (early.append(1), tail(renders, params))[1]
    out = tr.step(2)
    assert getattr(tr, "repeated_steps", 0) == 1 and len(seen) == 1 and len(early) == 1 and small[0] == 0
    for n, p in seen[0].items():
        assert torch.equal(p, before[n]), n          # the overflowed attempt left every parameter alone
    steps = {n: int(pc.optimizer.state[p]["step"]) for n, p in pc.named_parameters() if p in pc.optimizer.state}
    assert set(steps.values()) == {2}, steps
    assert not torch.equal(pc._scaling.detach(), before["_scaling"]) and not torch.equal(pc._mask.detach(), before["_mask"])
    assert torch.isfinite(out.loss)
    tr.step(3)                                       # and training goes on (the repeat drew its own plan)


@pytest.mark.parametrize("shape", ["configs2", "configs3"])
def test_dense_step_equals_per_render_step_at_cfg3_size(shape):
    """The same comparison at BASELINE.json configs[2] size with the production model: 1080p, 245 000 anchors x K = 10 in a
    64-frame cube (about 48 000 visible anchors / 480 000 Gaussians per render in the 16-frame slab), the 12 + 3 x 4-level
    hash grids with 8 features (cfg_20240919.yaml), rate term on (deterministic STE mode, rate over every visible
    anchor): the batched un-compacted step that bench.py times against four reference-style render() calls.
    "configs3": BASELINE.json configs[3]'s per-GPU workload, reference cfgs/cfg_20240919.yaml AS IS — 100 000 anchors
    (init_anchor_num), a 600-frame 1080p video, threshold = .05 (reference arguments/__init__.py:54: a +-48-frame z-slab, i.e.
    ~14 500 visible anchors / 145 000 Gaussians per render with 6x deeper tile lists per Gaussian footprint than the 16-frame
    slab), lambda = .004; `bench.py --workload train_step --cfg3` times this shape."""
    anchors, T, frame_idx, min_p = (245_000, 64, 30, 400_000) if shape == "configs2" else (100_000, 600, 300, 120_000)
    from gsvc_amd.arguments import cfg_20240919
    from gsvc_amd.frame import SyntheticFrameCube
    from gsvc_amd.model import GaussianModel
    from gsvc_amd.train import Trainer
    import gsvc_amd.generate as G
    res = []
    for batched in (True, False):
        mp_, opt, pipe = cfg_20240919()
        cube = SyntheticFrameCube(1080, 1920, T, seed=1234, device="cuda")
        if shape == "configs2":
            cube.materialize()
            mp_.threshold = 8.0 / cube.scale
        else:
            assert mp_.threshold == 0.05 and opt.lmbda == 0.004 and opt.init_anchor_num == anchors
        opt.full_precision_training_total = opt.quantized_training_total = opt.entropy_constrained_train_total = 0
        opt.ste_entropy_constrained_train_total = 100
        opt.start_stat, opt.pause_densification, opt.iterations = 0, 0, 1
        torch.manual_seed(0)
        np.random.seed(0)
        pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                           mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                           log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device="cuda")
        rng = np.random.default_rng(0)
        lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
        pc.create_from_points(rng.uniform(lim, -lim, (anchors, 3)), spatial_lr_scale=1.0)
        pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
        pc.training_setup(opt)
        old = G.SAMPLE_RATE
        G.SAMPLE_RATE = 2.0
        try:
            out = Trainer(pc, cube, opt, pipe, mp_, batched=batched).step(1, frame_idx=frame_idx)
        finally:
            G.SAMPLE_RATE = old
        grads = {n: p.grad.clone() for n, p in pc.named_parameters() if p.grad is not None}
        res.append((float(out.loss), grads, pc.opacity_accum.clone(), pc.anchor_demon.clone(), pc.offset_gradient_accum.clone(),
                    pc.offset_denom.clone(), out.image1.clone(), [r.num_rendered for r in out.renders],
                    [int(r.radii.numel()) for r in out.renders]))
        del pc, cube, out
        torch.cuda.empty_cache()
    (la, ga, oa, da, ofa, oda, ia, na, pa), (lb, gb, ob, db, ofb, odb, ib, nb, pb) = res
    assert min(pa) > min_p, pa                         # un-compacted: K Gaussians per visible anchor
    # instances per render: the two paths run their MLPs over different row sets (last-ulp differences in a Gaussian's scale), and
    # a 3-sigma radius on the fence of ceil() moves a tile rectangle: a handful of 1.3 M instances at the +-48-frame slab
    assert all(abs(a - b) <= 2e-5 * b for a, b in zip(na, nb)) and abs(la - lb) < 1e-5 * max(1.0, abs(lb)), (na, nb, la, lb)
    # 300-entry tile lists: a last-ulp difference in one Gaussian (the two paths run their MLPs over different row counts)
    # moves a pixel by ~1e-5; threshold decisions on the fence (alpha vs 1/255) by up to 1/255 on isolated pixels
    diff = (ia - ib).abs()
    assert diff.max().item() < 5e-3 and (diff > 5e-5).float().mean().item() < 1e-4, (diff.max().item(), (diff > 5e-5).float().mean().item())
    assert torch.equal(da, db) and torch.equal(oda, odb)
    assert torch.allclose(oa, ob, rtol=1e-5, atol=1e-6)
    # accumulated screen-space gradient norms: relative to the largest (a Gaussian whose gradient is 1e-6 of it is noise)
    assert (ofa - ofb).abs().max().item() <= 2e-3 * ofb.abs().max().item(), ((ofa - ofb).abs().max().item(), ofb.abs().max().item())
    missing = set(gb) - set(ga)       # STE mode: the steps enter detached, the batched step builds no graph through the quant_step nets
    assert "_anchor" in missing and len(ga) > 20
    for n in missing - {"_anchor"}:
        assert "quant_step_net" in n and float(gb[n].abs().max()) == 0.0, n
    for n in ga:
        scale = gb[n].abs().max().item()
        assert (ga[n] - gb[n]).abs().max().item() <= 2e-3 * scale + 1e-12, n


def test_fused_loss_terms_match_torch():
    """csrc/losses.hip against plain torch statements of the same terms: the regularisers over un-compacted per-render
    segments (values and gradients), and the optical-flow pair loss against the reference-style
    calc_optical_loss_one_frame run on compacted copies of the same renders."""
    from types import SimpleNamespace
    from gsvc_amd import loss_utils as LU
    g = torch.Generator().manual_seed(5)
    K, A = 4, 900
    # ---- regularisers
    counts = [37, 0, 250, 411]
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c * K)
    n = offs[-1]
    scaling = (torch.rand(n, 3, generator=g) * 0.1).cuda().requires_grad_(True)
    op = (torch.rand(n, 1, generator=g) * 2 - 1).cuda().requires_grad_(True)
    mask = (op.detach().view(-1) > 0)
    out = LU.render_regs(scaling, op, mask, offs)
    ref0 = sum((scaling[a:b].prod(1) * mask[a:b]).sum() / mask[a:b].sum() for a, b in zip(offs[:-1], offs[1:]) if b > a)
    ref1 = sum((1 - op[a:b]).mean() for a, b in zip(offs[:-1], offs[1:]) if b > a)
    # an empty render contributes 0/0 = nan in both statements (mean of an empty selection): compare without it
    offs2 = [o for i, o in enumerate(offs) if i == 0 or offs[i] != offs[i - 1]]
    out = LU.render_regs(scaling, op, mask, offs2)
    assert abs(float(out[0]) - float(ref0)) < 1e-6 * max(1.0, abs(float(ref0))) and abs(float(out[1]) - float(ref1)) < 1e-5
    (out[0] * 3.0 + out[1] * 0.5).backward()
    gs, go = scaling.grad.clone(), op.grad.clone()
    scaling.grad = op.grad = None
    (ref0 * 3.0 + ref1 * 0.5).backward()
    assert torch.allclose(gs, scaling.grad, rtol=1e-4, atol=1e-9) and torch.allclose(go, op.grad, rtol=1e-5, atol=1e-9)

    # ---- optical-flow pair
    H, W, scale = 48, 64, 32.0
    x_min, y_min = -W / 2 / scale, -H / 2 / scale
    flow = (torch.randn(2, H, W, generator=g) * 2).cuda()

    def fake_render(seed):
        gg = torch.Generator().manual_seed(seed)
        vis_mask = torch.rand(A, generator=gg) < 0.6
        vis = vis_mask.nonzero().squeeze(1).cuda()
        rows = vis.shape[0]
        world = torch.stack([torch.rand(rows * K, generator=gg) * 2.4 - 1.2, torch.rand(rows * K, generator=gg) * 1.8 - 0.9,
                             torch.rand(rows * K, generator=gg)], 1).cuda().requires_grad_(True)
        m = (torch.rand(rows * K, generator=gg) < 0.7).cuda()
        gsd = SimpleNamespace(world_xyz=world, mask=m)
        return SimpleNamespace(visible_mask=vis_mask.cuda(), visible_index=vis, generated_gaussians=gsd, dense=True), world, m

    r1, w1, m1 = fake_render(11)
    r2, w2, m2 = fake_render(12)
    w2.data[:, :2] = 0.0
    # make render 2 a displaced copy of render 1 where both exist, so that |d - uv| is informative
    loss = LU._optical_loss_dense(r1, r2, flow, x_min, y_min, scale, W, H, K)
    loss.backward()
    g1, g2 = w1.grad.clone(), w2.grad.clone()

    def compact(r, world, m):
        # reference-style result: concatenated_all rows = all K slots of the visible anchors; columns 0:6 scaling (ones),
        # 6:9 anchor (zeros), 19:22 offsets -> anchor + offsets * scaling[:, :3] = world
        n = world.shape[0]
        ca = torch.cat([torch.ones(n, 6, device="cuda"), torch.zeros(n, 13, device="cuda"), world], dim=1)
        return SimpleNamespace(visible_mask=r.visible_mask, generated_gaussians=SimpleNamespace(mask=m, concatenated_all=ca)), None

    w1.grad = w2.grad = None
    c1, _ = compact(r1, w1, m1)
    c2, _ = compact(r2, w2, m2)
    ref, pix, _ = LU.calc_optical_loss_one_frame(c1, c2, flow, x_min, y_min, scale, W, H, K)
    assert pix.shape[0] > 50
    assert abs(float(loss) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    ref.backward()
    assert torch.allclose(g1, w1.grad, rtol=1e-4, atol=1e-8) and torch.allclose(g2, w2.grad, rtol=1e-4, atol=1e-8)

    # ---- both pairs of a step over the concatenated renders (f1, b1, f2, b2) = the sum of the two per-pair losses
    rs = [fake_render(s_)[0] for s_ in (21, 22, 23, 24)]
    rs[1].generated_gaussians.mask[:] = False                 # a render with nothing alive: its pair contributes 0/0
    for empty_pair in (False, True):
        worlds = [r.generated_gaussians.world_xyz.detach().clone().requires_grad_(True) for r in rs]
        goff = [0]
        for wld in worlds:
            goff.append(goff[-1] + wld.shape[0])
        pairs = ((0, 2), (1, 3)) if empty_pair else ((0, 2),)
        world_all = torch.cat(worlds).detach().requires_grad_(True)
        many = LU._OpticalMany.apply(world_all, torch.cat([r.generated_gaussians.mask for r in rs]), torch.cat([r.visible_index for r in rs]),
                                     goff, pairs, flow, K, A, x_min, y_min, scale, W, H)
        for r, wld in zip(rs, worlds):
            r.generated_gaussians.world_xyz = wld
        each = sum(LU._optical_loss_dense(rs[a], rs[b], flow, x_min, y_min, scale, W, H, K) for a, b in pairs)
        if empty_pair:
            assert torch.isnan(many) and torch.isnan(each)
            continue
        assert float(many) == float(each)
        many.backward()
        each.backward()
        ref_g = torch.cat([wld.grad if wld.grad is not None else torch.zeros_like(wld) for wld in worlds])
        assert torch.equal(world_all.grad, ref_g) and float(ref_g.abs().sum()) > 0


def test_training_crosses_densification_steps():
    """The step keeps running through adjust_anchor (grow + prune every update_interval steps): parameter, statistic
    and Adam-state shapes stay consistent, the loss stays finite, the number of anchors changes."""
    pc, cube, opt, pipe, mp, Trainer = _setup(anchors=3000)
    opt.full_precision_training_total = 1000
    opt.start_stat, opt.update_from, opt.update_interval, opt.update_until, opt.pause_densification = 2, 6, 5, 40, 0
    opt.densify_grad_threshold, opt.success_threshold = 1e-7, 0.2
    pc.training_setup(opt)
    tr = Trainer(pc, cube, opt, pipe, mp)
    a0 = pc._anchor.shape[0]
    counts = []
    for it in range(1, 24):
        out = tr.step(it, frame_idx=4 + it % 3)
        assert np.isfinite(float(out.loss))
        counts.append(pc._anchor.shape[0])
    A, K = pc._anchor.shape[0], pc.n_offsets
    assert len(set(counts)) > 1 and A != a0
    assert pc._offset.shape == (A, K, 3) and pc._mask.shape == (A, K, 1) and pc._anchor_feat.shape[0] == A
    assert pc.opacity_accum.shape == (A, 1) and pc.anchor_demon.shape == (A, 1)
    assert pc.offset_gradient_accum.shape == (A * K, 1) and pc.offset_denom.shape == (A * K, 1)
    grp = [g for g in pc.optimizer.param_groups if g["name"] == "anchor_feat"][0]
    assert grp["params"][0] is pc._anchor_feat and pc.optimizer.state[pc._anchor_feat]["exp_avg"].shape == pc._anchor_feat.shape
    for n, p in pc.named_parameters():
        assert torch.isfinite(p).all(), n


def test_fused_adam_matches_torch_adam():
    """gsvc_amd.optim.FusedAdam (csrc/adam.hip, one launch for all tensors) against torch.optim.Adam over several steps with
    per-group learning rates, parameters that skip steps (no grad) and odd sizes."""
    from gsvc_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(9)
    shapes = [(1000, 50), (7,), (333, 10, 3), (1,), (4099,), (64, 64)]
    pa = [torch.randn(*s, generator=g).cuda().requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    groups = lambda ps: [{"params": [ps[0], ps[1]], "lr": 1e-2, "name": "a"}, {"params": [ps[2]], "lr": 3e-4, "name": "b"},
                         {"params": ps[3:], "lr": 0.0, "name": "c"}]  # noqa: E731
    oa, ob = FusedAdam(groups(pa), lr=0.0, eps=1e-15), torch.optim.Adam(groups(pb), lr=0.0, eps=1e-15)
    for it in range(6):
        for k, (x, y) in enumerate(zip(pa, pb)):
            if k == 1 and it % 2 == 0:
                x.grad = y.grad = None          # a parameter without gradient keeps its own step count
                continue
            gr = torch.randn(x.shape, generator=g).cuda() * (10.0 ** (k - 3))
            x.grad, y.grad = gr.clone(), gr.clone()
        for grp_a, grp_b in zip(oa.param_groups, ob.param_groups):
            grp_a["lr"] = grp_b["lr"] = grp_b["lr"] * 0.9 + 1e-5
        oa.step()
        ob.step()
    for x, y in zip(pa, pb):
        assert torch.allclose(x, y, rtol=1e-5, atol=2e-6)
        if y in ob.state:
            for key in ("exp_avg", "exp_avg_sq"):
                a, b = oa.state[x][key], ob.state[y][key]
                assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item(), key
            assert float(oa.state[x]["step"]) == float(ob.state[y]["step"])


def test_render_frames_equals_render_pair():
    """The batched decoder loop (one generation pass per batch of frames + one two-view pass per frame) returns the frames
    render_pair returns one by one."""
    from gsvc_amd.generate import GenerateMode
    from gsvc_amd.ortho_gaussian_renderer import render_frames, render_pair
    pc, cube, opt, pipe, mp, _ = _setup(anchors=4000)
    bg = torch.zeros(3)
    frames = [cube.get_dummy_frame(i) for i in range(2, 9)]
    batched = list(render_frames(frames, pc, pipe, bg, batch=3))
    assert len(batched) == len(frames)
    for fr, img in zip(frames, batched):
        ref = render_pair(fr, pc, pipe, bg, mode=GenerateMode.DECODING_AS_IS).rendered_image
        # the two generation paths differ in the last ulp (fused tail kernel vs torch ops), which can move a pixel across an
        # alpha >= 1/255 / T < 1e-4 decision: allow a handful of such pixels
        d = (img - ref).abs()
        assert (d > 2e-5).float().mean().item() < 2e-4 and d.max().item() < 5e-3


@pytest.mark.parametrize("per_row_q", [True, False])
def test_fused_noise_quant_matches_torch(per_row_q):
    """csrc/quant.hip against the PyTorch statement of the per-render UniformQuantizer (same noise tensor): values, the
    gradient w.r.t. x (1 where the clamp is inactive) and w.r.t. the per-row step Q."""
    import gsvc_amd.generate as G
    g = torch.Generator().manual_seed(21)
    counts = [300, 0, 1111, 64]
    seg = G._Segments(counts, torch.device("cuda"))
    rows = sum(counts)
    x = (torch.randn(rows, 10, 3, generator=g) * 3).cuda().requires_grad_(True)
    # a few rows far outside the +-15000-step window so that the clamp (and its dQ branch) is exercised
    x.data[5] = 4000.0
    x.data[700] = -3500.0
    Q = ((torch.rand(rows, 1, 1, generator=g) * 0.2 + 0.05).cuda().requires_grad_(True)) if per_row_q else 0.2
    noise = (torch.rand(rows, 30, generator=g) - 0.5).cuda()
    q_rows = Q.reshape(-1) if per_row_q else None
    y = G._NoiseQuant.apply(x.reshape(rows, 30), q_rows, 0.0 if per_row_q else Q, noise, seg.bounds).view(x.shape)
    lo, hi = G._seg_bounds(x, Q, seg)
    ref = torch.clamp(x / Q, min=lo, max=hi) * Q + noise.view(x.shape) * Q
    assert torch.allclose(y, ref, rtol=1e-5, atol=1e-5)
    w = torch.randn(x.shape, generator=g).cuda()
    (y * w).sum().backward()
    gx = x.grad.clone()
    gq = Q.grad.clone() if per_row_q else None
    x.grad = None
    if per_row_q:
        Q.grad = None
    (ref * w).sum().backward()
    assert torch.allclose(gx, x.grad, rtol=1e-5, atol=1e-6)
    if per_row_q:
        # inside the window autograd's dQ is a rounding residue of x/Q - x/Q: compare on the scale of the noise term
        assert (gq - Q.grad).abs().max().item() <= 1e-3 * Q.grad.abs().max().item()


@pytest.mark.gpu
def test_estimate_final_bits_matches_reference():
    """Bit accounting against the reference's own estimate_final_bits on the same tiny model
    (tests/golden/make_golden_bits.py; the reference ran on CPU with the grid oracle behind it)."""
    import os
    import numpy as np
    from gsvc_amd.arguments import ModelParams
    from gsvc_amd.model import GaussianModel
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g, b = np.load(os.path.join(here, "tiny_model.npz")), np.load(os.path.join(here, "final_bits.npz"))
    dev = torch.device("cuda")
    pc = GaussianModel(ModelParams(), feat_dim=8, n_offsets=4, voxel_size=0.001, update_depth=3, update_init_factor=16,
                       update_hierachy_factor=4, use_feat_bank=False, n_features_per_level=2, log2_hashmap_size=9,
                       log2_hashmap_size_2D=11, resolutions_list=(18, 24, 33), resolutions_list_2D=(130, 258), device=dev)
    sd = {k[4:]: torch.from_numpy(np.array(g[k])) for k in g.files if k.startswith("sd::")}
    for nm in ("_anchor", "_offset", "_mask", "_anchor_feat", "_scaling", "_rotation", "_opacity"):
        sd[nm] = torch.from_numpy(np.array(b["in::" + nm]))
        setattr(pc, nm, torch.nn.Parameter(sd[nm].clone().to(dev), requires_grad=nm not in ("_rotation", "_opacity")))
    pc.load_state_dict(sd, strict=True)
    pc.to(dev)
    pc.update_anchor_bound(float(g["x_lim"]), float(g["y_lim"]), float(g["z_lim"]))
    log, info = pc.estimate_final_bits()
    for f in ("bit_anchor", "bit_anchor_gpcc", "bit_mlp", "bit_mlp_encoded"):
        assert float(getattr(info, f)) == float(b["bits::" + f]), f
    for f in ("bit_feat", "bit_scaling", "bit_offsets", "bit_hash", "bit_masks"):
        ref = float(b["bits::" + f])
        assert abs(float(getattr(info, f)) - ref) <= 2e-5 * abs(ref) + 1e-3, (f, getattr(info, f), ref)
    assert log == str(b["log_info"])


def _run_bench(extra_env, *args, timeout=420):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # a rank that is still running after 300 s prints every thread's stack and exits: a deadlock fails the test with the
    # stacks in its message instead of hanging the suite
    env = dict(os.environ, GSVC_HANG_DUMP="300", **extra_env)
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], cwd=root, env=env, capture_output=True,
                          text=True, timeout=timeout)


@pytest.mark.gpu
def test_bench_headline_scene_is_frozen():
    """bench.py fits its untimed steps under GSVC_DETERMINISTIC=1 (round 6): two separate processes end in the SAME model — the
    checksum over every parameter's bits that the line carries — and the timed steps then see the same number of active Gaussians
    (the default-mode steps that follow move it by single Gaussians at most, not by the +-10 % a live fit draws); `--live-fit`
    says so in the line and `--scene-seed` names another scene."""
    import json
    small = ["--workload", "train_step", "--steps", "4", "--warmup", "2", "--pretrain", "24", "--anchors", "20000", "--height", "272",
             "--width", "480", "--no-cpu-baseline", "--no-side"]
    lines = []
    for extra in ([], [], ["--scene-seed", "3"], ["--live-fit"]):
        out = _run_bench({}, *small, *extra)
        assert out.returncode == 0, out.stderr[-2000:]
        lines.append(json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0]))
    a, b, other, live = [l["config"]["scene"] for l in lines]
    assert a["frozen"] and b["frozen"] and a["checksum"] == b["checksum"] and a["fit_steps"] == 24
    assert abs(lines[0]["config"]["active_per_render"] - lines[1]["config"]["active_per_render"]) <= 1e-3 * lines[0]["config"]["active_per_render"]
    assert other["frozen"] and other["checksum"] != a["checksum"] and other["seeds"]["trainer"] == 3
    assert not live["frozen"] and "live fit" in lines[3]["config"]["workload"] and a["checksum"] in lines[0]["config"]["workload"]


@pytest.mark.gpu
def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` (the driver's command form) must run TWO ranks or fail: with the single-GPU test knobs
    (both ranks on device 0, gloo) it starts torch.distributed.run as a child and relays rank 0's line with n_gpus = 2 —
    the whole multi-rank fitting step on GPU tensors: parameter broadcast, gradient all-reduce from the backward hooks, the
    early overflow decision, per-rank frame shards; without the knobs on a box with fewer than 2 GPUs it exits non-zero
    and prints no JSON line (it used to run one rank and report n_gpus 1)."""
    import json
    small = ["--workload", "train_step", "--steps", "3", "--warmup", "1", "--pretrain", "2", "--anchors", "20000",
             "--height", "272", "--width", "480", "--no-cpu-baseline"]
    out = _run_bench({"GSVC_DIST_BACKEND": "gloo", "GSVC_SHARE_GPU": "1"}, "--gpus", "2", *small)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["dist_backend"] == "gloo"
    assert res["steps"] == 3 and res["value"] > 0 and res["scaling"] == "weak"
    assert res["gradient_exchange"]["gradient_bytes_per_step"] > 0 and "exposed_ms_per_step" in res["gradient_exchange"]
    # every rank's own numbers, and the other per-anchor exchange (z-range ownership) measured in the same invocation (round 6)
    ge = res["gradient_exchange"]
    assert [r["rank"] for r in ge["per_rank"]] == [0, 1] and all(r["gradient_bytes_per_step"] > 0 and r["ms_per_step"] > 0 for r in ge["per_rank"])
    other = ge["other_exchange_same_run"]
    assert "error" not in other and "z-range" in other["per_anchor_exchange"], other
    assert [r["rank"] for r in other["per_rank"]] == [0, 1] and all(r["gradient_bytes_per_step"] > 0 for r in other["per_rank"])
    if torch.cuda.device_count() < 2:
        out = _run_bench({}, "--gpus", "2", *small)
        assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert "GPU(s)" in out.stderr


@pytest.mark.gpu
def test_bench_runs_its_data_parallel_path_on_a_one_rank_rccl_group():
    """bench.py's own multi-rank code — process-group setup as the driver's command form does it (backend "nccl", device_id), the
    parameter broadcast, the barriers around the timed region, the float64 SUM / MAX reductions of the line's numbers, the
    exchange-off pass with the re-broadcast — on a real RCCL communicator of ONE rank (GSVC_DP_FORCE=1), replicated exchange and
    z-range ownership: the line must say backend nccl and carry the gradient_exchange block."""
    import json
    small = ["--workload", "train_step", "--steps", "3", "--warmup", "1", "--pretrain", "2", "--anchors", "20000",
             "--height", "272", "--width", "480", "--no-cpu-baseline"]
    for extra, word in (({}, "dense"), ({"GSVC_DP_ZOWN": "1"}, "z-range")):
        out = _run_bench({"GSVC_DP_FORCE": "1", **extra}, "--gpus", "1", *small)
        assert out.returncode == 0, out.stderr[-2000:]
        res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        assert res["n_gpus"] == 1 and res["rccl_ranks"] == 1 and res["dist_backend"] == "nccl", (res["rccl_ranks"], res["dist_backend"])
        ge = res["gradient_exchange"]
        assert ge["gradient_bytes_per_step"] > 0 and word in ge["per_anchor_exchange"], ge
        other = ge["other_exchange_same_run"]          # the run's own exchange, then the other one, in one invocation
        assert "error" not in other and ("z-range" in other["per_anchor_exchange"]) == (word != "z-range"), other


def _run_dp_grad_worker(env_extra, ranks=2):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "_dp_grad_worker.py")]
    out = subprocess.run(cmd, cwd=root, env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=420)
    assert out.returncode == 0 and "DP_GRAD_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-2500:])
    return out.stdout


@pytest.mark.gpu
def test_two_rank_step_averages_gradients():
    """One data-parallel step leaves on every rank the mean of the two single-process gradients of the ranks' frame pairs
    (every parameter group; tests/_dp_grad_worker.py).  Both ranks on device 0 through gloo: runs on a 1-GPU box."""
    out = _run_dp_grad_worker({"GSVC_DIST_BACKEND": "gloo", "GSVC_SHARE_GPU": "1"})
    assert "backend=gloo ranks=2" in out and "planned=True sparse=True" in out, out


@pytest.mark.gpu
def test_two_rank_entropy_step_with_sparse_row_exchange():
    """The entropy-constrained step (rate, hash-table bits, the mask regulariser whose gradient touches every row of _mask) under
    data parallelism, STE form (deterministic): the planned step exchanges the per-anchor gradients as rows of the ranks' distinct
    visible anchors; the regulariser's dense gradient is added after the exchange (Trainer._add_mask_reg).  Same mean as two
    single-process steps."""
    out = _run_dp_grad_worker({"GSVC_DIST_BACKEND": "gloo", "GSVC_SHARE_GPU": "1", "GSVC_DP_MODE": "ste"})
    assert "planned=True sparse=True" in out, out
    out = _run_dp_grad_worker({"GSVC_DIST_BACKEND": "gloo", "GSVC_SHARE_GPU": "1", "GSVC_DP_MODE": "ste", "GSVC_DP_SPARSE": "0"})
    assert "planned=True sparse=False" in out, out


@pytest.mark.gpu
def test_two_rank_step_averages_gradients_at_configs3_shape():
    """The same at BASELINE.json configs[3]'s shape — reference cfgs/cfg_20240919.yaml as is: 100 000 anchors, 600 frames of
    1080p, threshold .05 (a +-48-frame slab) — two frame shards of 300 frames, one pair per rank (gloo, both ranks on device 0)."""
    assert "backend=gloo ranks=2" in _run_dp_grad_worker({"GSVC_DIST_BACKEND": "gloo", "GSVC_SHARE_GPU": "1",
                                                          "GSVC_DP_SHAPE": "configs3"})


@pytest.mark.gpu
def test_two_rank_step_averages_gradients_over_rccl():
    """The same over RCCL (backend "nccl"), one GPU per rank: the hook-driven asynchronous all-reduces run on RCCL's own
    streams next to the backward."""
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs 2 GPUs for one rank per GPU over RCCL; this box has {torch.cuda.device_count()}")
    assert "backend=nccl ranks=2" in _run_dp_grad_worker({"GSVC_DIST_BACKEND": "nccl"})


@pytest.mark.gpu
def test_one_rank_rccl_communicator_runs_every_collective_of_the_step():
    """What a single-GPU box can show of RCCL: GSVC_DP_FORCE=1 keeps the data-parallel machinery on for a process group of ONE
    rank, backend "nccl" — group creation (default + plan group), the hook-driven asynchronous all-reduces next to the backward, the
    flat bucket, the agreed order's broadcast, the row lists' all-gathers, the overflow MAX, the plan's count exchange all run on a
    real RCCL communicator (as the identity: the gradients must equal the single-process step's), with RCCL's device / dtype /
    contiguity rules and its stream ordering.  Dense, row-sparse and z-range-owned exchange."""
    out = _run_dp_grad_worker({"GSVC_DIST_BACKEND": "nccl", "GSVC_DP_FORCE": "1"}, ranks=1)
    assert "backend=nccl ranks=1" in out and "planned=True sparse=False" in out, out
    out = _run_dp_grad_worker({"GSVC_DIST_BACKEND": "nccl", "GSVC_DP_FORCE": "1", "GSVC_DP_SPARSE": "1", "GSVC_DP_MODE": "ste"}, ranks=1)
    assert "backend=nccl ranks=1" in out and "planned=True sparse=True" in out, out
    out = _run_dp_grad_worker({"GSVC_DIST_BACKEND": "nccl", "GSVC_DP_FORCE": "1", "GSVC_DP_ZOWN": "1"}, ranks=1)
    assert "backend=nccl ranks=1" in out and "zown=True" in out, out


@pytest.mark.gpu
def test_two_rank_step_with_z_range_ownership():
    """GSVC_DP_ZOWN=1 (gsvc_amd.dist.ZRangeOwnership, SURVEY 8e "Collective"): the per-anchor gradients of the halo rows travel to
    their owners; on the rows a rank owns the result is the mean of the two single-process gradients, in the deterministic
    (full-precision) step and in the entropy-constrained one (rate, mask regulariser added by the owner, the clamp centres' means
    from the owners' partial sums)."""
    out = _run_dp_grad_worker({"GSVC_DIST_BACKEND": "gloo", "GSVC_SHARE_GPU": "1", "GSVC_DP_ZOWN": "1"})
    assert "zown=True" in out and "per_anchor=0" not in out, out
    out = _run_dp_grad_worker({"GSVC_DIST_BACKEND": "gloo", "GSVC_SHARE_GPU": "1", "GSVC_DP_ZOWN": "1", "GSVC_DP_MODE": "ste"})
    assert "zown=True" in out, out


@pytest.mark.gpu
def test_two_ranks_through_every_phase_at_the_headline_shape():
    """Two data-parallel ranks (gloo, both on device 0) through FULL_PRECISION, QUANTIZED, TRAINING_ENTROPY and STE_ENTROPY steps at
    BASELINE.json configs[2] size, where the plan of the next step is queued from inside the backward in every phase and the
    generation runs once per frame in the two phases without noise: no rank waits for a collective another rank never launches,
    every loss is finite and the replicas' parameters stay identical (tests/_dp_phases_worker.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29400 + os.getpid() % 150
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "_dp_phases_worker.py")]
    out = subprocess.run(cmd, cwd=root, env=dict(os.environ, GSVC_DIST_BACKEND="gloo", GSVC_SHARE_GPU="1"), capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "DP_PHASES_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-2500:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("DP_PHASE ")]
    assert len(lines) == 4 and all("replicas_identical True" in l for l in lines), lines
    # the early plan ran in every phase — the STE phase too, whose detached attributes give _scaling / _offset / _anchor_feat no
    # gradient: the reducer agrees on a new launch order per phase, so no launch waits behind a hook that never fires
    assert all("early_steps 0" not in l for l in lines), lines


@pytest.mark.gpu
def test_z_range_ownership_through_every_phase_equals_the_replicated_run():
    """GSVC_DP_ZOWN=1 through the four phases at the headline shape (two ranks, 32-frame blocks, +-8-frame halo), with the check
    that no gradient row lies outside a rank's block + halo: finite losses, replicas identical once made whole
    (Trainer.sync_replicas), and the full-precision phase's losses (no noise drawn) equal to the replicated run's row exchange to the
    printed digits.  (Bit equality of the exchange itself is tests/test_dist_cpu.py's; two runs of the SAME configuration differ in
    the last bits of the MLP gradients here, so the runs' parameters are not compared bit for bit.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29400 + os.getpid() % 150
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "_dp_phases_worker.py")]
    full = {}
    for tag, extra in (("zown", {"GSVC_DP_ZOWN": "1", "GSVC_DP_ZOWN_CHECK": "1"}), ("rows", {"GSVC_DP_SPARSE": "1"})):
        out = subprocess.run(cmd, cwd=root, env=dict(os.environ, GSVC_DIST_BACKEND="gloo", GSVC_SHARE_GPU="1", GSVC_DP_PHASE_STEPS="3", **extra),
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "DP_PHASES_OK" in out.stdout, (tag, out.stdout[-1500:], out.stderr[-2500:])
        lines = [l for l in out.stdout.splitlines() if l.startswith("DP_PHASE ")]
        assert len(lines) == 4 and all("replicas_identical True" in l and f"zown {tag == 'zown'}" in l for l in lines), lines
        full[tag] = lines[0].split("losses ")[1].split("]")[0]
    assert full["zown"] == full["rows"], full


@pytest.mark.gpu
def test_two_ranks_keep_identical_anchors_through_densification():
    """Data-parallel densification: statistics summed over ranks + a per-iteration seed for the random thinning keep the
    replicas' anchor sets, parameters and Adam moments identical across adjust_anchor (tests/_dp_densify_worker.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29900 + os.getpid() % 90
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "_dp_densify_worker.py")]
    out = subprocess.run(cmd, cwd=root, env=dict(os.environ), capture_output=True, text=True, timeout=420)
    assert out.returncode == 0 and "DP_DENSIFY_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-2500:])


@pytest.mark.gpu
def test_evaluate_and_checkpoint_round_trip(tmp_path):
    """report.evaluate returns the reference's evaluation quantities; a checkpoint restores a model that renders the same
    frames bit for bit and keeps training (optimizer state included)."""
    from gsvc_amd.ortho_gaussian_renderer import render_frames
    from gsvc_amd.report import evaluate, load_checkpoint, save_checkpoint
    pc, cube, opt, pipe, mp, Trainer = _setup(anchors=4000, H=192, W=256)
    opt.full_precision_training_total = 1000
    opt.start_stat = 0                                   # densification statistics from the first step (checked below)
    pc.training_setup(opt)
    tr = Trainer(pc, cube, opt, pipe, mp)
    for it in range(1, 9):
        tr.step(it)
    bg = torch.zeros(3)
    ev = evaluate(pc, cube, pipe, bg, frame_ids=range(2, 8))
    assert ev["frames"] == 6 and ev["fps"] > 0
    for k in ("l1", "psnr", "ssim", "msssim"):
        assert np.isfinite(ev[k]), (k, ev)
    assert 0.0 <= ev["ssim"] <= 1.0 and 0.0 <= ev["msssim"] <= 1.0 and ev["psnr"] > 0
    path = str(tmp_path / "ck.pt")
    pc.spatial_lr_scale = 2.5
    save_checkpoint(pc, path, iteration=8)
    pc2, cube2, opt2, pipe2, mp2, _ = _setup(anchors=4000, H=192, W=256, seed=5)      # different initial values
    assert load_checkpoint(pc2, path, training_args=opt2) == 8
    frames = [cube.get_dummy_frame(i) for i in (3, 4)]
    a = list(render_frames(frames, pc, pipe, bg))
    b = list(render_frames(frames, pc2, pipe, bg))
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    st1, st2 = pc.optimizer.state[pc._anchor_feat], pc2.optimizer.state[pc2._anchor_feat]
    assert torch.equal(st1["exp_avg"], st2["exp_avg"]) and float(st1["step"]) == float(st2["step"])
    # resuming keeps the densification statistics (training_setup re-zeroes them: they are restored after it) and the
    # learning-rate scale of the per-anchor groups
    assert pc2.spatial_lr_scale == 2.5
    assert float(pc.opacity_accum.abs().sum()) > 0
    for name in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"):
        assert torch.equal(getattr(pc, name), getattr(pc2, name)), name
    opt2.full_precision_training_total = 1000
    out = Trainer(pc2, cube2, opt2, pipe2, mp2).step(9)
    assert np.isfinite(float(out.loss))


@pytest.mark.parametrize("phase", ["full", "ste"])
def test_generation_once_per_frame_equals_generation_per_view(monkeypatch, phase):
    """FULL_PRECISION and STE_ENTROPY draw nothing per render and the two opposite views of a frame share the camera position, so
    their generators see the same rows (reference guassian.py:225-273 evaluates them per view, with equal results): the batched
    step runs features, conditioning and the four networks once per (frame, distinct visible anchor) and hands the rows to the two
    views (gsvc_amd/generate.py _ViewRows; the two sides see all but ~0.2 % of the same anchors).  Against GSVC_NO_VIEW_SHARE=1
    (per view): same loss, image, statistics and gradients at BASELINE.json configs[2] size — the two views' gradients are added
    at another point of the graph."""
    from gsvc_amd.arguments import cfg_20240919
    from gsvc_amd.frame import SyntheticFrameCube
    from gsvc_amd.model import GaussianModel
    from gsvc_amd.train import Trainer
    import gsvc_amd.generate as G
    res = []
    for share in (True, False):
        if share:
            monkeypatch.delenv("GSVC_NO_VIEW_SHARE", raising=False)
        else:
            monkeypatch.setenv("GSVC_NO_VIEW_SHARE", "1")
        mp_, opt, pipe = cfg_20240919()
        cube = SyntheticFrameCube(1080, 1920, 64, seed=1234, device="cuda").materialize()
        mp_.threshold = 8.0 / cube.scale
        opt.full_precision_training_total = opt.quantized_training_total = opt.entropy_constrained_train_total = 0
        opt.ste_entropy_constrained_train_total = 0
        if phase == "full":
            opt.full_precision_training_total = 100
        else:
            opt.ste_entropy_constrained_train_total = 100
        opt.start_stat, opt.pause_densification, opt.iterations = 0, 0, 10
        torch.manual_seed(0)
        np.random.seed(0)
        pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                           mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                           log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device="cuda")
        rng = np.random.default_rng(0)
        lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
        pc.create_from_points(rng.uniform(lim, -lim, (245_000, 3)), spatial_lr_scale=1.0)
        pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
        pc.training_setup(opt)
        from gsvc_amd import _lib
        torch.manual_seed(5)
        tr = Trainer(pc, cube, opt, pipe, mp_)
        tr.step(1, frame_idx=30)                              # (no plan yet: per view either way; it queues the plan of step 2)
        _lib.profile_enable(True)
        try:
            out = tr.step(2)
            prof = _lib.profile_collect()
        finally:
            _lib.profile_enable(False)
        grads = {n: p.grad.clone() for n, p in pc.named_parameters() if p.grad is not None}
        # shared: the four networks' output gradients were formed per (frame, anchor) row from the two views' (one launch each)
        assert prof.get("k_pair_rows_sum", (0, 0.0))[0] == (4 if share else 0), prof.get("k_pair_rows_sum")
        assert prof["k_trunk_fwd"][0] == 1
        trunk_ms = prof["k_trunk_fwd"][1]
        out2 = tr.step(3)
        res.append((float(out.loss), grads, out.image1.clone(), [r.num_rendered for r in out.renders], pc.offset_gradient_accum.clone(),
                    pc.opacity_accum.clone(), float(out2.loss), trunk_ms))
        del pc, cube, out, out2, tr
        torch.cuda.empty_cache()
    (la, ga, ia, na, ofa, oa, l2a, ta), (lb, gb, ib, nb, ofb, ob, l2b, tb) = res
    assert ta < 0.7 * tb, (ta, tb)              # the generators' trunk kernel ran on about half the rows
    assert abs(la - lb) < 1e-5 * max(1.0, abs(lb)) and abs(l2a - l2b) < 1e-4 * max(1.0, abs(l2b)), (la, lb, l2a, l2b)
    assert all(abs(a - b) <= 2e-5 * b for a, b in zip(na, nb)), (na, nb)
    diff = (ia - ib).abs()
    assert diff.max().item() < 5e-3 and (diff > 5e-5).float().mean().item() < 1e-4
    assert torch.allclose(oa, ob, rtol=1e-5, atol=1e-6)
    assert (ofa - ofb).abs().max().item() <= 2e-3 * ofb.abs().max().item()
    assert set(ga) == set(gb)
    for n in ga:
        scale = gb[n].abs().max().item()
        assert (ga[n] - gb[n]).abs().max().item() <= 1e-3 * scale + 1e-12, (n, (ga[n] - gb[n]).abs().max().item(), scale)


@pytest.mark.gpu
def test_host_resident_video_steps_equal_device_resident_steps():
    """gsvc_amd.frame.HostResidentCube (pictures + flow in pinned host memory, uploaded one step ahead on a copy stream — what the
    reference's step does inside its own timer, pipeline/train.py:407-408) feeds the fitting step the same numbers as the
    device-resident cube: the same losses step by step, one upload per step once the prefetch is running."""
    from gsvc_amd.frame import HostResidentCube
    losses = []
    for host in (False, True):
        pc, cube, opt, pipe, mp, Trainer = _setup(anchors=4000, H=96, W=160, T=12, seed=3)
        opt.full_precision_training_total, opt.quantized_training_total = 2, 1
        opt.entropy_constrained_train_total = 100
        pc.training_setup(opt)
        cube.materialize()
        ds = HostResidentCube(cube, "cuda") if host else cube
        tr = Trainer(pc, ds, opt, pipe, mp, seed=5)
        losses.append([float(tr.step(it).loss) for it in range(1, 8)])
        if host:
            assert 7 <= ds.uploads <= 8, ds.uploads          # one per step (the first step's happens at use)
        tr.close()
    # (identical in the phases without float atomics; the entropy phase's scatter-adds order their sums run by run: 1e-7)
    assert losses[0][:3] == losses[1][:3] and np.allclose(losses[0], losses[1], rtol=1e-4, atol=0), losses


@pytest.mark.gpu
def test_host_resident_video_with_adjacent_pairs_and_no_sync(monkeypatch):
    """consecutive steps on ADJACENT pairs (a, a+1), (a+1, a+2), ... — frame a+1 then sits in two slots —
    with the next step's pair prefetched from INSIDE the backward (GSVC_EARLY_PLAN=1) and no host synchronisation between steps.  A
    step must read both its frames from the slot of ITS pair, and the prefetch must never write a slot handed out since the last
    step_done(): the losses equal the device-resident run's, and the cube's own book-keeping never saw a busy slot overwritten."""
    from gsvc_amd import switches
    from gsvc_amd.frame import HostResidentCube
    monkeypatch.setenv("GSVC_EARLY_PLAN", "1")
    switches.reload()

    class Walk:                      # the trainer's frame draw: a walk over adjacent pairs, forwards then backwards
        def __init__(self):
            self.seq = [2, 3, 4, 5, 4, 3, 2, 3, 3, 4, 6, 5]
            self.k = 0

        def randint(self, lo, hi):
            v = self.seq[self.k % len(self.seq)]
            self.k += 1
            return v
    try:
        losses, written_busy = [], 0
        for host in (False, True):
            pc, cube, opt, pipe, mp, Trainer = _setup(anchors=4000, H=96, W=160, T=12, seed=3)
            opt.full_precision_training_total, opt.quantized_training_total = 4, 2
            opt.entropy_constrained_train_total = 100
            pc.training_setup(opt)
            cube.materialize()
            ds = HostResidentCube(cube, "cuda") if host else cube
            if host:
                real_upload = ds._upload

                def checked(slot, idx, _real=real_upload):
                    nonlocal written_busy
                    written_busy += bool(slot["busy"])
                    return _real(slot, idx)
                ds._upload = checked
            tr = Trainer(pc, ds, opt, pipe, mp, seed=5)
            tr.rng = Walk()
            outs = [tr.step(it).loss.detach() for it in range(1, 13)]          # no float(): nothing waits for the GPU between steps
            losses.append([float(x) for x in outs])
            if host:
                assert getattr(tr, "early_steps", 0) > 0, "the early tail (prefetch from inside the backward) never ran"
                assert len(ds._slots) == 2
            tr.close()
        assert written_busy == 0
        # (a wrong or half-written ground-truth picture moves the loss in its second digit; summation order moves the seventh)
        # (and the entropy phase's float atomics, amplified over its six steps, the fifth: seen up to 4e-6 — two orders below the bar)
        assert np.allclose(losses[0], losses[1], rtol=2e-4, atol=0), losses
    finally:
        monkeypatch.delenv("GSVC_EARLY_PLAN")
        switches.reload()


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["replicated", "zown"])
def test_four_rank_dry_run_at_the_configs3_shape(exchange):
    """BASELINE.json configs[3] rehearsed on one GPU: four data-parallel ranks (gloo, device 0: the pool's process guard allows 6
    on a card, not the 8 of the real run) step the reference's own configuration through 20 iterations with one densification and
    one step that every rank repeats because a single rank overflowed; replicas identical at the end; rank 0 logs the exchange
    (tests/_dp_dryrun_worker.py).  What RCCL will see for the first time on an 8-GPU node has at least run under gloo in this form."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29250 + os.getpid() % 100
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "_dp_dryrun_worker.py")]
    # "zown": the same rehearsal with the per-anchor tensors owned by z-range (GSVC_DP_ZOWN=1): the densification makes the replicas
    # whole first, the row lists follow the new anchors, the repeated step repeats its exchanges on every rank
    env = dict(os.environ, GSVC_DIST_BACKEND="gloo", GSVC_SHARE_GPU="1", **({"GSVC_DP_ZOWN": "1", "GSVC_DP_ZOWN_CHECK": "1"} if exchange == "zown" else {}))
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "DP_DRYRUN_OK ranks=4" in out.stdout, (out.stdout[-2500:], out.stderr[-3000:])
    steps = [l for l in out.stdout.splitlines() if l.startswith("DRYRUN step")]
    assert len(steps) == 20 and "ranks = 4" in out.stdout + out.stderr
    assert all(("to their z-range owners" in l) == (exchange == "zown") for l in steps), steps[:2]
    print("\n".join(steps[:3] + steps[-3:]))


@pytest.mark.gpu
def test_views_without_a_visible_anchor():
    """Edge of the sliding window: a frame whose z-slab holds no anchor renders the background — through the per-render path and
    through the batched pass beside a populated view, in a phase without and a phase with the entropy context — instead of failing
    on empty tensors."""
    from gsvc_amd.generate import GenerateMode
    from gsvc_amd.ortho_gaussian_renderer import render, render_many
    from gsvc_amd.rasterizer import resolve_deferred
    pc, cube, opt, pipe, mp, Trainer = _setup(anchors=3000, T=12)
    with torch.no_grad():
        keep = pc._anchor[:, 2] < (6.5 - 12 / 2) / cube.scale - mp.threshold        # nothing in the slabs of frames 7 ..
        for n in ("_anchor", "_offset", "_mask", "_anchor_feat", "_scaling", "_rotation", "_opacity"):
            setattr(pc, n, torch.nn.Parameter(getattr(pc, n)[keep].clone(), requires_grad=getattr(pc, n).requires_grad))
    opt.full_precision_training_total, opt.quantized_training_total, opt.entropy_constrained_train_total = 1, 1, 100
    opt.start_stat = 0
    pc.training_setup(opt)
    bg = torch.tensor([0.25, 0.5, 0.75])
    empty_fr, full_fr = cube.get_dummy_frame(10), cube.get_dummy_frame(2)
    for mode in (GenerateMode.TRAINING_FULL_PRECISION, GenerateMode.TRAINING_ENTROPY):
        with torch.no_grad():
            r = render(empty_fr, pc, pipe, bg, mode=mode)
            assert int(r.visible_mask.sum()) == 0 and int(r.num_rendered) == 0
            assert torch.allclose(r.rendered_image, bg.cuda().view(3, 1, 1).expand_as(r.rendered_image))
            many = render_many([full_fr, empty_fr], pc, pipe, bg, mode=mode, dense=True, anchor_grad=False)
            _, overflowed = resolve_deferred([m.raster_state for m in many])
            assert not overflowed and int(many[0].visible_mask.sum()) > 0 and int(many[1].visible_mask.sum()) == 0
            assert torch.allclose(many[1].rendered_image, bg.cuda().view(3, 1, 1).expand_as(many[1].rendered_image))
    # the decoder's loop over a run of frames that starts and ends outside the anchors' slabs
    from gsvc_amd.ortho_gaussian_renderer import render_frames, render_pair
    with torch.no_grad():
        p = render_pair(empty_fr, pc, pipe, bg, mode=GenerateMode.DECODING_AS_IS).rendered_image
        assert torch.allclose(p, bg.cuda().view(3, 1, 1).expand_as(p))
        imgs = list(render_frames([empty_fr, full_fr, cube.get_dummy_frame(11)], pc, pipe, bg))
        assert len(imgs) == 3 and torch.allclose(imgs[0], bg.cuda().view(3, 1, 1).expand_as(imgs[0])) and torch.equal(imgs[0], imgs[2])
        assert not torch.allclose(imgs[1], imgs[0])
    # (the LOSS of a step with an empty view is NaN in the reference too — its regularisers and rates are means over empty
    # selections, pipeline/train.py:415-436, guassian.py:110-132 — and is not defined here either: GSVC's anchors cover every frame)


@pytest.mark.gpu
def test_deterministic_mode_gives_the_same_fit_bit_for_bit(monkeypatch):
    """GSVC_DETERMINISTIC=1 (SURVEY section 5 "deterministic-mode switch for bwd atomics"): every float sum
    of a fitting step in a fixed order — sorted row scatters (csrc/generate.hip k_segment_rows_sum) instead of float atomics, one
    workgroup per hash-table slice, no wall-clock measurement picking a launch form.  Two fits of 24 steps through all four phases
    (densification statistics on) from the same seeds end in the SAME parameters and accumulators, bit for bit; the default mode's
    fit stays within rounding of it early on (the same arithmetic up to summation order)."""
    from gsvc_amd import switches

    def fit(det):
        if det:
            monkeypatch.setenv("GSVC_DETERMINISTIC", "1")
        else:
            monkeypatch.delenv("GSVC_DETERMINISTIC", raising=False)
        switches.reload()
        try:
            pc, cube, opt, pipe, mp, Trainer = _setup(anchors=5000, H=96, W=160, T=12, seed=11)
            opt.full_precision_training_total, opt.quantized_training_total = 6, 6
            opt.entropy_constrained_train_total, opt.ste_entropy_constrained_train_total = 6, 6
            opt.start_stat, opt.update_until, opt.pause_densification = 0, 10 ** 9, 0
            pc.training_setup(opt)
            tr = Trainer(pc, cube, opt, pipe, mp, seed=3)
            losses = [tr.step(it).loss.detach() for it in range(1, 25)]
            torch.cuda.synchronize()
            state = {n: p.detach().clone() for n, p in pc.named_parameters()}
            state.update({n: getattr(pc, n).clone() for n in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom")})
            tr.close()
            return [float(x) for x in losses], state
        finally:
            monkeypatch.delenv("GSVC_DETERMINISTIC", raising=False)
            switches.reload()
    l1, s1 = fit(True)
    l2, s2 = fit(True)
    assert l1 == l2
    differing = [n for n in s1 if not torch.equal(s1[n], s2[n])]
    assert not differing, differing
    l0, _ = fit(False)
    assert np.allclose(l0[:8], l1[:8], rtol=2e-5, atol=0), (l0[:8], l1[:8])


@pytest.mark.gpu
def test_deterministic_mode_computes_the_default_modes_gradients(monkeypatch):
    """The fixed-order forms behind GSVC_DETERMINISTIC=1 (sorted row scatters, one workgroup per hash-table slice, per-wave rate sums)
    are the SAME sums in another order: one step per phase from the same parameters, seeds and frame in both modes gives every
    parameter gradient, the loss and the densification accumulators equal to rounding (tools/ab/det_vs_default_grads.py on the
    headline shape: 3e-7 of each tensor's scale)."""
    import os
    from gsvc_amd import switches
    pc, cube, opt, pipe, mp, Trainer = _setup(anchors=5000, H=96, W=160, T=12, seed=5)
    B = 10 ** 9
    opt.full_precision_training_total, opt.quantized_training_total = 0, 0
    opt.entropy_constrained_train_total, opt.ste_entropy_constrained_train_total = B, 0
    opt.start_stat, opt.update_until, opt.pause_densification, opt.update_from = 0, B, 0, B
    pc.training_setup(opt)
    tr = Trainer(pc, cube, opt, pipe, mp, seed=2)
    for it in range(1, 13):
        tr.step(it)
    captured = {}

    def capture(*a, **k):
        if k.get("only") is not None:
            return None
        captured.clear()
        captured.update({n: p.grad.detach().clone() for n, p in pc.named_parameters() if p.grad is not None})
    pc.optimizer.step = capture
    accs = ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom")
    try:
        for phase, totals in (("full", (B, 0, 0, 0)), ("quantized", (0, B, 0, 0)), ("entropy", (0, 0, B, 0)), ("ste", (0, 0, 0, B))):
            (opt.full_precision_training_total, opt.quantized_training_total, opt.entropy_constrained_train_total,
             opt.ste_entropy_constrained_train_total) = totals
            res = {}
            for det in (False, True):
                if det:
                    monkeypatch.setenv("GSVC_DETERMINISTIC", "1")
                else:
                    monkeypatch.delenv("GSVC_DETERMINISTIC", raising=False)
                switches.reload()
                for n in accs:
                    getattr(pc, n).zero_()
                tr._plan = tr._plan_idx = None
                tr.rng.seed(7)
                torch.manual_seed(99)
                tr.controller.current_iteration = 100
                tr.controller._entropy_constrained = False      # sticky, as the reference's; the warm-up ran in the entropy phase
                out = tr.step(100, frame_idx=5)
                torch.cuda.synchronize()
                res[det] = (dict(captured), float(out.loss), {n: getattr(pc, n).clone().float() for n in accs})
            (g0, l0, a0), (g1, l1, a1) = res[False], res[True]
            assert sorted(g0) == sorted(g1), phase
            assert abs(l0 - l1) <= 2e-6 * abs(l0), (phase, l0, l1)
            for n in g0:
                scale = float(g0[n].abs().max())
                assert float((g0[n] - g1[n]).abs().max()) <= 2e-5 * scale + 1e-30, (phase, n)
            for n in accs:
                assert float((a0[n] - a1[n]).abs().max()) <= 2e-5 * float(a0[n].abs().max()) + 1e-30, (phase, n)
    finally:
        os.environ.pop("GSVC_DETERMINISTIC", None)
        switches.reload()
        tr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("batched", [True, False])
def test_weight_gradients_on_their_own_stream_are_the_same_gradients(monkeypatch, batched):
    """Inside a step's backward the generators' / mlp_deform's weight-gradient products run on a side stream behind the chain kernels
    (gsvc_set_wgrad_stream; mlp.wgrad_overlap) while the step's stream carries the feature gradient on.  The same kernels on the same
    operands: under GSVC_DETERMINISTIC=1 every parameter's gradient has the SAME BITS with and without the side stream — in the
    production form (one generation pass per step) and in the per-render form (four passes: only the first one's products leave the
    step's stream, the others' are added to them by autograd on it), in the entropy and the straight-through phase, step after step."""
    import os
    from gsvc_amd import switches
    monkeypatch.setenv("GSVC_DETERMINISTIC", "1")
    switches.reload()
    try:
        res = {}
        for off in (False, True):
            if off:
                monkeypatch.setenv("GSVC_NO_WGRAD_OVERLAP", "1")
            else:
                monkeypatch.delenv("GSVC_NO_WGRAD_OVERLAP", raising=False)
            switches.reload()
            pc, cube, opt, pipe, mp, Trainer = _setup(anchors=5000, H=96, W=160, T=12, seed=7)
            opt.full_precision_training_total, opt.quantized_training_total = 2, 2
            opt.entropy_constrained_train_total, opt.ste_entropy_constrained_train_total = 3, 3
            opt.start_stat, opt.update_until, opt.pause_densification = 10 ** 9, 10 ** 9, 0
            pc.training_setup(opt)
            tr = Trainer(pc, cube, opt, pipe, mp, seed=4, batched=batched)
            grads = []
            real_step = pc.optimizer.step

            def step(*a, **k):
                if k.get("only") is None:
                    grads.append({n: p.grad.detach().clone() for n, p in pc.named_parameters() if p.grad is not None})
                return real_step(*a, **k)
            pc.optimizer.step = step
            for it in range(1, 11):
                tr.step(it, frame_idx=3 + it % 4)
            torch.cuda.synchronize()
            res[off] = (grads, {n: p.detach().clone() for n, p in pc.named_parameters()})
            tr.close()
        (g_on, p_on), (g_off, p_off) = res[False], res[True]
        assert len(g_on) == len(g_off) == 10
        for i, (a, b) in enumerate(zip(g_on, g_off)):
            assert sorted(a) == sorted(b)
            differing = [n for n in a if not torch.equal(a[n], b[n])]
            assert not differing, (i, differing[:5])
        assert not [n for n in p_on if not torch.equal(p_on[n], p_off[n])]
        assert any("mlp_opacity" in n or "mlp_cov" in n for n in g_on[0])
    finally:
        os.environ.pop("GSVC_DETERMINISTIC", None)
        os.environ.pop("GSVC_NO_WGRAD_OVERLAP", None)
        switches.reload()


@pytest.mark.gpu
def test_sorted_row_scatter_equals_index_add():
    """gsvc_segment_rows_sum (the deterministic mode's scatter-add of rows): the values of index_add_ to rounding, the same bits on
    every call, rows without a target untouched (accumulate) or zero (fresh), an empty list a no-op."""
    from gsvc_amd.generate import det_scatter_rows
    g = torch.Generator(device="cuda").manual_seed(2)
    idx = torch.randint(0, 700, (20000,), device="cuda", generator=g)
    src = torch.randn(20000, 13, device="cuda", generator=g)
    a = det_scatter_rows(idx, src, 1000)
    ref = torch.zeros(1000, 13, device="cuda", dtype=torch.float64).index_add_(0, idx, src.double())
    assert float((a.double() - ref).abs().max()) < 1e-4 and float(a[700:].abs().max()) == 0.0
    assert all(torch.equal(a, det_scatter_rows(idx, src, 1000)) for _ in range(3))
    base = torch.randn(1000, 13, device="cuda", generator=g)
    b = det_scatter_rows(idx, src, 1000, out=base.clone())
    assert torch.equal(b[700:], base[700:]) and float((b.double() - (base.double() + ref)).abs().max()) < 1e-4
    e = det_scatter_rows(idx[:0], src[:0], 10)
    assert e.shape == (10, 13) and float(e.abs().max()) == 0.0
