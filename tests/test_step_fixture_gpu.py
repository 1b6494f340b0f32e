"""One whole fitting step against the reference's OWN step body (tests/golden/step_fixture.npz, written by
tests/golden/make_golden_step.py, which reads /root/reference/pipeline/train.py:348-462,560-565 at generation time and executes it on
PyTorch-CPU with oracle/ in the native slots): four renders with the view swap, the f/b flip-average, L1 + SSIM + scaling / opacity
regularisers + optical-flow term + lambda * (rates + hash-table bits / denominator) + mask regulariser with the reference's weights
and denominators, ``loss.backward()`` and the four ``training_statis`` calls — in each phase of the schedule (FULL_PRECISION,
QUANTIZED, TRAINING_ENTROPY, STE_ENTROPY) at production dimensions (feat 50, K 10, 192-wide grid feature, 4 783 / 4 629 visible
anchors per frame).  ``gsvc_amd.train.Trainer.step`` must reproduce the loss, the averaged images, every parameter gradient and the
densification accumulators: in the reference-shaped per-render form (``batched=False``, draws in the reference's order as they
come) and in the PRODUCTION form (one dense generation pass for the four views, with and without a step plan), whose batch-wide
draws are composed from the reference's per-render draws (``BatchedDraws``).

Tolerances: loss 2e-5 relative (measured 3e-6); pixels 1e-4 with at most 2e-3 of them beyond (threshold decisions of Gaussians that differ in the
last bits between CPU and GPU MLPs); gradients 1e-3 of the tensor's largest entry / absolute sum; accumulators 1e-3.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
PER_RENDER_DRAWS = {0: 0, 1: 3, 2: 4, 3: 1}      # reference guassian.py:172-221: noise for feat / scaling / offsets, then the 5 % sample


@pytest.fixture(scope="module")
def fix():
    from tests import _prod_model
    g = np.load(os.path.join(HERE, "golden", "step_fixture.npz"))
    pc, mp, fn = _prod_model.build(g)
    return pc, mp, fn, g


class _Dataset:
    """The two adjacent frames of the fixture with the seeded pictures and flow, addressed like FrameCubeDataset."""

    def __init__(self, device="cuda"):
        from tests.golden import seeded
        from gsvc_amd.frame import Frame
        sc = seeded.SCENE
        H, W, T, i0 = sc["H"], sc["W"], sc["T"], sc["frame"]
        self.height, self.width, self.len_z_frames = H, W, T
        self._frames = {}
        for i in (i0, i0 + 1):
            fn = seeded.frame_numbers(H, W, T, i)
            self._frames[i] = Frame(image_id=i, plane="xy", image=seeded.gt_image(H, W, i, sc["seed"]).permute(0, 2, 1).contiguous().to(device),
                                    x_min=fn["x_min"], y_min=fn["y_min"], z=fn["z"], image_width=W, image_height=H,
                                    view_matrix=fn["view_matrix"], view_matrix_s=fn["view_matrix_s"], scale=fn["scale"], cam_pos=fn["cam_pos"])
        fn = seeded.frame_numbers(H, W, T, i0)
        self.x_min, self.y_min, self.z_min, self.scale = fn["x_min"], fn["y_min"], fn["z_min"], fn["scale"]
        self._flow = seeded.optical_flow(H, W, i0, sc["seed"]).to(device)

    def __getitem__(self, i):
        import copy
        return copy.copy(self._frames[i])

    def get_optical_flow(self, i):
        return self._flow


class BatchedDraws:
    """The reference draws per render, in the order (feat noise, scaling noise, offsets noise, rate sample) x (1f, 1b, 2f, 2b):
    draw i = generator(seed + i) (tests/golden/seeded.SeededDraws).  The production step draws once per batch of the four renders'
    concatenated rows; this context answers each batch-wide draw with the per-render draws side by side, so that the batched step
    sees the reference's noise: the k-th ``uniform_`` of a [rows, ...] tensor is the concatenation over renders r of draw
    ``per_render * r + k``; the rate sample (``rand_like`` of a [rows] tensor, or — with a step plan — ``torch.rand(R, A)`` in anchor
    space) is draw ``per_render * r + per_render - 1`` laid out over render r's visible anchors."""

    def __init__(self, seed, counts, per_render, visible):
        self.seed, self.counts, self.per, self.visible = seed, counts, per_render, visible
        self.n_uniform = self.n_sample = 0
        self._saved = (torch.rand_like, torch.Tensor.uniform_, torch.rand)

    def _draw(self, i, n):
        return torch.rand(n, generator=torch.Generator().manual_seed(self.seed + i))

    def _rows(self, k, like):
        inner = like[0].numel() if like.dim() > 1 else 1
        assert like.shape[0] == sum(self.counts), (like.shape, self.counts)
        parts = [self._draw(self.per * r + k, v * inner) for r, v in enumerate(self.counts)]
        return torch.cat(parts).view(like.shape).to(like.device)

    def __enter__(self):
        me = self
        rand_like0, uniform0, rand0 = self._saved

        def rand_like(x, *a, **k):
            me.n_sample += 1
            return me._rows(me.per - 1, x)

        def uniform_(t, lo=0.0, hi=1.0, **k):
            u = me._rows(me.n_uniform, t)
            me.n_uniform += 1
            return t.copy_(u * (hi - lo) + lo)

        def rand(*size, **k):
            R, A = len(me.counts), me.visible[0].shape[0]
            if k.get("generator") is not None or tuple(size) != (R, A):
                return rand0(*size, **k)
            me.n_sample += 1
            u = torch.full((R, A), 2.0)
            for r in range(R):
                u[r, me.visible[r]] = me._draw(me.per * r + me.per - 1, me.counts[r])
            return u.to(k.get("device", "cpu"))

        torch.rand_like, torch.Tensor.uniform_, torch.rand = rand_like, uniform_, rand
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.Tensor.uniform_, torch.rand = self._saved


def _bits(packed, n):
    return np.unpackbits(packed)[:n].astype(bool)


def _run_step(fix, mode_value, batched, planned=False, white=False):
    from tests.golden import seeded
    from gsvc_amd.arguments import OptimizationParams, PipelineParams
    from gsvc_amd.generate import GenerateMode
    from gsvc_amd.ortho_gaussian_renderer import plan_views
    from gsvc_amd.train import Trainer
    pc, mp, fn, g = fix
    sc, ST = seeded.SCENE, seeded.STEP
    pre = f"m{mode_value}w::" if white else f"m{mode_value}::"
    opt = OptimizationParams()
    opt.lmbda, opt.opacity_reg = ST["lmbda"], ST["opacity_reg"]
    assert np.allclose([opt.lambda_dssim, opt.scaling_reg, opt.opacity_reg, opt.optical_lambda, opt.lmbda], g["meta::weights"], rtol=0, atol=0)
    iteration = int(g[pre + "iteration"])
    pc.spatial_lr_scale = 1.0
    pc.training_setup(opt)
    for p in pc.parameters():
        p.grad = None
    grads = {}

    def snapshot(*a, **k):      # in place of Adam: keep the gradients, leave the (module-scoped) parameters where they are
        assert not k.get("only")
        for n, p in pc.named_parameters():
            if p.grad is not None:
                grads[n] = p.grad.detach().clone()
    pc.optimizer.step = snapshot
    import copy
    mp_run = copy.copy(mp)
    mp_run.white_background = bool(white)          # reference pipeline/train.py:327
    tr = Trainer(pc, _Dataset(), opt, PipelineParams(), mp_run, batched=batched, prefetch=False)
    tr.controller.current_iteration = iteration
    assert tr.controller.render_mode == GenerateMode(mode_value)
    A = sc["A"]
    visible = [torch.from_numpy(_bits(row, A)) for row in g[pre + "visible_masks"]]
    counts = [int(v) for v in g[pre + "counts"][:, 0]]
    seed = 5000 + 100 * mode_value
    if batched:
        draws = BatchedDraws(seed, counts, PER_RENDER_DRAWS[mode_value], visible)
    else:
        draws = seeded.SeededDraws(seed)
    with draws:
        if planned:
            with torch.no_grad():
                tr._plan = plan_views(tr._views(sc["frame"]), pc, tr.pipe, tr.background, tr.controller.render_mode)
            tr._plan_idx, tr._plan_mode = sc["frame"], tr.controller.render_mode
        out = tr.step(iteration, frame_idx=sc["frame"])
    if batched:
        per = PER_RENDER_DRAWS[mode_value]
        assert draws.n_uniform == (3 if per >= 3 else 0) and draws.n_sample == (1 if per in (1, 4) else 0), (draws.n_uniform, draws.n_sample)
    else:
        assert draws.count == int(g[pre + "n_draws"])
    assert getattr(tr, "repeated_steps", 0) == 0
    return pc, g, pre, out, grads, tr


def _compare(pc, g, pre, out, grads, mode_value, batched):
    from tests.golden import seeded
    sc = seeded.SCENE
    A, K = sc["A"], pc.n_offsets
    want_loss = float(g[pre + "loss"])
    got_loss = float(out.loss)
    assert abs(got_loss - want_loss) <= 2e-5 * abs(want_loss), (got_loss, want_loss)
    # the four renders: visibility bit for bit, counts within the rounding of generated Gaussians
    for r, (res, row) in enumerate(zip(out.renders, g[pre + "counts"])):
        V, P, active, instances = [int(v) for v in row]
        assert np.array_equal(res.visible_mask.cpu().numpy(), _bits(g[pre + "visible_masks"][r], A))
        assert abs(int(res.active_gaussains) - active) <= max(2, int(1e-3 * active)), (r, int(res.active_gaussains), active)
        assert abs(int(res.num_rendered) - instances) <= max(4, int(1e-3 * instances)), (r, int(res.num_rendered), instances)
        assert int(res.selection_mask.sum()) in range(P - 2, P + 3)
    if mode_value in (2, 3):
        want = g[pre + "rates"]
        for r, res in enumerate(out.renders):
            got = [float(getattr(res, nm)) for nm in ("bit_per_param", "bit_per_feat_param", "bit_per_scaling_param", "bit_per_offsets_param")]
            assert np.allclose(got, want[r], rtol=2e-4, atol=0), (r, got, want[r])
    # the f/b flip-averaged images (reference pipeline/train.py:368-393), every other row
    for nm, img in (("image1", out.image1), ("image2", out.image2)):
        err = np.abs(img.detach().cpu().numpy()[:, ::2] - g[pre + nm])
        assert (err > 1e-4).mean() <= 2e-3 and err.max() < 5e-2, (nm, float((err > 1e-4).mean()), float(err.max()))
    # every parameter gradient of the step
    rs, gs_ = 16, 5
    checked, worst = 0, (0.0, "")
    for name, p in pc.named_parameters():
        key = f"{pre}sum::{name}"
        if batched and name == "_anchor":      # the production step does not differentiate positions (GSVC trains them with lr 0)
            assert name not in grads
            continue
        if key not in g.files or float(g[key][2]) == 0.0:
            assert name not in grads or float(grads[name].abs().max()) == 0.0, name
            continue
        s, sabs, smax = [float(v) for v in g[key]]
        assert name in grads, name
        gr = grads[name]
        e_abs = abs(float(gr.double().abs().sum()) - sabs) / sabs
        e_sum = abs(float(gr.double().sum()) - s) / sabs
        assert e_abs <= 1e-3 and e_sum <= 1e-3, (name, e_abs, e_sum)
        rk = f"{pre}grad::{name}"
        if rk in g.files:
            want = g[rk]
            got = (gr[::rs] if name.startswith("_") else gr[::4] if name.endswith("params") else
                   gr[::gs_] if (gr.dim() == 2 and gr.numel() > 2048) else gr).cpu().numpy()
            assert got.shape == want.shape, name
            e = float(np.abs(got - want).max()) / smax
            worst = max(worst, (e, name))
            assert e <= 1e-3, (name, e, smax)
        checked += 1
    n_keys = sum(1 for k in g.files if k.startswith(pre + "sum::") and float(g[k][2]) > 0.0)
    assert n_keys >= {0: 57, 1: 57, 2: 87, 3: 72}[mode_value] and checked == n_keys - (1 if batched else 0), (checked, n_keys)
    # densification statistics (reference pipeline/train.py:560-565, scene/gaussian_model.py:1281-1314)
    assert bool(g[pre + "gaussian_statis"]) == (mode_value != 3)
    for nm in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"):
        got = getattr(pc, nm)
        s, sabs = [float(v) for v in g[pre + "statis_sum::" + nm]]
        want = g[pre + "statis::" + nm]
        if not bool(g[pre + "gaussian_statis"]):
            assert float(got.abs().sum()) == 0.0 and sabs == 0.0
            continue
        assert sabs > 0
        slack = 8.0 if nm.endswith(("demon", "denom")) else 1e-3 * sabs       # counts: a selection / radius flip moves them by one
        assert abs(float(got.double().sum()) - s) <= slack, (nm, float(got.double().sum()), s)
        d = np.abs(got[::8].cpu().numpy() - want)
        if nm.endswith(("demon", "denom")):
            assert (d > 0).sum() <= 4 and d.max() <= 1, (nm, int((d > 0).sum()))
        else:
            assert d.max() <= 1e-3 * float(np.abs(want).max()) or (d > 1e-3 * float(np.abs(want).max())).sum() <= 4, (nm, float(d.max()))
    return checked, worst


@pytest.mark.parametrize("mode_value", [0, 1, 2, 3])
def test_per_render_step_matches_the_reference_step(fix, mode_value):
    """Trainer.step in the reference's shape: four render() calls, the draws in the reference's order."""
    pc, g, pre, out, grads, _ = _run_step(fix, mode_value, batched=False)
    checked, worst = _compare(pc, g, pre, out, grads, mode_value, batched=False)
    print(f"[m{mode_value} per-render] loss {float(out.loss):.7f} vs {float(g[pre + 'loss']):.7f}; {checked} gradients, worst row error {worst[0]:.2e} ({worst[1]})")


@pytest.mark.parametrize("planned", [False, True])
@pytest.mark.parametrize("mode_value", [0, 1, 2, 3])
def test_production_step_matches_the_reference_step(fix, mode_value, planned):
    """Trainer.step in the production shape — one dense generation pass for the four views (chain kernels, fused quantisers, the
    sampled rate as one launch), rasterize_many, fused SSIM pair, one stacked loss — with and without a step plan."""
    pc, g, pre, out, grads, tr = _run_step(fix, mode_value, batched=True, planned=planned)
    checked, worst = _compare(pc, g, pre, out, grads, mode_value, batched=True)
    print(f"[m{mode_value} production{' planned' if planned else ''}] loss {float(out.loss):.7f} vs {float(g[pre + 'loss']):.7f}; "
          f"{checked} gradients, worst row error {worst[0]:.2e} ({worst[1]})")


@pytest.mark.parametrize("batched", [False, True])
def test_step_on_a_white_background_matches_the_reference_step(fix, batched):
    """The full-precision step once more with ``white_background`` (reference pipeline/train.py:327): the composite C + T bg and
    the background's share of dL/dalpha — the compositing backward's general (non-black) instantiation — inside a whole step."""
    pc, g, pre, out, grads, _ = _run_step(fix, 0, batched=batched, white=True)
    assert pre == "m0w::" and abs(float(g[pre + "loss"]) - float(g["m0::loss"])) > 1e-3       # the background matters in this scene
    checked, worst = _compare(pc, g, pre, out, grads, 0, batched=batched)
    print(f"[m0 white {'production' if batched else 'per-render'}] loss {float(out.loss):.7f} vs {float(g[pre + 'loss']):.7f}; {checked} gradients")
