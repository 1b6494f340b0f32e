"""CPU tests: the host side of gsvc_amd (quantisers, embedder, schedules, losses, MLP blocks, Gaussian
generation, densification statistics, optimiser wiring) and the numpy/C oracles against golden vectors
captured from the reference's own Python (tests/golden/make_golden.py).  float32 tolerances are stated inline.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def T(a):
    return torch.tensor(np.asarray(a))


# ------------------------------------------------------------------------------------------ rate oracle
def test_rate_oracle_matches_reference():
    from oracle import rate_oracle as ro
    g = load("rate_entropy_gaussian")
    bits, _, _, _ = ro.entropy_gaussian_bits(g["x"], g["mean"], g["scale"], g["Q"], float(g["x_mean"]))
    assert np.abs(bits - g["bits"]).max() < 2e-3 * max(1.0, np.abs(g["bits"]).max())  # fp32 erf cancellation in the reference
    finite = g["bits"] < 15.9
    assert np.abs(bits - g["bits"])[finite].max() < 5e-3
    assert (g["bits"] == 16.0).sum() >= 4  # the Low_bound cases are in the fixture
    dx, dmean, dscale, dQ = ro.entropy_gaussian_grads(g["x"], g["mean"], g["scale"], g["Q"], g["gout"], float(g["x_mean"]))
    for ours, ref, nm in ((dx, g["dx"], "dx"), (dmean, g["dmean"], "dmean"), (dscale, g["dscale"], "dscale"), (dQ, g["dQ"], "dQ")):
        scale = np.abs(ref).max()
        assert np.abs(ours - ref).max() < 2e-3 * scale, nm
    # gradient is exactly zero where the likelihood sat below the bound, and for the clamped x
    assert np.all(g["dmean"][g["bits"] == 16.0] == 0) and np.all(dmean[g["bits"] == 16.0] == 0)
    assert g["dx"][2, 0] == 0 and dx[2, 0] == 0
    bits_s, _, _, _ = ro.entropy_gaussian_bits(g["x"], g["mean"], g["scale"], 0.2, None)
    ok = g["bits_scalar_q"] < 15.9
    assert np.abs(bits_s - g["bits_scalar_q"])[ok].max() < 5e-3


# ------------------------------------------------------------------------------------------ quantisers
def test_quantizers_match_reference():
    from gsvc_amd import encodings as E
    g = load("quantizers")
    x, Q = T(g["x"]), T(g["Qrow"])
    assert torch.equal(E.STE_multistep.apply(x, Q, x.mean()), T(g["ste_tensorQ"]))
    assert torch.equal(E.STE_multistep.apply(x, 0.2), T(g["ste_scalarQ"]))
    assert torch.equal(E.STE_multistep.quantize(x, Q, -20, 20), T(g["ste_quantize"]))
    xb = T(g["xb"]).requires_grad_(True)
    yb = E.STE_binary.apply(xb)
    assert torch.equal(yb, T(g["yb"]))
    (yb * T(g["gb"])).sum().backward()
    assert torch.equal(xb.grad, T(g["dxb"]))
    torch.manual_seed(1234)
    assert torch.equal(E.UniformQuantizer()(x, Q, x.mean()), T(g["uq"]))
    aq, sym = E.Quantize_anchor.apply(T(g["anchors"]), T(g["bound_min"]), T(g["bound_max"]))
    assert torch.equal(aq, T(g["anchors_q"])) and torch.equal(sym, T(g["anchors_sym"]))
    q2, interval, mn = E.Quantize_anchor.quantized(T(g["anchors"]), T(g["bound_min"]), T(g["bound_max"]))
    assert torch.equal(E.Quantize_anchor.dequantized(q2, interval, mn), T(g["anchors_q"]))
    # STE_multistep gradient is the identity
    xr = x.clone().requires_grad_(True)
    E.STE_multistep.apply(xr, Q, x.mean()).sum().backward()
    assert torch.equal(xr.grad, torch.ones_like(x))


def test_binary_vxl_size_and_embedder_and_offsets():
    from gsvc_amd import encodings as E
    from gsvc_amd.time_util import get_embedder
    g = load("binary_vxl_size")
    p, bits, mb, tot = E.get_binary_vxl_size(T(g["table"]))
    assert torch.allclose(p, T(g["p"])) and torch.allclose(bits, T(g["bits"]), rtol=1e-6)
    assert abs(mb - float(g["mb"])) < 1e-9 and tot == int(g["total"])
    g = load("embedder")
    emb, dim = get_embedder(16, 1)
    assert dim == int(g["dim"]) == 33
    out = emb(T(g["z"]))
    assert out.shape == g["out"].shape
    assert np.abs(out.numpy() - g["out"]).max() < 2e-6
    g = load("grid_offsets")
    assert E.level_offsets(g["res3"].tolist(), 3, 13) == g["off3"].tolist()
    assert E.level_offsets(g["res2"].tolist(), 2, 15) == g["off2"].tolist()
    enc = E.GridEncoder(num_dim=2, n_features=8, resolutions_list=(130, 258, 514, 1026), log2_hashmap_size=15)
    assert enc.offsets_list.tolist() == g["off2"].tolist() and enc.params.shape == (g["off2"][-1], 8)


def test_schedules_match_reference():
    from gsvc_amd.arguments import OptimizationParams
    from gsvc_amd.model import get_expon_lr_func
    from gsvc_amd.train_util import TrainingController
    g = load("schedules")
    ctl = TrainingController(OptimizationParams())
    for it, m, st, ad, cl in zip(g["its"], g["modes"], g["stat"], g["adj"], g["clean"]):
        ctl.current_iteration = int(it)
        mode = ctl.render_mode
        assert (-1 if mode is None else mode.value) == m, it
        assert ctl.gaussian_statis == bool(st) and ctl.gaussian_adjust_anchor == bool(ad) and ctl.clean_denorm == bool(cl), it
    # switches at 1 / 10001 / 15001 / 35001
    assert ctl.entropy_constrained
    f1 = get_expon_lr_func(lr_init=0.005, lr_final=0.00001, lr_delay_mult=0.33, max_steps=40000)
    f2 = get_expon_lr_func(lr_init=0.01, lr_final=0.0001, lr_delay_mult=0.01, max_steps=40000)
    f3 = get_expon_lr_func(lr_init=0.0, lr_final=0.0, max_steps=40000)
    for s, a, b, c in zip(g["steps"], g["lr1"], g["lr2"], g["lr3"]):
        assert f1(int(s)) == a and f2(int(s)) == b and f3(int(s)) == c


def test_image_losses_match_reference():
    from gsvc_amd import loss_utils as LU
    g = load("image_losses")
    a, b = T(g["img1"]), T(g["img2"])
    assert abs(float(LU.l1_loss_func(a, b)) - float(g["l1"])) < 1e-7
    assert abs(float(LU.ssim_func(a, b)) - float(g["ssim"])) < 2e-6   # separable window vs 11x11 conv, fp32
    per = LU.ssim_func(a.unsqueeze(0), b.unsqueeze(0), size_average=False)
    assert np.abs(per.numpy() - g["ssim_per"]).max() < 2e-6


# ------------------------------------------------------------------------------------------ tiny model
@pytest.fixture(scope="module")
def tiny():
    from gsvc_amd.arguments import ModelParams
    from gsvc_amd.model import GaussianModel
    g = load("tiny_model")
    mp = ModelParams()
    mp.threshold = 0.08
    pc = GaussianModel(mp, feat_dim=8, n_offsets=4, voxel_size=0.001, update_depth=3, update_init_factor=16,
                       update_hierachy_factor=4, use_feat_bank=False, n_features_per_level=2, log2_hashmap_size=9,
                       log2_hashmap_size_2D=11, resolutions_list=(18, 24, 33), resolutions_list_2D=(130, 258), device="cpu")
    sd = {k[4:]: T(g[k]) for k in g.files if k.startswith("sd::")}
    for nm in ("_anchor", "_offset", "_mask", "_anchor_feat", "_scaling", "_rotation", "_opacity"):
        setattr(pc, nm, torch.nn.Parameter(sd[nm].clone(), requires_grad=nm not in ("_rotation", "_opacity")))
    missing, unexpected = pc.load_state_dict(sd, strict=True)
    pc.update_anchor_bound(float(g["x_lim"]), float(g["y_lim"]), float(g["z_lim"]))
    return pc, g


def test_state_dict_keys_equal_reference(tiny):
    pc, g = tiny
    ours = set(pc.state_dict().keys())
    ref = {k[4:] for k in g.files if k.startswith("sd::")}
    assert ours == ref


def test_mlp_blocks_and_getters(tiny):
    pc, g = tiny
    f, pe = T(g["mlp_feat_in"]), T(g["mlp_pe_in"])
    tol = dict(rtol=1e-5, atol=1e-6)
    assert torch.allclose(pc.mlp_opacity(f, pe), T(g["mlp_opacity_out"]), **tol)
    assert torch.allclose(pc.mlp_cov(f, pe), T(g["mlp_cov_out"]), **tol)
    assert torch.allclose(pc.mlp_color(f, pe), T(g["mlp_color_out"]), **tol)
    assert torch.allclose(pc.mlp_deform(torch.cat([f, pe], 1)), T(g["mlp_deform_out"]), **tol)
    ctx = T(g["enet_in"])
    for nm in ("mlp_feature_enet", "mlp_scaling_enet", "mlp_offset_enet"):
        m, s, q = getattr(pc, nm)(ctx)
        assert torch.allclose(m, T(g[nm + "_mean"]), **tol) and torch.allclose(s, T(g[nm + "_scale"]), **tol)
        assert torch.allclose(q, T(g[nm + "_q"]), **tol)
    assert torch.equal(pc.get_anchor, T(g["get_anchor"]))
    assert torch.equal(pc.get_scaling, T(g["get_scaling"]))
    assert torch.equal(pc.get_mask, T(g["get_mask"]))
    assert torch.equal(pc.get_mask_anchor, T(g["get_mask_anchor"]))
    assert torch.equal(pc.get_encoding_params(), T(g["encoding_params"]))


@pytest.mark.parametrize("mode_value,seeded", [(0, False), (1, True)])
def test_generate_neural_gaussians_cpu_modes(tiny, mode_value, seeded):
    from gsvc_amd.generate import GenerateMode, generate_neural_gaussians
    pc, g = tiny
    frame = SimpleNamespace(cam_pos=torch.tensor([0.0, 0.0, float(g["z_cam"])]))
    vis = T(g["visible_mask"])
    torch.manual_seed(100 + mode_value)
    gss = generate_neural_gaussians(frame, pc, vis, GenerateMode(mode_value))
    pre = f"gen{mode_value}::"
    assert torch.equal(gss.mask, T(g[pre + "mask"]))
    for nm in ("xyz", "color", "opacity", "scaling", "rot", "neural_opacity", "concatenated_all"):
        assert torch.allclose(getattr(gss, nm), T(g[pre + nm]), rtol=1e-5, atol=1e-6), nm
    assert gss.bit_per_param is None


def test_generate_gradients_match_reference(tiny):
    from gsvc_amd.generate import GenerateMode, generate_neural_gaussians
    pc, g = tiny
    frame = SimpleNamespace(cam_pos=torch.tensor([0.0, 0.0, float(g["z_cam"])]))
    pc.zero_grad()
    gss = generate_neural_gaussians(frame, pc, T(g["visible_mask"]), GenerateMode.TRAINING_FULL_PRECISION)
    s = (gss.xyz.sum() + (gss.color ** 2).sum() + gss.opacity.sum() + gss.scaling.sum() * 100 + gss.rot[:, 1].sum())
    s.backward()
    for nm, ours in (("_anchor_feat", pc._anchor_feat.grad), ("_offset", pc._offset.grad), ("_scaling", pc._scaling.grad),
                     ("mlp_cov.out_linear.weight", pc.mlp_cov.out_linear.weight.grad),
                     ("mlp_deform.0.weight", pc.mlp_deform[0].weight.grad)):
        ref = T(g["grad::" + nm])
        assert torch.allclose(ours, ref, rtol=1e-4, atol=1e-6 * float(ref.abs().max())), nm


def test_optical_loss_statis_and_optimizer(tiny):
    from gsvc_amd import loss_utils as LU
    from gsvc_amd.arguments import OptimizationParams
    from gsvc_amd.common.base import RenderResults
    from gsvc_amd.generate import GenerateMode, generate_neural_gaussians
    pc, g = tiny
    K = pc.n_offsets
    vis1, vis2 = T(g["visible_mask"]), T(g["optical::visible2"])
    f1 = SimpleNamespace(cam_pos=torch.tensor([0.0, 0.0, float(g["z_cam"])]))
    f2 = SimpleNamespace(cam_pos=torch.tensor([0.0, 0.0, float(g["optical::z2"])]))
    g1 = generate_neural_gaussians(f1, pc, vis1, GenerateMode.TRAINING_FULL_PRECISION)
    g2 = generate_neural_gaussians(f2, pc, vis2, GenerateMode.TRAINING_FULL_PRECISION)
    rr1 = SimpleNamespace(visible_mask=vis1, generated_gaussians=g1)
    rr2 = SimpleNamespace(visible_mask=vis2, generated_gaussians=g2)
    loss, pix, _ = LU.calc_optical_loss_one_frame(rr1, rr2, T(g["optical::flow"]), -1.0, -0.5625, 48.0, 96, 54, n_offsets=K)
    assert torch.equal(pix, T(g["optical::pix"]))
    assert abs(float(loss) - float(g["optical::loss"])) < 1e-6
    # densification statistics
    pc.spatial_lr_scale = 1.0
    pc.training_setup(OptimizationParams())
    vsp = torch.zeros(int(g1.mask.sum()), 3)
    vsp.grad = T(g["statis::viewspace_grad"])
    rr = RenderResults(rendered_image=None, viewspace_points=vsp, visible_mask=vis1,
                       visibility_filter=T(g["statis::visibility_filter"]), radii=None, active_gaussains=0, num_rendered=0,
                       selection_mask=g1.mask, neural_opacity=g1.neural_opacity)
    pc.training_statis(rr)
    pc.training_statis(rr)
    for nm in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"):
        assert torch.allclose(getattr(pc, nm), T(g["statis::" + nm]), rtol=1e-5, atol=1e-6), nm
    # optimiser: 15 groups in the reference's order, Adam eps 1e-15, scheduled learning rates
    assert [gp["name"] for gp in pc.optimizer.param_groups] == g["opt::group_names"].tolist()
    assert pc.optimizer.param_groups[0]["eps"] == float(g["opt::eps"]) == 1e-15
    pc.update_learning_rate(12345)
    assert np.allclose([gp["lr"] for gp in pc.optimizer.param_groups], g["opt::lr_at_12345"], rtol=1e-12, atol=0)


# ------------------------------------------------------------------------------------------ grid oracle vs fixtures
@pytest.mark.parametrize("tag", ["3d", "2d"])
def test_grid_oracle_reproduces_fixture(oracle_lib, tag):
    g = load("grid_encoder_" + tag)
    emb = np.where(g["params"] >= 0, 1.0, -1.0).astype(np.float32)  # STE_binary
    out, dy = oracle_lib.grid_forward(g["x"], emb, g["offsets"], g["resolutions"], calc_dy_dx=True)
    L, N, C = out.shape
    assert np.array_equal(out.transpose(1, 0, 2).reshape(N, L * C), g["out"])
    assert np.all(g["out"][2] == 0)  # out-of-range point
    gout = g["gout"].reshape(N, L, C).transpose(1, 0, 2)
    ge, gi = oracle_lib.grid_backward(gout, g["x"], emb, g["offsets"], g["resolutions"], dy)
    mask = (np.abs(g["params"]) <= 1).astype(np.float32)
    assert np.allclose(ge * mask, g["dparams"], rtol=1e-5, atol=1e-6)
    assert np.allclose(gi, g["dx"], rtol=1e-5, atol=1e-5)
    # finite-difference sanity of dy_dx away from cell boundaries (the oracle's own derivative)
    eps = 1e-3
    xs = np.clip(g["x"][10:40].copy(), 0.05, 0.95)
    o0, d0 = oracle_lib.grid_forward(xs, emb, g["offsets"], g["resolutions"], calc_dy_dx=True)
    D = xs.shape[1]
    assert d0.shape == (xs.shape[0], L * D * C)


def test_ms_ssim_against_independent_numpy():
    """gsvc_amd.metrics.ms_ssim (restatement of pytorch_msssim's algorithm, parity unpinned) against a
    NumPy / SciPy implementation written from the same published description."""
    from scipy.ndimage import correlate1d
    from gsvc_amd.metrics import MS_WEIGHTS, ms_ssim, msssim_fn
    rng = np.random.default_rng(3)
    H, W = 200, 181                                           # odd side: exercises the pooling's padding rule
    a = rng.random((2, 3, H, W))
    b = np.clip(a + rng.normal(0, 0.08, a.shape) + 0.05 * np.sin(np.arange(W) / 7.0), 0, 1)

    g = np.exp(-((np.arange(11) - 5) ** 2) / (2 * 1.5 ** 2))
    g /= g.sum()

    def filt(x):                                              # valid correlation along H then W
        x = correlate1d(x, g, axis=-2, mode="constant")[..., 5:-5, :]
        return correlate1d(x, g, axis=-1, mode="constant")[..., :, 5:-5]

    def pool(x):                                              # 2x2 average, zero padding on odd sides, padded cells counted
        ph, pw = x.shape[-2] % 2, x.shape[-1] % 2
        x = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
        h2, w2 = x.shape[-2] // 2, x.shape[-1] // 2
        return x[..., :2 * h2, :2 * w2].reshape(x.shape[0], x.shape[1], h2, 2, w2, 2).mean((3, 5))

    x, y = a.copy(), b.copy()
    terms = []
    for lvl in range(5):
        mu1, mu2 = filt(x), filt(y)
        s1, s2, s12 = filt(x * x) - mu1 ** 2, filt(y * y) - mu2 ** 2, filt(x * y) - mu1 * mu2
        cs = (2 * s12 + 0.03 ** 2) / (s1 + s2 + 0.03 ** 2)
        ss = (2 * mu1 * mu2 + 0.01 ** 2) / (mu1 ** 2 + mu2 ** 2 + 0.01 ** 2) * cs
        if lvl < 4:
            terms.append(np.maximum(cs.mean((-1, -2)), 0))
            x, y = pool(x), pool(y)
        else:
            terms.append(np.maximum(ss.mean((-1, -2)), 0))
    ref = np.prod(np.stack(terms) ** np.array(MS_WEIGHTS).reshape(-1, 1, 1), axis=0).mean()
    got = ms_ssim(T(a.astype(np.float32)), T(b.astype(np.float32)))
    assert abs(float(got) - ref) < 2e-5, (float(got), ref)
    assert abs(float(msssim_fn(T(a[0].astype(np.float32))[None], T(b[0].astype(np.float32))[None])) -
               np.prod(np.stack(terms)[:, 0] ** np.array(MS_WEIGHTS).reshape(-1, 1), axis=0).mean()) < 2e-5
    assert float(ms_ssim(T(a.astype(np.float32)), T(a.astype(np.float32)))) > 0.99999
    with pytest.raises(ValueError):
        ms_ssim(torch.zeros(1, 3, 100, 300), torch.zeros(1, 3, 100, 300))
