#!/usr/bin/env python3
"""Generate the golden vectors of tests/golden/ by running the REFERENCE's own Python on PyTorch-CPU.

Runs only in the build container (needs /root/reference; see _ref_import.py for the stubs).  The reference's
`_gridencoder` slot is filled with oracle/grid_oracle.c, so fixtures that involve the hash grid pin the
Python around the native kernel (offset tables, permutes, STE, MLPs, rate) on top of the restated kernel.
Outputs are inputs + expected outputs only (no reference source text).  Usage:  python tests/golden/make_golden.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import _ref_import  # noqa: E402

mode = _ref_import.install()


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrays.items()})
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


class RandomTape:
    """Records every tensor produced by torch.rand_like / Tensor.uniform_ so GPU tests can replay them."""

    def __init__(self):
        self.tape = []
        self._rand_like, self._uniform = torch.rand_like, torch.Tensor.uniform_

    def __enter__(self):
        tape = self.tape
        depth = [0]   # the TorchFunctionMode re-enters the patched python functions: record the outermost call only

        def rand_like(x, *a, **k):
            depth[0] += 1
            try:
                r = self._rand_like(x, *a, **k)
            finally:
                depth[0] -= 1
            if depth[0] == 0:
                tape.append(r.detach().clone().numpy())
            return r

        def uniform_(t, *a, **k):
            depth[0] += 1
            try:
                r = self._uniform(t, *a, **k)
            finally:
                depth[0] -= 1
            if depth[0] == 0:
                tape.append(r.detach().clone().numpy())
            return r

        torch.rand_like = rand_like
        torch.Tensor.uniform_ = uniform_
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.Tensor.uniform_ = self._rand_like, self._uniform


with mode:
    import arguments as A
    import utils.encodings as E
    import utils.entropy_models as EM
    import utils.general_utils as GU
    import utils.loss_utils as LU
    import utils.time_util as TU
    import utils.train_util as TT
    import scene.gaussian_model as GM
    import ortho_gaussian_renderer.guassian as G
    from common.base import RenderResults

    # ---------------------------------------------------------------- G1 rate (EntropyGaussian + Low_bound)
    g = torch.Generator().manual_seed(11)
    n, c = 40, 12
    x = (torch.randn(n, c, generator=g) * 3).requires_grad_(True)
    mean = torch.randn(n, c, generator=g).requires_grad_(True)
    scale = (torch.rand(n, c, generator=g) * 2 + 0.05).requires_grad_(True)
    with torch.no_grad():
        x[0, :4] += 60.0       # likelihood far below 2^-16 -> Low_bound active
        scale[1, :4] = 1e-3
        x[2, 0] = 1e6          # outside the +-15000 Q clamp
    Q = (torch.rand(n, 1, generator=g) * 0.5 + 0.05).requires_grad_(True)
    x_mean = torch.tensor(0.123)
    eg = EM.EntropyGaussian(Q=1)
    bits = eg(x, mean, scale, Q, x_mean)
    gout = torch.randn(n, c, generator=g)
    (bits * gout).sum().backward()
    bits_s = eg(x.detach(), mean.detach(), scale.detach(), 0.2, None)   # python-float Q, x_mean from x
    save("rate_entropy_gaussian", x=x, mean=mean, scale=scale, Q=Q, x_mean=x_mean, bits=bits, gout=gout,
         dx=x.grad, dmean=mean.grad, dscale=scale.grad, dQ=Q.grad, bits_scalar_q=bits_s)

    # ---------------------------------------------------------------- G2 quantisers
    g = torch.Generator().manual_seed(12)
    xq = torch.randn(30, 7, generator=g) * 5
    Qrow = torch.rand(30, 1, generator=g) * 0.3 + 0.01
    ste_t = E.STE_multistep.apply(xq, Qrow, xq.mean())
    ste_s = E.STE_multistep.apply(xq, 0.2)
    ste_q = E.STE_multistep.quantize(xq, Qrow, -20, 20)
    xb = (torch.randn(50, 4, generator=g) * 1.5).requires_grad_(True)
    yb = E.STE_binary.apply(xb)
    gb = torch.randn(50, 4, generator=g)
    (yb * gb).sum().backward()
    torch.manual_seed(1234)
    with RandomTape() as tape:
        uq = E.UniformQuantizer()(xq, Qrow, xq.mean())
    anchors = torch.randn(64, 3, generator=g) * 0.4
    mn, mx = torch.tensor([[-1.1, -0.62, -0.34]]), torch.tensor([[1.1, 0.62, 0.34]])
    aq, qv = E.Quantize_anchor.apply(anchors, mn, mx)
    save("quantizers", x=xq, Qrow=Qrow, ste_tensorQ=ste_t, ste_scalarQ=ste_s, ste_quantize=ste_q,
         xb=xb, yb=yb, gb=gb, dxb=xb.grad, uq=uq, uq_noise=tape.tape[0], anchors=anchors, bound_min=mn, bound_max=mx,
         anchors_q=aq, anchors_sym=qv)

    # ---------------------------------------------------------------- G3 hash-table bit count
    tbl = (torch.rand(4000, 8, generator=g) > 0.3).float()
    p, bitsv, mb, tot = E.get_binary_vxl_size(tbl)
    save("binary_vxl_size", table=tbl, p=p, bits=bitsv, mb=np.float64(mb), total=np.int64(tot))

    # ---------------------------------------------------------------- G4 embedder
    emb, dim = TU.get_embedder(16, 1)
    z = torch.linspace(-0.4, 0.4, 64).view(-1, 1)
    save("embedder", z=z, out=emb(z), dim=np.int64(dim))

    # ---------------------------------------------------------------- G8 grid offsets at the shipped config
    e3 = E.GridEncoder(num_dim=3, n_features=8, resolutions_list=(18, 24, 33, 44, 59, 80, 108, 148, 201, 275, 376, 514), log2_hashmap_size=13)
    e2 = E.GridEncoder(num_dim=2, n_features=8, resolutions_list=(130, 258, 514, 1026), log2_hashmap_size=15)
    save("grid_offsets", off3=e3.offsets_list, res3=e3.resolutions_list, off2=e2.offsets_list, res2=e2.resolutions_list)

    # ---------------------------------------------------------------- grid encoder through the reference's autograd wrapper
    for D, res, log2, Cf, tag in ((3, (6, 9, 14, 20), 9, 4, "3d"), (2, (10, 18, 34), 8, 8, "2d")):
        torch.manual_seed(21 + D)
        enc = E.GridEncoder(num_dim=D, n_features=Cf, resolutions_list=res, log2_hashmap_size=log2)
        enc.params.data.uniform_(-1.5, 1.5)
        xin = torch.rand(300, D)
        xin[0] = 0.0
        xin[1] = 1.0
        xin[2, 0] = 1.5          # out of range -> zeros
        xin[3] = 0.5
        xin.requires_grad_(True)
        out = enc(xin)
        go = torch.randn_like(out)
        (out * go).sum().backward()
        save(f"grid_encoder_{tag}", params=enc.params, offsets=enc.offsets_list, resolutions=enc.resolutions_list,
             x=xin, out=out, gout=go, dparams=enc.params.grad, dx=xin.grad)

    # ---------------------------------------------------------------- G9 schedules
    opt = A.OptimizationParams()
    ctl = TT.TrainingController(opt)
    its = [1, 499, 500, 501, 1500, 1600, 9999, 10000, 10001, 10999, 11000, 11001, 11100, 15000, 15001, 24900, 25000,
           25001, 34999, 35000, 35001, 40000, 40001]
    modes, stat, adj, clean = [], [], [], []
    for it in its:
        ctl.current_iteration = it
        m = ctl.render_mode
        modes.append(-1 if m is None else m.value)
        stat.append(ctl.gaussian_statis)
        adj.append(ctl.gaussian_adjust_anchor)
        clean.append(ctl.clean_denorm)
    f1 = GU.get_expon_lr_func(lr_init=0.005, lr_final=0.00001, lr_delay_mult=0.33, max_steps=40000)
    f2 = GU.get_expon_lr_func(lr_init=0.01, lr_final=0.0001, lr_delay_mult=0.01, max_steps=40000)
    f3 = GU.get_expon_lr_func(lr_init=0.0, lr_final=0.0, max_steps=40000)
    steps = np.array([0, 1, 100, 5000, 20000, 39999, 40000, 50000])
    save("schedules", its=np.array(its), modes=np.array(modes), stat=np.array(stat), adj=np.array(adj), clean=np.array(clean),
         steps=steps, lr1=np.array([f1(s) for s in steps]), lr2=np.array([f2(s) for s in steps]),
         lr3=np.array([f3(s) for s in steps]))

    # ---------------------------------------------------------------- G10 image losses
    g = torch.Generator().manual_seed(31)
    i1 = torch.rand(3, 48, 64, generator=g)
    i2 = (i1 + 0.1 * torch.randn(3, 48, 64, generator=g)).clamp(0, 1)
    save("image_losses", img1=i1, img2=i2, l1=LU.l1_loss_func(i1, i2), ssim=LU.ssim_func(i1, i2),
         ssim_per=LU.ssim_func(i1.unsqueeze(0), i2.unsqueeze(0), size_average=False))

    # ---------------------------------------------------------------- G5-G7, G11, G12: tiny model
    mp = A.ModelParams()
    mp.threshold = 0.08
    torch.manual_seed(41)
    ref = GM.GaussianModel(mp, feat_dim=8, n_offsets=4, voxel_size=0.001, update_depth=3, update_init_factor=16,
                           update_hierachy_factor=4, use_feat_bank=False, n_features_per_level=2, log2_hashmap_size=9,
                           log2_hashmap_size_2D=11, resolutions_list=(18, 24, 33), resolutions_list_2D=(130, 258))
    Aa, K, F = 160, 4, 8
    g = torch.Generator().manual_seed(42)
    xl, yl, zl = -1.0, -0.5625, -0.3125
    ref.update_anchor_bound(xl, yl, zl)
    lim = torch.tensor([[-xl, -yl, -zl]])
    import torch.nn as nn
    ref._anchor = nn.Parameter((torch.rand(Aa, 3, generator=g) * 2 - 1) * lim)
    ref._offset = nn.Parameter(torch.randn(Aa, K, 3, generator=g) * 0.5)
    ref._mask = nn.Parameter(torch.randn(Aa, K, 1, generator=g) * 3)
    ref._anchor_feat = nn.Parameter(torch.randn(Aa, F, generator=g))
    ref._scaling = nn.Parameter(torch.randn(Aa, 6, generator=g) * 0.3 - 4.0)
    rots = torch.zeros(Aa, 4)
    rots[:, 0] = 1
    ref._rotation = nn.Parameter(rots, requires_grad=False)
    ref._opacity = nn.Parameter(torch.zeros(Aa, 1), requires_grad=False)
    for enc in (ref.encoding_xyz.encoding_xyz, ref.encoding_xyz.encoding_xy, ref.encoding_xyz.encoding_xz, ref.encoding_xyz.encoding_yz):
        enc.params.data.uniform_(-1.2, 1.2)
    state = {k: v for k, v in ref.state_dict().items()}
    z_cam = 0.05
    frame = SimpleNamespace(cam_pos=torch.tensor([0.0, 0.0, z_cam]))
    visible = (ref.get_anchor[:, 2] - z_cam).abs() < 0.2
    out = {"visible_mask": visible, "z_cam": np.float32(z_cam), "x_lim": np.float32(xl), "y_lim": np.float32(yl), "z_lim": np.float32(zl)}
    for k, v in state.items():
        out["sd::" + k] = v

    # MLP blocks on their own
    feat_in = torch.randn(20, F, generator=g)
    pe_in = torch.randn(20, 66, generator=g)
    out["mlp_feat_in"], out["mlp_pe_in"] = feat_in, pe_in
    out["mlp_opacity_out"] = ref.mlp_opacity(feat_in, pe_in)
    out["mlp_cov_out"] = ref.mlp_cov(feat_in, pe_in)
    out["mlp_color_out"] = ref.mlp_color(feat_in, pe_in)
    out["mlp_deform_out"] = ref.mlp_deform(torch.cat([feat_in, pe_in], dim=1))
    ctx_in = torch.randn(20, ref.encoding_xyz.output_dim, generator=g)
    out["enet_in"] = ctx_in
    for nm in ("mlp_feature_enet", "mlp_scaling_enet", "mlp_offset_enet"):
        a, b, c_ = getattr(ref, nm)(ctx_in)
        out[nm + "_mean"], out[nm + "_scale"], out[nm + "_q"] = a, b, c_
    # getters
    out["get_anchor"], out["get_scaling"], out["get_mask"] = ref.get_anchor, ref.get_scaling, ref.get_mask
    out["get_mask_anchor"] = ref.get_mask_anchor
    out["encoding_params"] = ref.get_encoding_params()
    # entropy context (grid through the oracle backend)
    ec = ref.calc_entropy_context(ref.get_anchor[visible])
    for nm in ("mean_feat", "scale_feat", "mean_scaling", "scale_scaling", "mean_offsets", "scale_offsets",
               "Q_feat_adj", "Q_scaling_adj", "Q_offsets_adj"):
        out["ec::" + nm] = getattr(ec, nm)
    out["interp_feat"] = ref.calc_interp_feat(ref.get_anchor[visible])

    gss_by_mode = {}
    for md in (G.GenerateMode.TRAINING_FULL_PRECISION, G.GenerateMode.TRAINING_QUANTIZED, G.GenerateMode.TRAINING_ENTROPY,
               G.GenerateMode.TRAININ_STE_ENTROPY):
        torch.manual_seed(100 + md.value)
        with RandomTape() as tape:
            gss = G.generate_neural_gaussians(frame, ref, visible, md)
        gss_by_mode[md] = gss
        pre = f"gen{md.value}::"
        for nm in ("xyz", "color", "opacity", "scaling", "rot", "neural_opacity", "mask", "concatenated_all"):
            out[pre + nm] = getattr(gss, nm)
        for nm in ("bit_per_param", "bit_per_feat_param", "bit_per_scaling_param", "bit_per_offsets_param"):
            v = getattr(gss, nm)
            if v is not None:
                out[pre + nm] = v
        for i, t in enumerate(tape.tape):
            out[pre + f"rand{i}"] = t
        out[pre + "n_rand"] = np.int64(len(tape.tape))

    # gradient of a scalar of the FULL_PRECISION output w.r.t. a few parameters (pins the autograd path)
    ref.zero_grad()
    gss = G.generate_neural_gaussians(frame, ref, visible, G.GenerateMode.TRAINING_FULL_PRECISION)
    s = (gss.xyz.sum() + (gss.color ** 2).sum() + gss.opacity.sum() + gss.scaling.sum() * 100 + gss.rot[:, 1].sum())
    s.backward()
    out["grad::_anchor_feat"] = ref._anchor_feat.grad
    out["grad::_offset"] = ref._offset.grad
    out["grad::_scaling"] = ref._scaling.grad
    out["grad::mlp_cov.out_linear.weight"] = ref.mlp_cov.out_linear.weight.grad
    out["grad::mlp_deform.0.weight"] = ref.mlp_deform[0].weight.grad

    # G11 optical loss on two fabricated renders (generation only; no rasterizer involved)
    frame2 = SimpleNamespace(cam_pos=torch.tensor([0.0, 0.0, z_cam + 1.0 / 960]))
    visible2 = (ref.get_anchor[:, 2] - float(frame2.cam_pos[2])).abs() < 0.2
    g1 = G.generate_neural_gaussians(frame, ref, visible, G.GenerateMode.TRAINING_FULL_PRECISION)
    g2 = G.generate_neural_gaussians(frame2, ref, visible2, G.GenerateMode.TRAINING_FULL_PRECISION)
    Hh, Ww, sc = 54, 96, 48.0
    flow = torch.randn(2, Hh, Ww, generator=g)
    rr1 = SimpleNamespace(visible_mask=visible, generated_gaussians=g1)
    rr2 = SimpleNamespace(visible_mask=visible2, generated_gaussians=g2)
    loss, pix, uvp = LU.calc_optical_loss_one_frame(rr1, rr2, flow, -1.0, -0.5625, sc, Ww, Hh, n_offsets=K)
    out["optical::visible2"], out["optical::flow"], out["optical::loss"], out["optical::pix"] = visible2, flow, loss, pix
    out["optical::z2"] = np.float32(float(frame2.cam_pos[2]))

    # G12 training_statis on a fabricated RenderResults
    ta = A.OptimizationParams()
    ref.spatial_lr_scale = 1.0
    ref.training_setup(ta)
    P = int(g1.mask.sum())
    vf = torch.rand(P, generator=g) > 0.3
    vsp = torch.zeros(P, 3)
    vsp.grad = torch.randn(P, 3, generator=g)
    rr = RenderResults(rendered_image=None, viewspace_points=vsp, visible_mask=visible, visibility_filter=vf, radii=None,
                       active_gaussains=0, num_rendered=0, selection_mask=g1.mask, neural_opacity=g1.neural_opacity)
    ref.training_statis(rr)
    ref.training_statis(rr)
    out["statis::visibility_filter"], out["statis::viewspace_grad"] = vf, vsp.grad
    out["statis::opacity_accum"], out["statis::anchor_demon"] = ref.opacity_accum, ref.anchor_demon
    out["statis::offset_gradient_accum"], out["statis::offset_denom"] = ref.offset_gradient_accum, ref.offset_denom
    # optimiser wiring
    out["opt::group_names"] = np.array([gp["name"] for gp in ref.optimizer.param_groups])
    ref.update_learning_rate(12345)
    out["opt::lr_at_12345"] = np.array([gp["lr"] for gp in ref.optimizer.param_groups], dtype=np.float64)
    out["opt::eps"] = np.float64(ref.optimizer.param_groups[0]["eps"])
    save("tiny_model", **out)
print("done")
