#!/usr/bin/env python3
"""End-to-end fit -> encode -> decode -> evaluate of the synthetic video: one rate-distortion point through every stage the
reference's pipeline has (pipeline/train.py:325-583: the four phases of the schedule with anchor densification; utils/codec_utils.py:
89-108: encode + decode; utils/report_utils.py:268-407: evaluation of the decoded model).

The reference's 40 000-iteration schedule (10 k full precision / 5 k quantised / 20 k entropy-constrained / 5 k straight-through;
statistics from 500, densification 1 500 .. 25 000 every 100, paused 1 000 iterations at the first phase change) is scaled to
``--steps`` iterations with the same proportions.  Reports PSNR / SSIM / MS-SSIM of the decoded video, bits per pixel of the
written streams (anchor geometry + attributes + masks + hash tables + 8-bit MLP file), and checks the two identities the codec is
built on:
  * the decoder reproduces the straight-through model: the decoded model (pruned, reordered, entropy-coded) renders what the fitted
    model renders with its attributes replaced by their straight-through values (same MLPs) — PSNR within 0.01 dB; in the 1080p
    runs of profiles/r05 the SAME PIXELS bit for bit (``decoded_vs_quantised_model_max_abs`` 0.0), at the test's toy size 0.3 % of
    the pixels differ (whole-step flips of isolated attributes: the quantisation steps come from the networks run over all anchors
    here and per z-slab in the codec).  The STE-phase render of the live model is 0.001-0.06 dB away from both (39-47 dB): the
    anchor-level visibility test reads the unquantised scalings while training and the decoded ones afterwards — in the reference
    too (``ste_vs_decoded_one_frame`` counts the anchors that are in one visible set only);
  * the streams are as long as the entropy model says: the attribute streams' coded payload within 2 % of ``estimate_final_bits``
    (payload = stream bytes minus the framing that buys the decoder its parallelism: 36-byte header + 9 bytes per independently
    decodable segment, gsvc_amd/codec.py — at ~0.3 bit per symbol that framing is itself ~5 % and is reported beside it).

usage: python tools/fit_synthetic.py [--steps 2000] [--height 1080 --width 1920 --frames 64 --anchors 100000] [--json out.json]
"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--anchors", type=int, default=100_000)
    ap.add_argument("--lmbda", type=float, default=0.004)
    ap.add_argument("--slab-frames", type=float, default=16.0, help="z-slab of a render in frames (2 x threshold x scale)")
    ap.add_argument("--eval-frames", type=int, default=16)
    ap.add_argument("--densify-grad-threshold", type=float, default=None, help="default: the reference's 5e-4")
    ap.add_argument("--payload-tol", type=float, default=0.02, help="allowed |coded payload / estimate_final_bits - 1| of the attribute streams")
    ap.add_argument("--json", default=None)
    ap.add_argument("--dump-coder-inputs", default=None, help="npz of what the first slabs hand to the entropy coder (symbols, mu, sigma)")
    ap.add_argument("--lpips-weights", default=None, help="backbone (+ --lpips-lin-weights) file for gsvc_amd.lpips.LPIPS")
    ap.add_argument("--lpips-lin-weights", default=None)
    args = ap.parse_args(argv)

    from gsvc_amd.arguments import cfg_20240919
    from gsvc_amd.frame import SyntheticFrameCube
    from gsvc_amd.generate import GenerateMode
    from gsvc_amd.loss_utils import psnr_func
    from gsvc_amd.model import GaussianModel
    from gsvc_amd.ortho_gaussian_renderer import render_pair
    from gsvc_amd.report import evaluate
    from gsvc_amd.stream_codec import conduct_stream_decoding, conduct_stream_encoding
    from gsvc_amd.train import Trainer

    # under torch.distributed.run: data-parallel fit (frames sharded over the ranks, gsvc_amd/dist.py); rank 0 encodes and evaluates.
    # GSVC_DIST_BACKEND=gloo GSVC_SHARE_GPU=1 puts every rank on device 0 (rehearsal on one GPU)
    from gsvc_amd import dist as gdist
    share = bool(os.environ.get("GSVC_SHARE_GPU"))
    local = 0 if share else int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    rank, world, _ = gdist.init_from_env(os.environ.get("GSVC_DIST_BACKEND") or None)
    H, W, T, N = args.height, args.width, args.frames, args.steps
    mp_, opt, pipe = cfg_20240919()
    cube = SyntheticFrameCube(H, W, T, seed=1234, device=dev).materialize()
    mp_.threshold = args.slab_frames / 2.0 / cube.scale
    s = N / 40_000.0
    opt.iterations, opt.lmbda = N, args.lmbda
    opt.full_precision_training_total, opt.quantized_training_total = int(10_000 * s), int(5_000 * s)
    opt.entropy_constrained_train_total = int(20_000 * s)
    opt.ste_entropy_constrained_train_total = N - int(35_000 * s)
    opt.start_stat, opt.update_from, opt.update_until = int(500 * s), int(1_500 * s), int(25_000 * s)
    opt.update_interval = max(20, int(100 * s))
    opt.pause_densification = int(1_000 * s)
    if args.densify_grad_threshold is not None:
        opt.densify_grad_threshold = args.densify_grad_threshold
    for name in dir(opt):                        # the learning-rate schedules decay over the run's length
        if name.endswith("_max_steps"):
            setattr(opt, name, N)
    torch.manual_seed(0)
    np.random.seed(0)
    pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                       mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                       log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device=dev)
    # reference frame_cube/utils.py:6-15 (init_point_cloud, bleed 0.1)
    lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
    pc.create_from_points(np.random.default_rng(0).uniform(lim, -lim, (args.anchors, 3)), spatial_lr_scale=1.0)
    pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
    pc.training_setup(opt)
    gdist.broadcast_parameters(pc)
    trainer = Trainer(pc, cube, opt, pipe, mp_, seed=0)
    bg = trainer.background
    log = {"config": {"H": H, "W": W, "frames": T, "steps": N, "anchors_init": args.anchors, "lmbda": args.lmbda,
                      "slab_frames": args.slab_frames, "schedule": [opt.full_precision_training_total, opt.quantized_training_total,
                                                                    opt.entropy_constrained_train_total, opt.ste_entropy_constrained_train_total],
                      "densify": [opt.start_stat, opt.update_from, opt.update_interval, opt.update_until, opt.pause_densification]},
           "phases": []}
    eval_ids = [int(round(i)) for i in np.linspace(0, T - 1, args.eval_frames)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mode, t_phase, it_phase, losses = None, t0, 1, []
    for it in range(1, N + 1):
        m = trainer.controller.render_mode
        if m != mode:
            if mode is not None:
                torch.cuda.synchronize()
                log["phases"].append({"mode": mode.name, "iterations": it - it_phase, "ms_per_step": 1e3 * (time.perf_counter() - t_phase) / max(it - it_phase, 1),
                                      "anchors": int(pc._anchor.shape[0]), "loss_tail": float(np.mean(losses[-20:]))})
                print(json.dumps(log["phases"][-1]), flush=True)
            mode, t_phase, it_phase, losses = m, time.perf_counter(), it, []
        a0 = int(pc._anchor.shape[0])
        out = trainer.step(it)
        if int(pc._anchor.shape[0]) != a0:
            log.setdefault("adjust_anchor", []).append([it, a0, int(pc._anchor.shape[0])])
        if it % 10 == 0:
            losses.append(float(out.loss))
    torch.cuda.synchronize()
    log["phases"].append({"mode": mode.name, "iterations": N + 1 - it_phase, "ms_per_step": 1e3 * (time.perf_counter() - t_phase) / max(N + 1 - it_phase, 1),
                          "anchors": int(pc._anchor.shape[0]), "loss_tail": float(np.mean(losses[-20:])) if losses else None})
    print(json.dumps(log["phases"][-1]), flush=True)
    log["fit_seconds"] = time.perf_counter() - t0
    log["repeated_steps"] = int(getattr(trainer, "repeated_steps", 0))
    trainer.sync_replicas()          # z-range ownership (GSVC_DP_ZOWN=1): every replica whole before the model is read as a whole
    log["data_parallel"] = {"ranks": world, "backend": torch.distributed.get_backend() if world > 1 else None,
                            "per_anchor_exchange": "z-range ownership" if trainer._zown is not None else ("rows / dense" if world > 1 else None)}
    trainer.close()
    if world > 1:
        torch.distributed.barrier()
        if rank != 0:
            torch.distributed.destroy_process_group()
            return

    @torch.no_grad()
    def psnr_of(model, mode_):
        vals = []
        for i in eval_ids:
            fr = cube[i]
            img = torch.clamp(render_pair(fr, model, pipe, bg, mode=mode_).rendered_image, 0, 1)
            vals.append(float(psnr_func(img, torch.clamp(fr.image.to(dev), 0, 1).permute(0, 2, 1))))
        return float(np.mean(vals))

    with torch.no_grad():
        log["psnr_full_precision"] = psnr_of(pc, GenerateMode.TRAINING_FULL_PRECISION)
        log["psnr_ste_phase"] = psnr_of(pc, GenerateMode.TRAININ_STE_ENTROPY)
        _, est = pc.estimate_final_bits()
        # (a) the streams under the trained fp32 MLPs: what the two identities are checked on
        if args.dump_coder_inputs:
            from gsvc_amd import stream_codec as SC
            real, dump = SC.encoder_gaussian, {}

            def spy(x, mean, scale, Q, lo, hi, file_name=None):
                i = len(dump) // 3
                if i < 12:
                    dump[f"sym{i}"] = x.detach().reshape(-1).cpu().numpy().astype(np.int32)
                    dump[f"mu{i}"] = (mean / Q).detach().reshape(-1).cpu().numpy()
                    dump[f"sigma{i}"] = (scale / Q).detach().reshape(-1).cpu().numpy()
                return real(x, mean, scale, Q, lo, hi, file_name)
            SC.encoder_gaussian = spy
            try:
                pack = conduct_stream_encoding(pc)
            finally:
                SC.encoder_gaussian = real
            np.savez_compressed(args.dump_coder_inputs, **dump)
        pack = conduct_stream_encoding(pc)
        dec = conduct_stream_decoding(copy.deepcopy(pc), pack)
        log["psnr_decoded"] = psnr_of(dec, GenerateMode.DECODING_AS_IS)
        # Exactly what does the decoder reproduce?  The model whose attributes are replaced by their straight-through values (what the
        # STE phase computes per step: reference guassian.py:197-221) and stored as a decoded model — same anchors, same order, nothing
        # pruned.  The decoded model (pruned by the masks, reordered by the geometry codec and the z-slabs) must render the same
        # picture (0.01 dB; bit for bit in the 1080p runs).  The STE-phase render itself differs from both by its anchor-level visibility test, which reads the
        # UNquantised scalings while training and the decoded ones afterwards (reference ortho_gaussian_renderer/preprocess.py:99-104
        # on pc.get_scaling): a few hundred anchors at the slab's edge are in one set and not the other.
        from gsvc_amd.encodings import STE_multistep
        ec_all = pc.calc_entropy_context(pc.get_anchor)
        qm = copy.deepcopy(pc)
        qm._anchor_feat = torch.nn.Parameter(STE_multistep.apply(pc._anchor_feat, 1 * ec_all.Q_feat_adj, pc._anchor_feat.mean()))
        qm._offset = torch.nn.Parameter(STE_multistep.apply(pc._offset, (0.2 * ec_all.Q_offsets_adj).unsqueeze(1), pc._offset.mean()))
        qm._scaling = torch.nn.Parameter(STE_multistep.apply(pc.get_scaling, 0.001 * ec_all.Q_scaling_adj, pc.get_scaling.mean()))
        qm._anchor, qm._mask = torch.nn.Parameter(pc.get_anchor.clone()), torch.nn.Parameter(pc.get_mask.clone())
        qm.decoded_version = True
        log["psnr_quantised_model"] = psnr_of(qm, GenerateMode.DECODING_AS_IS)
        fr0 = cube[eval_ids[len(eval_ids) // 2]]
        img_q = render_pair(fr0, qm, pipe, bg, mode=GenerateMode.DECODING_AS_IS).rendered_image
        img_d = render_pair(fr0, dec, pipe, bg, mode=GenerateMode.DECODING_AS_IS).rendered_image
        img_s = render_pair(fr0, pc, pipe, bg, mode=GenerateMode.TRAININ_STE_ENTROPY).rendered_image
        log["decoded_vs_quantised_model_max_abs"] = float((img_q - img_d).abs().max())
        d_img = (img_s - img_d).abs()
        from gsvc_amd.ortho_gaussian_renderer import prefilter_voxel
        v_train, v_dec = prefilter_voxel(fr0, pc, pipe, bg), prefilter_voxel(fr0, qm, pipe, bg)
        log["ste_vs_decoded_one_frame"] = {"pixels_over_1e-3": int((d_img.amax(dim=0) > 1e-3).sum()), "max_abs": float(d_img.max()),
                                           "mean_abs": float(d_img.mean()), "pixels": int(d_img[0].numel()),
                                           "visible_anchors_training_test": int(v_train.sum()), "visible_anchors_decoded_test": int(v_dec.sum()),
                                           "anchors_in_one_set_only": int((v_train != v_dec).sum())}
        bits = pack.bits()
        from gsvc_amd.codec import _HEADER, _parse
        framing = 8 * sum(_HEADER.size + 9 * _parse(st)[0][4] for grp in (pack.feat, pack.scaling, pack.offsets) for st in grp if len(st))
        est_attr = est.bit_feat + est.bit_scaling + est.bit_offsets
        got_attr = bits["bit_feat"] + bits["bit_scaling"] + bits["bit_offsets"]
        log["attribute_framing_bits"] = int(framing)
        log["attribute_payload_vs_estimate"] = (got_attr - framing) / max(est_attr, 1.0)
        log["bits_measured"] = {k: int(v) for k, v in bits.items()}
        log["bits_estimated"] = {k: float(v) for k, v in vars(est).items()}
        log["attribute_bytes_vs_estimate"] = got_attr / max(est_attr, 1.0)
        # (b) the shipped form (reference scene/gaussian_model.py:2313-2317): MLPs quantised to 8 bits first, streams coded under them
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            q = copy.deepcopy(pc)
            mlp_file = os.path.join(tmp, "mlp.bin")
            pack_q = conduct_stream_encoding(q, mlp_file=mlp_file)
            pack_q.save(os.path.join(tmp, "streams"))
            total_bytes = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(tmp) for f in fs)
            dec_q = conduct_stream_decoding(copy.deepcopy(q), pack_q, mlp_file=mlp_file)
        lp = None
        if args.lpips_weights:
            from gsvc_amd.lpips import lpips_fn_from
            lp = lpips_fn_from(args.lpips_weights, lin_weights_path=args.lpips_lin_weights, device=dev)
        ev = evaluate(dec_q, cube, pipe, bg, frame_ids=eval_ids, lpips_fn=lp)
        log["decoded_8bit_mlp"] = ev
        log["total_bytes"] = int(total_bytes)
        log["bpp"] = 8.0 * total_bytes / (H * W * T)
        log["anchors_final"], log["anchors_coded"] = int(pc._anchor.shape[0]), int(pack.n)
    d_psnr = abs(log["psnr_decoded"] - log["psnr_quantised_model"])
    log["checks"] = {"decoded_equals_quantised_model_dB": d_psnr, "decoded_vs_quantised_model_max_abs": log["decoded_vs_quantised_model_max_abs"],
                     "ste_phase_minus_decoded_dB": log["psnr_ste_phase"] - log["psnr_decoded"], "attribute_payload_within_2pct": abs(log["attribute_payload_vs_estimate"] - 1.0)}
    print(json.dumps(log))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(log, f, indent=1)
    assert d_psnr <= 0.01, \
        f"decoded PSNR {log['psnr_decoded']:.5f} vs the quantised model's {log['psnr_quantised_model']:.5f}, max pixel difference {log['decoded_vs_quantised_model_max_abs']:.2e}"
    assert abs(log["attribute_payload_vs_estimate"] - 1.0) <= args.payload_tol, (log["attribute_payload_vs_estimate"], log["attribute_bytes_vs_estimate"])
    if world > 1:
        torch.distributed.destroy_process_group()
    print(f"RD point: {log['decoded_8bit_mlp']['psnr']:.2f} dB PSNR, MS-SSIM {log['decoded_8bit_mlp']['msssim']:.4f} at {log['bpp']:.4f} bpp "
          f"({total_bytes / 2 ** 20:.2f} MiB for {T} frames {W}x{H}); fit {log['fit_seconds']:.1f} s")


if __name__ == "__main__":
    main()
