#!/usr/bin/env python3
"""bench.py — headline benchmark of the GSVC hot path on MI355X (contract: see the task statement).

    python bench.py --gpus N --steps K --warmup W [--workload headline|train_step|raster_fwd|raster_fwdbwd]

Workloads (a "step" is one pass of the hot path over synthetic UVG-shaped 1080p input resident in HBM):
  train_step     BASELINE.json configs[2] (the largest single-GPU configuration of the metric): 1080p, 16-frame z-slab,
                 ~50k visible anchors x K=10 = ~500k Gaussians submitted per render, one full fitting step = 4 renders
                 (2 adjacent frames x 2 opposite views) forward + backward, hash grid, entropy loss (lambda 0.004,
                 TRAINING_ENTROPY mode), L1 / SSIM / optical-flow losses, Adam; 200 untimed fitting steps first
  raster_fwd     BASELINE.json configs[1]: 1080p single frame, 200k Gaussians, forward raster only (render fps)
  raster_fwdbwd  same scene, forward + backward of the rasterizer alone
  stream_decode  BASELINE.json configs[4] shape on one GPU: 4K frames, anchors such that ~2 M Gaussians are generated per frame
                 (16-frame slab of a 300-frame cube), attributes drawn from the model's own entropy context; the whole model is
                 entropy-coded, decoded (masks, hash tables, three attribute streams per z-slab) and 24 two-view frames are
                 rendered from the decoded model: stream_decode fps = frames / (decode + render)
  headline       (default) train_step as the top-level value / ms_per_step / roofline / cpu_baseline, plus a short
                 raster_fwd run reported under the side key "raster_fwd" (render fps, single and two-view)

--gpus N > 1: when not already under torch.distributed.run (no WORLD_SIZE in the environment) this process starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD before anything touches the GPU,
relays its JSON line and exits with its return code.  One rank per GPU over RCCL; frames shard across ranks
(rank r samples its pairs from its own block of frames), gradients are all-reduced once per step -> weak scaling.

One JSON line on rank 0.  `value` = Gaussians rasterized per second, whole job (sum over ranks and over the 4
renders of Gaussians with radius > 0, x steps, / max-over-ranks wall time).  `roofline` is for the dominant
rasterizer kernel, timed with HIP events on its launch stream inside this process (a second pass of K steps with
the library's per-kernel event hooks on); `roofline.traffic` comes from rocprofv3 PMC passes of this same command
(they cannot run inside this process): the file and binary version they were taken from are named beside it.
`cpu_baseline` = the CPU oracle (a port, not the product) on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
METRIC = "train-step Gaussians/sec + render fps @1080p"
PAIR_NOTE = ("two-view frames (reference utils/report_utils.py:297-319: view + opposite view + flip + average) come from "
             "gsvc_raster_forward_pair: one binning, the opposite view composited by a second pass over the same sorted list from "
             "its end at that view's own fp32 coordinates with its own alpha / T decisions — within 1e-4 of two separate renders "
             "on every pixel without a borderline decision (tests/test_raster_gpu.py::test_two_view_pair_kernel_matches_two_renders)")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="headline",
                    choices=["headline", "raster_fwd", "raster_fwdbwd", "train_step", "stream_decode"])
    ap.add_argument("--gaussians", type=int, default=200_000, help="raster workloads: Gaussians in the slab")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--frames", type=int, default=600, help="raster workloads: frames of the cube")
    ap.add_argument("--sigma-px", type=float, nargs=2, default=(0.5, 4.0),
                    help="raster workloads: range of the Gaussians' per-axis sigma in pixels (log-uniform); BASELINE.md section 2 "
                         "uses 0.5 .. 4; a fitting render has footprints of ~13 tiles per Gaussian (try 2 .. 12)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="headline: skip the raster_fwd side run")
    ap.add_argument("--anchors", type=int, default=245_000,
                    help="train_step: anchors in the cube (245k in 64 frames x 1.1 bleed -> ~50k in a 16-frame slab)")
    ap.add_argument("--train-frames", type=int, default=64, help="train_step: frames of the synthetic video")
    ap.add_argument("--cfg3", action="store_true",
                    help="train_step: BASELINE.json configs[3]'s per-GPU workload, reference cfgs/cfg_20240919.yaml AS IS: 100 000 "
                         "anchors (init_anchor_num), a 600-frame 1080p video, threshold = .05 (a +-48-frame z-slab, reference "
                         "arguments/__init__.py:54), lambda = .004; overrides --anchors / --train-frames")
    ap.add_argument("--scene-seed", type=int, default=0,
                    help="train_step: seed of the headline scene's fit (torch / numpy / frame draws); recorded in config.scene")
    ap.add_argument("--live-fit", action="store_true",
                    help="train_step: fit the untimed steps in the default (fast, float-atomic) mode — the model the timed steps start "
                         "from then differs run to run (the fit is chaotic: +-10 %% active Gaussians); default: the FROZEN scene, "
                         "fitted under GSVC_DETERMINISTIC=1 (the same bits every run: config.scene.checksum)")
    ap.add_argument("--pretrain", type=int, default=200,
                    help="train_step: untimed fitting steps before the warmup (BASELINE.md section 2: 200, so that opacities "
                         "and scales are no longer at their initial values; they also let the caching allocator see every "
                         "tensor size the varying visible set produces)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def spawn_ranks(args) -> int:
    """--gpus N without WORLD_SIZE: run N ranks of this script under torch.distributed.run as a child process.
    Nothing in this (parent) process has touched the GPU: torch.cuda.device_count() does not initialise it."""
    import socket
    import torch
    share = bool(os.environ.get("GSVC_SHARE_GPU"))
    have = torch.cuda.device_count()
    if not share and have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} requested but this node exposes {have} GPU(s); refusing to run fewer "
                         f"ranks under the same label (GSVC_SHARE_GPU=1 GSVC_DIST_BACKEND=gloo is the single-GPU test knob)\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        else:
            sys.stderr.write(out + "\n")
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited without a JSON line\n")
        rc = 1
    return rc


# ------------------------------------------------------------------------------------------------ helpers
def csrc_fingerprint():
    """sha256 (first 16 hex digits) over the kernel sources the library is built from (tools/pmc_extract.py writes the same
    number next to the counters it extracts)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gsvc_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def reference_step_cpu(which):
    """The REFERENCE's own step body (pipeline/train.py:348-462) timed on PyTorch-CPU with oracle/ in the native slots, from the
    committed record (tests/golden/time_reference_step_cpu.py -> profiles/r05/reference_step_cpu_timing.json): it was measured in
    the build container — /root/reference does not exist on the GPU box — so the host is stated with it.  Same work as the
    headline step minus the optimizer; None when the record is missing."""
    try:
        data = json.load(open(os.path.join(ROOT, "profiles", "r05", "reference_step_cpu_timing.json")))
        case = next(c for c in data["cases"] if c["case"].startswith(which))
        return {"value": case["gaussians_per_s"], "unit": "Gaussians/s", "kind": "reference", "cores": data["host"]["cpus"],
                "seconds_per_step": case["seconds_per_step"]["median"], "active_gaussians_per_step": case["active_gaussians_per_step"],
                "host": data["host"]["where"], "what": data["what"], "case": case["case"], "measured_live": False,
                "file": "profiles/r05/reference_step_cpu_timing.json"}
    except Exception:  # noqa: BLE001
        return None


def pmc_traffic(workload, kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC extract (tools/pmc_extract.py) + where it came from.  The counters
    cannot be collected inside this process (rocprofv3 wraps the command), so the number is only as good as the file: when the
    kernel sources have changed since the counters were taken the traffic is REFUSED (None + the reason), not reported stale."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        data = json.load(open(path))
        val = data.get(workload, {}).get(kernel)
        valu = data.get("valu", {}).get(workload, {}).get(kernel)
        src = {"measured_live": False, "file": "profiles/pmc_latest.json", "valu": valu,
               "binary": data.get("_binary", {}).get(workload, "unknown"),
               "passes": data.get("_source", {}).get(workload, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes")}
        # (the forward kernels' counters of `raster_fwd` are the ones of the raster_fwdbwd pass: same launches, same fingerprint entry)
        sha = data.get("_csrc_sha16", {})
        taken, now = sha.get(workload, sha.get("raster_fwdbwd") if workload == "raster_fwd" else None), csrc_fingerprint()
        if taken != now:
            src["valu"] = None
            src["refused"] = (f"gsvc_amd/csrc has changed since these counters were taken (sources then {taken}, now {now}): "
                              f"re-run tools/profile_round.sh")
            return None, src
        return val, src
    except Exception:  # noqa: BLE001
        return None, {"measured_live": False, "file": None}


# Instruction census and prices of the compositing backward (csrc/raster_bwd.hip), settled in round 5 with measurements whose answer
# is known (tools/micro/valu_issue.hip -> profiles/r05/microbench_valu_issue_allcu.txt, microbench_valu_issue_pmc_SQ.csv):
#   * the polynomial replay of one list entry over one 8x8 quadrant is 33 vector instructions in the built kernel (32.4 on average
#     over the quadrants of an entry): 26 plain all-register fma / fmac / mul / sub ("fast"), 5 that read the constant bus, a literal
#     or VCC (v_min literal, two v_cmp against SGPRs, two v_cndmask: "slow"), 2 transcendental (v_exp_f32, v_rcp_f32: "quarter");
#   * the nine-sum reduction is 74 cross-lane instructions per four entries: 27 permlane swaps ("quarter"), 27 adds ("fast"),
#     20 DPP adds ("slow").
# What one SIMD sustains per wave-instruction at this kernel's 4 waves per SIMD with every CU busy (ns; cycles at the clock the
# chip then holds, 1.95-2.4 GHz): fast 1.33 ns (2.7 cycles; 2.4 at 8 waves: the guide's 2-cycle issue is approached only by
# all-register plain instructions at 8 waves per SIMD), slow 1.90 ns (4.5 cycles at 4 and at 8 waves), quarter 3.54 ns (8.4 cycles).
# GUIDE_NS: MI355X_MICROARCH.md's price, 2 cycles per wave64 instruction at 2.4 GHz (transcendentals 4x), for comparison.
REPLAY_CLASSES = {"fast": 25.5, "slow": 4.9, "quarter": 2.0}                  # per (entry, quadrant) replay: 32.4 instructions
REDUCE_CLASSES = {"fast": 6.75, "slow": 5.0, "quarter": 6.75}                 # per entry: 18.5 instructions
CLASS_NS = {"fast": 1.33, "slow": 1.90, "quarter": 3.54}
GUIDE_NS = {"fast": 2 / 2.4, "slow": 2 / 2.4, "quarter": 8 / 2.4}
REPLAY_INSTS, REDUCE_INSTS_PER_ENTRY, SIMDS = sum(REPLAY_CLASSES.values()), sum(REDUCE_CLASSES.values()), 256 * 4


def roofline_valu(probe, kern_dom, one_stream_us, traffic_src):
    """The roof k_blend_bwd_tile is really under: vector-instruction issue.  ``probe`` = [replays, valid lanes, entries, launches]
    counted by the kernel itself on the timed scene (gsvc_profile_enable bit 1).  The issue time is priced twice: with the
    per-class rates measured on this chip (``issue_model_us``) and with the guide's 2 cycles per instruction at 2.4 GHz
    (``issue_guide_us``: a rate this instruction mix cannot reach — see the census above)."""
    replays, lanes, entries, launches = probe
    if not launches or not replays:
        return None
    rep, ent = replays / launches, entries / launches

    def issue_us(prices):
        per_rep = sum(n * prices[c] for c, n in REPLAY_CLASSES.items())
        per_ent = sum(n * prices[c] for c, n in REDUCE_CLASSES.items())
        return (rep * per_rep + ent * per_ent) / SIMDS * 1e-3
    model_us, guide_us = issue_us(CLASS_NS), issue_us(GUIDE_NS)
    valu = (traffic_src or {}).get("valu") or {}
    out = {"kernel": "k_blend_bwd", "bound": "valu", "measured_live": True,
           "entries_replayed_per_launch": ent, "quadrant_replays_per_launch": rep, "replays_per_entry": rep / max(ent, 1.0),
           "valid_lane_frac": lanes / (64.0 * replays),
           "insts_per_entry_pixel": {"replay": REPLAY_INSTS, "reduction_per_entry": REDUCE_INSTS_PER_ENTRY,
                                     "per_useful_pixel": REPLAY_INSTS / max(lanes / (64.0 * replays), 1e-9) / 64.0},
           "instruction_classes": {"replay": REPLAY_CLASSES, "reduction_per_entry": REDUCE_CLASSES,
                                   "ns_per_wave_instruction_and_simd": CLASS_NS,
                                   "source": "tools/micro/valu_issue.hip at 4 waves per SIMD, every CU busy (profiles/r05/microbench_valu_issue_allcu.txt)"},
           "issue_model_us": model_us,
           "issue_model": "(replays x replay classes + entries x reduction classes) x measured ns per class / 1024 SIMDs; list walking, "
                          "culling tests, the entries' LDS reads and the row stores are not in it",
           "issue_guide_us": guide_us,
           "issue_guide": "the same instruction counts at MI355X_MICROARCH.md's 2 cycles per wave64 instruction and 2.4 GHz (8 cycles for "
                          "transcendentals and permlane swaps)",
           "avg_launch_us_one_stream": one_stream_us, "avg_launch_us": None if kern_dom is None else kern_dom["avg_us"],
           "frac_of_issue_model": None if not one_stream_us else model_us / one_stream_us,
           "frac_of_guide_issue": None if not one_stream_us else guide_us / one_stream_us,
           "valu_issue_share_pmc": valu.get("valu_issue_share_at_2p4GHz"), "valu_insts_per_launch_pmc": valu.get("valu_insts_per_launch"),
           "note": "probe counters come from the kernel's diagnostic instantiation on the timed scene; the *_pmc entries are the SQ "
                   "counters of profiles/pmc_latest.json (null when the kernel sources changed since).  SQ_ACTIVE_INST_VALU is a static "
                   "count (1 per plain, 2 per transcendental instruction: exactly 1.000 / 1.091 per instruction on the microbenchmark "
                   "at every occupancy), not a measure of issue-port occupancy"}
    return out


def _dp_on(world):
    """The data-parallel path: more than one rank, or GSVC_DP_FORCE=1 (test knob: the same collectives on a one-rank RCCL group)."""
    return world > 1 or os.environ.get("GSVC_DP_FORCE") == "1"


def timed(torch, dist, world, fn, steps):
    """Barrier + synchronize on both sides, max over ranks."""
    torch.cuda.synchronize()
    if _dp_on(world):
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if _dp_on(world):
        dist.barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def reduce_sum_max(torch, dist, world, dev, total, elapsed):
    t = torch.tensor([float(total), float(elapsed)], device=dev, dtype=torch.float64)
    if _dp_on(world):
        a, b = t[0:1].clone(), t[1:2].clone()
        dist.all_reduce(a, op=dist.ReduceOp.SUM)
        dist.all_reduce(b, op=dist.ReduceOp.MAX)
        return float(a.item()), float(b.item())
    return float(t[0].item()), float(t[1].item())


# ------------------------------------------------------------------------------------------------ stream_decode (cfg-5 shape)
def run_stream_decode(args, dev, height=2160, width=3840, frames=300, gaussians_per_frame=2_000_000, n_render=24):
    """Entropy decode of a whole coded model + decoder render loop at the BASELINE.json configs[4] shape (reference
    pipeline/stream_decode.py: conduct_stream_decoding, then render_frames)."""
    import copy
    import numpy as np
    import torch
    from gsvc_amd.arguments import cfg_20240919
    from gsvc_amd.frame import SyntheticFrameCube
    from gsvc_amd.model import GaussianModel
    from gsvc_amd.ortho_gaussian_renderer import render_frames
    from gsvc_amd.stream_codec import conduct_stream_decoding, conduct_stream_encoding
    mp_, opt, pipe = cfg_20240919()
    cube = SyntheticFrameCube(height, width, frames, seed=1234, device=dev)
    mp_.threshold = 8.0 / cube.scale
    K = mp_.n_offsets
    anchors = int(gaussians_per_frame / K * frames * 1.1 / 16)        # V = anchors * 16 / (frames * 1.1 bleed)
    torch.manual_seed(0)
    pc = GaussianModel(mp_, mp_.anchor_feature_dim, K, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                       mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                       log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device=dev)
    rng = np.random.default_rng(0)
    lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
    pc.create_from_points(rng.uniform(lim, -lim, (anchors, 3)), 1.0)
    pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
    g = torch.Generator(device=dev).manual_seed(17)
    with torch.no_grad():
        # a calibrated model without fitting it: entropy nets away from their initial values, attributes drawn from the
        # context model itself, 1 in 11 anchors without a live offset, ~15 % of the offsets masked
        for net in (pc.mlp_feature_enet, pc.mlp_scaling_enet, pc.mlp_offset_enet):
            for prm in net.parameters():
                prm.add_(torch.randn(prm.shape, device=dev, generator=g) * 0.02)
        pc._mask.copy_(torch.randn(pc._mask.shape, device=dev, generator=g) * 4.0 + 4.0)
        pc._mask[::11] = -9.0
        for lo in range(0, anchors, 1 << 19):
            sl = slice(lo, lo + (1 << 19))
            ec = pc.calc_entropy_context(pc.get_anchor[sl])
            pc._anchor_feat[sl] = ec.mean_feat + ec.scale_feat * torch.randn(ec.mean_feat.shape, device=dev, generator=g)
            off = ec.mean_offsets + ec.scale_offsets * torch.randn(ec.mean_offsets.shape, device=dev, generator=g)
            pc._offset[sl] = off.view(-1, K, 3)
            del ec, off      # (the scalings stay at their 3-NN initial values: footprints of a few pixels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pack = conduct_stream_encoding(pc)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    target = copy.deepcopy(pc)
    bg = torch.zeros(3)
    mid = frames // 3      # away from z = 0: the decoded model pads its tensors to the encoder's anchor count with rows at the origin
    fr = [cube.get_dummy_frame(i) for i in range(mid - n_render // 2, mid + n_render // 2)]
    warm = conduct_stream_decoding(copy.deepcopy(pc), pack)              # allocator + kernel warm-up, as a player that loops
    for _ in render_frames(fr[:4], warm, pipe, bg):
        pass
    del warm
    from gsvc_amd import anchor_codec
    anchor_codec.decode_anchors_gpu(pack.anchor_stream, dev)     # warm-up (first-use allocations), as for the streams above
    torch.cuda.synchronize()
    ta0 = time.perf_counter()
    geo = anchor_codec.decode_anchors_gpu(pack.anchor_stream, dev)      # the anchor geometry: the reference's tmc3 step
    torch.cuda.synchronize()
    anchor_decode_s = time.perf_counter() - ta0
    assert np.array_equal(geo.cpu().numpy().astype(np.uint16), pack.anchors_q)
    pack.anchors_q_dev = geo            # the attribute decode below starts from the decoded geometry where it is: on the device
    th0 = time.perf_counter()
    anchor_codec.decode_anchors(pack.anchor_stream)
    anchor_host_s = time.perf_counter() - th0
    t2 = time.perf_counter()
    dec = conduct_stream_decoding(target, pack)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    n = 0
    act = 0
    for img in render_frames(fr, dec, pipe, bg):
        n += 1
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    from gsvc_amd.ortho_gaussian_renderer import prefilter_voxel
    vis = int(prefilter_voxel(fr[len(fr) // 2], dec, pipe, bg).sum().item())
    bits = pack.bits()
    return {"workload": f"stream_decode (BASELINE.json configs[4] shape): {height}x{width}, {frames}-frame cube, {anchors} anchors x K={K}, "
                        f"16-frame slab, {n} two-view frames rendered from the decoded model",
            "anchors_coded": pack.n, "gaussians_generated_per_frame": vis * K, "slabs": len(pack.slabs),
            "encode_ms": (t1 - t0) * 1e3, "decode_ms": (t3 - t2) * 1e3, "render_ms_per_frame": (t4 - t3) * 1e3 / max(n, 1),
            "stream_decode_fps": n / (t4 - t2), "render_fps_after_decode": n / (t4 - t3),
            "anchor_geometry_decode_ms": anchor_decode_s * 1e3, "anchor_geometry_decode_host_numpy_ms": anchor_host_s * 1e3,
            "anchor_bits_per_anchor": bits["bit_anchor"] / max(pack.n, 1),
            "stream_decode_fps_incl_anchor_geometry": n / (t4 - t2 + anchor_decode_s),
            "anchor_geometry_note": "occupancy octree + rANS over the voxel lattice (gsvc_amd/anchor_codec.py; the reference runs the "
                                    "external tmc3 executable here), decoded on the GPU (csrc/anchor.hip: one workgroup per level's "
                                    "entropy stream, then the expansion level by level); decode_ms / stream_decode_fps are the attribute, "
                                    "mask and hash-table streams, the _incl_ number adds the geometry decode",
            "megabytes": {k[4:]: round(v / 8 / 2 ** 20, 3) for k, v in bits.items()},
            "render_fps_note": PAIR_NOTE}


_AFFINITY0 = _AFFINITY_GPU = None


def _gpu_cores():
    """Back onto the GPU's NUMA node after a CPU leg."""
    if _AFFINITY_GPU is not None:
        try:
            os.sched_setaffinity(0, _AFFINITY_GPU)
        except OSError:
            pass


def _cgroup_cpus():
    """CPU time this process may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown.  A GPU box of
    this pool hands a one-GPU job 16 cores' worth of a 256-thread host: threads beyond the quota only take turns (measured: the
    oracle's backward 0.44 s on 16 threads, 2.4 s on 256)."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1, int(float(quota) / float(period) + 0.5))
    except Exception:  # noqa: BLE001
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            return max(1, int(q / p + 0.5))
    except Exception:  # noqa: BLE001
        pass
    return None


def _all_cores() -> int:
    """Threads for the CPU baseline: every core the process was started with (the GPU-side NUMA binding is undone for it), capped
    by the cgroup's CPU quota — the number reported as ``cores`` is the number of threads that really run."""
    n = os.cpu_count() or 1
    if _AFFINITY0 is not None:
        try:
            os.sched_setaffinity(0, _AFFINITY0)
            n = len(_AFFINITY0)
        except OSError:
            pass
    q = _cgroup_cpus()
    return max(1, min(n, q)) if q else n


# ------------------------------------------------------------------------------------------------ train_step
def run_train_step(args, rank, world, dev):
    """BASELINE.json configs[2]; frames shard over ranks with one gradient all-reduce per step."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from gsvc_amd import _lib
    from gsvc_amd import dist as gdist
    from gsvc_amd.arguments import cfg_20240919
    from gsvc_amd.frame import SyntheticFrameCube
    from gsvc_amd.model import GaussianModel
    from gsvc_amd.train import Trainer

    if args.cfg3:
        args.anchors, args.train_frames = 100_000, 600
    H, W, T = args.height, args.width, args.train_frames
    mp_, opt, pipe = cfg_20240919()
    # the whole synthetic video resident on the device before the timed region (as the reference holds its video in memory)
    cube = SyntheticFrameCube(H, W, T, seed=1234, device=dev).materialize()
    if not args.cfg3:
        mp_.threshold = 8.0 / cube.scale                  # 16-frame sliding window (configs[2]); --cfg3 keeps the yaml's .05
    slab_frames = 2.0 * mp_.threshold * cube.scale
    # jump straight to the entropy-constrained phase (lambda = 0.004, noise quantisation + sampled rate)
    opt.full_precision_training_total, opt.quantized_training_total = 0, 0
    opt.entropy_constrained_train_total = 10 ** 9
    opt.start_stat, opt.update_until = 0, 10 ** 9         # densification statistics on (adjust_anchor itself runs from
                                                          # iteration 1500 every 100 steps: not reached by this run)
    opt.pause_densification = 0
    torch.manual_seed(args.scene_seed)
    np.random.seed(args.scene_seed)
    pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                       mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                       log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device=dev)
    rng = np.random.default_rng(0)
    lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
    pts = rng.uniform(lim, -lim, (args.anchors, 3))
    pc.create_from_points(pts, spatial_lr_scale=1.0)
    pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
    pc.training_setup(opt)
    if _dp_on(world):
        gdist.broadcast_parameters(pc)       # replicas start identical (they are built from the same seeds anyway)
    trainer = Trainer(pc, cube, opt, pipe, mp_, seed=args.scene_seed)
    it = [0]
    last = [None]

    def step():
        it[0] += 1
        last[0] = trainer.step(it[0])
        return last[0]

    # The headline scene, FROZEN: the untimed fit runs in deterministic mode — every float sum in a fixed
    # order, launch choices by row count only (gsvc_amd.switches GSVC_DETERMINISTIC) — so the model the timed steps start from is
    # the same bits in every run of this command (its checksum goes into the line), and `value` stops moving +-10 % with the
    # run's draw of active Gaussians.  The warm-up and the timed steps then run in the default (fast) mode.
    from gsvc_amd import switches as _sw
    frozen = not args.live_fit and not _dp_on(world)
    if frozen:
        os.environ["GSVC_DETERMINISTIC"] = "1"
        _sw.reload()
    try:
        for _ in range(args.pretrain):
            step()
    finally:
        if frozen:
            os.environ.pop("GSVC_DETERMINISTIC", None)
            _sw.reload()
    scene_sum = 0
    with torch.no_grad():          # exact, order-free: the parameters' bit patterns added up as integers
        for _n, p_ in sorted(pc.named_parameters()):
            if p_.dtype == torch.float32 and p_.numel():
                scene_sum = (scene_sum + int(p_.detach().contiguous().view(torch.int32).to(torch.int64).sum())) % (1 << 61)
    scene = {"frozen": bool(frozen), "fit_steps": args.pretrain, "seeds": {"torch": args.scene_seed, "numpy": args.scene_seed, "anchors": 0, "trainer": args.scene_seed, "video": 1234},
             "checksum": f"{scene_sum:016x}",
             "note": ("the untimed fit ran under GSVC_DETERMINISTIC=1: this checksum (integer sum of every parameter's bits after the fit) is "
                      "the same in every run of the command on this library build" if frozen else
                      "live fit in the default mode: float atomics order the fit's sums differently run by run, the model differs")}
    for _ in range(args.warmup):
        step()
    active = torch.zeros((), device=dev, dtype=torch.float64)
    submitted = [0]

    def counted():
        nonlocal active
        o = step()
        active += o.active_gaussians
        submitted[0] += sum(int(r.radii.numel()) for r in o.renders)      # shapes: host metadata, no device read

    mem0 = torch.cuda.memory_stats(dev)
    rep0 = getattr(trainer, "repeated_steps", 0)
    elapsed = timed(torch, dist, world, counted, args.steps)
    mem1 = torch.cuda.memory_stats(dev)
    timed_region = {"device_mallocs": int(mem1.get("num_device_alloc", 0) - mem0.get("num_device_alloc", 0)),
                    "device_frees": int(mem1.get("num_device_free", 0) - mem0.get("num_device_free", 0)),
                    "alloc_retries": int(mem1.get("num_alloc_retries", 0) - mem0.get("num_alloc_retries", 0)),
                    "repeated_steps": int(getattr(trainer, "repeated_steps", 0) - rep0),
                    "reserved_GiB": round(mem1.get("reserved_bytes.all.current", 0) / 2 ** 30, 2)}
    elapsed_local = elapsed                    # this rank's own clock over the timed steps (the line's time is the MAX over the ranks)
    total_units, elapsed = reduce_sum_max(torch, dist, world, dev, active.item(), elapsed)
    if os.environ.get("GSVC_BENCH_AB"):        # diagnostic: alternate a switch on this trainer, same process (stderr)
        from gsvc_amd import switches
        var = os.environ["GSVC_BENCH_AB"]
        for rep in range(3):
            for val in (None, "1"):
                os.environ.pop(var, None) if val is None else os.environ.__setitem__(var, val)
                switches.reload()
                for _ in range(3):
                    step()
                e = timed(torch, dist, world, step, 30)
                sys.stderr.write(f"AB {var}={val}: {1e3 * e / 30:.3f} ms/step\n")
        os.environ.pop(var, None)
        switches.reload()

    # cold ground truth: the same K steps with the video in pinned HOST memory and the step's two frames + flow uploaded per step
    # (what the reference's own step timer includes: pipeline/train.py:332,407-408,464), one step ahead on a copy stream
    cold = None
    host_cube = None
    pinned_bytes = 4 * T * (3 + 2) * H * W            # pictures + flow fields of the whole video in pinned host memory
    if world == 1 and pinned_bytes > 24 * 2 ** 30:
        cold = {"skipped": f"the video would take {pinned_bytes / 2 ** 30:.1f} GiB of pinned host memory (budget 24 GiB)"}
    elif world == 1:
        try:
            from gsvc_amd.frame import HostResidentCube
            host_cube = HostResidentCube(cube, dev)
            trainer.dataset = host_cube
            for _ in range(3):
                step()
            up0 = host_cube.uploads
            e_cold = timed(torch, dist, world, step, args.steps)
            # the link by itself: one pair's upload (pinned -> device, nothing else queued), so that a slow box reads as such
            torch.cuda.synchronize()
            th = time.perf_counter()
            for _ in range(4):
                host_cube._upload(host_cube._new_slot(), 0)
            host_cube._copy.synchronize()
            h2d_s = (time.perf_counter() - th) / 4
            nbytes = int(2 * host_cube._images[0].numel() * 4 + host_cube._flows[0].numel() * 4)
            cold = {"ms_per_step_cold": 1e3 * e_cold / args.steps, "uploads_per_step": (host_cube.uploads - 4 - up0) / args.steps,
                    "bytes_per_upload": nbytes, "upload_alone_ms": 1e3 * h2d_s, "h2d_GBps": nbytes / h2d_s / 1e9,
                    "note": "ground-truth frames + flow in pinned host memory, uploaded per step on a copy stream one step ahead "
                            "(gsvc_amd.frame.HostResidentCube); `value` / ms_per_step above keep the video resident in HBM"}
        except Exception as e:  # noqa: BLE001
            cold = {"error": f"{type(e).__name__}: {e}"}
        finally:
            trainer.dataset = cube
            host_cube = None

    # exposed communication: the same K steps with the gradient exchange switched off (replicas diverge: last thing
    # measured on the model's gradients; parameters are re-broadcast afterwards)
    comm = None
    if _dp_on(world):
        sent_bytes, sparse_used = trainer.reducer.bytes_sent, trainer.reducer._sparse is not None
        zown = getattr(trainer, "_zown", None)
        if zown is not None:      # GSVC_DP_ZOWN=1: halo rows of gradients to their owners + the owners' updated rows back
            sent_bytes += zown.bytes_sent
        early_steps = getattr(trainer, "early_steps", 0)
        trainer.reducer.enabled = False
        for _ in range(3):
            step()
        e2 = timed(torch, dist, world, step, args.steps)
        trainer.reducer.enabled = True
        e2_local = e2
        _, e2 = reduce_sum_max(torch, dist, world, dev, 0.0, e2)
        gdist.broadcast_parameters(pc)
        # every rank's own numbers (the first multi-GPU run should say per rank what the exchange cost and moved)
        mine = {"rank": rank, "ms_per_step": 1e3 * elapsed_local / args.steps, "exposed_ms_per_step": 1e3 * (elapsed_local - e2_local) / args.steps,
                "gradient_bytes_per_step": int(sent_bytes)}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        # ... and the OTHER per-anchor exchange in the same invocation: z-range ownership (halo rows to their owners, updated rows back;
        # gsvc_amd.dist.ZRangeOwnership) when the run used the replicated exchange, and the other way round — same model, same frames
        other = None
        try:
            if os.environ.get("GSVC_BENCH_ONE_EXCHANGE"):      # opt-out: only the run's own exchange
                raise RuntimeError("skipped: GSVC_BENCH_ONE_EXCHANGE is set")
            if zown is None:
                trainer._zown = gdist.ZRangeOwnership(cube.len_z_frames, cube.scale, mp_.threshold)
            else:
                trainer.sync_replicas()
                trainer._zown = None
            for _ in range(3):
                step()
            e3_local = timed(torch, dist, world, step, args.steps)
            _, e3 = reduce_sum_max(torch, dist, world, dev, 0.0, e3_local)
            zo = trainer._zown
            b3 = int(trainer.reducer.bytes_sent + (zo.bytes_sent if zo is not None else 0))
            mine3 = {"rank": rank, "ms_per_step": 1e3 * e3_local / args.steps, "exposed_ms_per_step": 1e3 * (e3_local - e2_local) / args.steps,
                     "gradient_bytes_per_step": b3}
            per_rank3 = [None] * world
            dist.all_gather_object(per_rank3, mine3)
            other = {"per_anchor_exchange": "z-range ownership (halo exchange)" if zo is not None else "replicated (rows of the distinct visible anchors / dense all-reduce)",
                     "ms_per_step": 1e3 * e3 / args.steps, "exposed_ms_per_step": 1e3 * (e3 - e2) / args.steps, "per_rank": per_rank3}
        except Exception as e:  # noqa: BLE001          (the line above must not depend on the second measurement)
            other = {"error": f"{type(e).__name__}: {e}"}
        finally:
            if trainer._zown is not None and zown is None:
                trainer.sync_replicas()            # every replica whole again before the model is read as a whole below
            trainer._zown = zown
            pc._zown = None
            gdist.broadcast_parameters(pc)
        comm = {"ms_per_step_without_exchange": 1e3 * e2 / args.steps,
                "exposed_ms_per_step": 1e3 * (elapsed - e2) / args.steps,
                # what this rank handed to the collectives in the last exchanged step: per-anchor gradients as rows of its distinct
                # visible anchors (index + row lists, padded to the largest count over the ranks), hash tables and MLPs dense
                "gradient_bytes_per_step": int(sent_bytes),
                "dense_gradient_bytes": 4 * sum(p.numel() for g in pc.optimizer.param_groups for p in g["params"] if p.requires_grad),
                "per_anchor_exchange": ("z-range ownership: halo rows of gradients to their owners, updated rows back (all_to_all between neighbours)"
                                        if zown is not None else
                                        "rows of the distinct visible anchors (all-gather + scatter-add)" if sparse_used else "dense all-reduce"),
                "early_plan_steps": early_steps, "per_rank": per_rank, "other_exchange_same_run": other}

    # per-kernel pass: same K steps with HIP events around every launch on the launch stream
    _lib.profile_enable(True)
    inst = inst_api = 0
    for _ in range(args.steps):
        out = step()
        # the instances the kernels WORK on (the tile lists' length): the renderer lists a Gaussian only where its alpha box reaches
        # (GSVC_RASTER_TIGHT_BINNING), num_rendered stays the 3-sigma count of the API
        inst += sum(r.raster_state.listed_instances() for r in out.renders)
        inst_api += sum(r.num_rendered for r in out.renders)
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    _lib.profile_enable(False)
    # the step's four renders run on two streams: a rasterizer kernel's launch duration above includes the time it shares the chip
    # with the other stream's kernels.  The same kernels with the renders on ONE stream (their exclusive durations), beside
    one_stream = {}
    old_streams = os.environ.get("GSVC_RASTER_STREAMS")
    os.environ["GSVC_RASTER_STREAMS"] = "1"
    from gsvc_amd import switches
    switches.reload()
    try:
        _lib.profile_enable(True)
        for _ in range(max(args.steps // 4, 5)):
            step()
        torch.cuda.synchronize()
        one_stream = {k: 1e3 * ms / max(n, 1) for k, (n, ms) in _lib.profile_collect().items()}
    finally:
        _lib.profile_enable(False)
        if old_streams is None:
            os.environ.pop("GSVC_RASTER_STREAMS", None)
        else:
            os.environ["GSVC_RASTER_STREAMS"] = old_streams
        switches.reload()
    # the compositing backward's replay counters, live on the timed scene (gsvc_profile_enable bit 1: its diagnostic instantiation
    # adds, per launch, the (entry, quadrant) replays, the lanes of those replays that held a contributing pixel and the entries
    # replayed to spare words of the render's counters block, which the forward zeroes)
    probe = [0, 0, 0, 0]
    _lib.profile_enable(2)
    try:
        for _ in range(2):
            o = step()
            torch.cuda.synchronize()
            for r in o.renders:
                c = r.raster_state.binning[64:88].view(torch.int64).tolist()
                for i in range(3):
                    probe[i] += c[i]
                probe[3] += 1
    finally:
        _lib.profile_enable(False)
    out = last[0]

    # decoder loop (reference utils/report_utils.py:297-319: per frame the visibility test, the anchor -> Gaussian
    # generation with the MLPs, and the two-view frame) on every rank's own frames
    from gsvc_amd.generate import GenerateMode
    from gsvc_amd.ortho_gaussian_renderer import render_frames, render_pair
    frames_e2e = [cube.get_dummy_frame(i) for i in range(trainer.lo, max(trainer.lo + 1, min(trainer.hi, trainer.lo + 48)))]
    n_fr = len(frames_e2e)
    for fr in frames_e2e[:4]:
        render_pair(fr, pc, pipe, trainer.background, mode=GenerateMode.DECODING_AS_IS)

    def pair_loop():
        for fr in frames_e2e:
            render_pair(fr, pc, pipe, trainer.background, mode=GenerateMode.DECODING_AS_IS)

    def frames_loop():
        for _ in render_frames(frames_e2e, pc, pipe, trainer.background):
            pass

    tp = timed(torch, dist, world, pair_loop, 1)
    n_all, tp = reduce_sum_max(torch, dist, world, dev, n_fr, tp)
    pair_fps = n_all / tp
    for _ in render_frames(frames_e2e[:8], pc, pipe, trainer.background):
        pass
    tf = timed(torch, dist, world, frames_loop, 1)
    n_all, tf = reduce_sum_max(torch, dist, world, dev, n_fr, tf)
    frames_fps = n_all / tf
    if rank != 0:
        return None
    # the decoder loop's kernels: per frame from the library's launch events over one more pass of the loop
    # (co-running on the loop's streams), and its dominant kernel — the two-view compositing pass — ALONE on the chip on one frame's
    # Gaussians, against the HBM roof with the bytes of BOTH walks over the sorted list
    decoder_loop = None
    try:
        from gsvc_amd.generate import generate_neural_gaussians_many
        from gsvc_amd.ortho_gaussian_renderer.preprocess import prefilter_geometry, prefilter_voxels_many, raster_settings_for
        from gsvc_amd.rasterizer import raster_forward, settings_to_c
        _lib.profile_enable(True)
        frames_loop()
        torch.cuda.synchronize()
        prof_d = _lib.profile_collect()
        _lib.profile_enable(False)
        with torch.no_grad():
            geometry = prefilter_geometry(pc)
            fr = frames_e2e[n_fr // 2]
            vis = prefilter_voxels_many([fr], pc, pipe, trainer.background, geometry=geometry)
            gss = generate_neural_gaussians_many([fr], pc, vis, GenerateMode.DECODING_AS_IS, dense=True, anchors=geometry[0])[0]
            a_ = tuple(t.contiguous() for t in (gss.xyz, gss.color, gss.opacity, gss.scaling, gss.rot))
            cs_ = settings_to_c(raster_settings_for(fr, pc, pipe, trainer.background, 1.0))
            for _ in range(3):
                _, _, st_ = raster_forward(cs_, *a_, pair=True)
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            for _ in range(10):
                raster_forward(cs_, *a_, pair=True, sync=False)
            torch.cuda.synchronize()
            alone = {k: 1e3 * ms / max(c, 1) for k, (c, ms) in _lib.profile_collect().items()}
            _lib.profile_enable(False)
        I_pair = st_.listed_instances()
        pair_bytes = 2 * 40 * I_pair + 20 * H * W
        dom_us = alone.get("k_blend_pair", 0.0)
        decoder_loop = {
            "frames": n_fr, "us_per_frame_wall": 1e6 / frames_fps * world,
            "kernels_us_per_frame_co_running": {k: round(1e3 * ms / n_fr, 1) for k, (c, ms) in sorted(prof_d.items(), key=lambda kv: -kv[1][1])},
            "rasterizer_alone_us": {k: round(v, 1) for k, v in sorted(alone.items())},
            "roofline": {"bound": "hbm", "kernel": "k_blend_pair", "algorithmic_bytes_per_launch": pair_bytes, "avg_launch_us": dom_us,
                         "achieved": pair_bytes / max(dom_us, 1e-9) / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": pair_bytes / max(dom_us, 1e-9) / 1e3 / HBM_PEAK_GBS, "instances": I_pair,
                         "note": "2 x 40 I (the sorted list is walked once per view) + 20 HW; like k_blend the kernel is vector-issue bound, "
                                 "not HBM bound; exclusive duration (alone on the chip), the loop runs it beside the next batch's generation"},
            "note": "render_frames: batches of 8 frames, generation on the current stream, two-view passes on two side streams, batches "
                    "software-pipelined; GPU-bound (profiles/r06/decoder_loop_profile.txt)"}
    except Exception as e:  # noqa: BLE001
        decoder_loop = {"error": f"{type(e).__name__}: {e}"}

    HW = H * W
    n_inst = inst / (4 * args.steps)                      # instances per render (listed: what the roofline's bytes are counted on)
    n_inst_api = inst_api / (4 * args.steps)              # num_rendered per render (tiles of the 3-sigma rectangles)
    P = submitted[0] / (4.0 * args.steps)                 # Gaussians submitted per render, mean over the timed steps
    n_vis = total_units / (4 * args.steps * world)
    kern = {k: {"launches": n, "avg_us": 1e3 * ms / max(n, 1)} for k, (n, ms) in prof.items()}
    alg = {"k_blend": 40 * n_inst + 20 * HW, "k_blend_bwd": 40 * n_inst + 20 * HW, "k_preprocess": 60 * P + 44 * n_vis,
           "k_gaussian_bwd": 88 * n_vis + 124 * P}
    # roofline: the dominant RASTERIZER kernel (north_star: "HBM GB/s on the rasterizer vs the chip's peak"); the MLP /
    # grid kernels of the step are listed under "kernels"
    dom = max((k for k in kern if k in alg), key=lambda k: kern[k]["avg_us"] * kern[k]["launches"])
    dom_bytes = alg[dom]
    achieved = dom_bytes / (kern[dom]["avg_us"] * 1e-6) / 1e9
    kernel_us = sum(v["avg_us"] * v["launches"] for v in kern.values()) / args.steps
    default_shape = (not args.cfg3 and args.anchors == 245_000 and args.train_frames == 64 and (H, W) == (1080, 1920) and world == 1)
    if default_shape:
        traffic, traffic_src = pmc_traffic("train_step", dom)
    else:
        # the PMC passes were taken on the default headline shape (245 k anchors, 64 frames, one GPU): no counter traffic is quoted
        # for any other workload until it has its own pass
        traffic, traffic_src = None, {"measured_live": False, "file": None,
                                      "refused": "profiles/pmc_latest.json holds counters of the default train_step shape only"}
    res = {
        "metric": METRIC, "value": total_units / elapsed, "unit": "Gaussians/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"train_step ({'BASELINE.json configs[3] per-GPU workload, cfg_20240919.yaml as is' if args.cfg3 else 'BASELINE.json configs[2]'}): "
                               f"{H}x{W}, {T}-frame synthetic video, {pc._anchor.shape[0]} "
                               f"anchors x K=10, {slab_frames:.0f}-frame z-slab (threshold {mp_.threshold:.5f}); 4 renders/step fwd+bwd + hash grid + entropy loss "
                               f"(lambda={opt.lmbda}, TRAINING_ENTROPY) + L1/SSIM/optical + Adam; one frame pair per rank; "
                               f"{args.pretrain} untimed fitting steps before the warmup ({'frozen scene: deterministic fit, seed ' + str(args.scene_seed) + ' / video 1234, checksum ' + scene['checksum'] if scene['frozen'] else 'live fit'})",
                   "scene": scene,
                   "gaussians_per_render": P, "active_per_render": n_vis, "active_fraction": n_vis / max(P, 1.0),
                   "instances_per_render": n_inst, "num_rendered_per_render": n_inst_api, "tiles_per_active_gaussian": n_inst / max(n_vis, 1.0),
                   "instances_note": "instances_per_render = entries of the tile lists (a Gaussian is listed in the tiles its alpha >= 1/255 box "
                                     "touches: GSVC_RASTER_TIGHT_BINNING); num_rendered_per_render = the API's count over the 3-sigma rectangles",
                   "visible_anchors_per_render": P / pc.n_offsets,
                   "value_counts": "Gaussians with radius > 0 (active_per_render x 4 renders x steps / s); gaussians_per_render are "
                                   "submitted un-compacted (K per visible anchor), both means over the timed steps",
                   "parallelism": f"frame-shard x{world} + gradient all-reduce" if world > 1 else "single GPU"},
        "effective_batch": world,      # frame pairs per optimizer step: loss = mean over ranks (the reference steps on one pair)
        "dp_anchor_optimizer": ("row-sparse / all-reduce exchange + replicated Adam" if world > 1 else None),
        "rccl_ranks": (dist.get_world_size() if _dp_on(world) else 1),
        "dist_backend": (dist.get_backend() if _dp_on(world) else None),
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "note": "this kernel's binding unit is the vector ALU, not HBM: traffic_source.valu holds the SQ counters of the "
                             "same command (share of the kernel's cycles its SIMDs spent executing vector instructions)",
                     "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_us": kern[dom]["avg_us"],
                     "one_stream": (None if dom not in one_stream else
                                    {"avg_launch_us": one_stream[dom], "achieved": dom_bytes / (one_stream[dom] * 1e-6) / 1e9,
                                     "frac": dom_bytes / (one_stream[dom] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                     "note": "the step's four renders run on two HIP streams, so avg_launch_us above is the launch's "
                                             "duration while it shares the chip with the other stream's kernels (what rocprofv3 reports for "
                                             "the same command); this is the same kernel with the renders on one stream "
                                             "(GSVC_RASTER_STREAMS=1): its exclusive duration"})},
        "roofline_valu": roofline_valu(probe, kern.get("k_blend_bwd"), one_stream.get("k_blend_bwd"), traffic_src),
        "gsvc_kernel_us_per_step": kernel_us,
        "timed_region": timed_region,
        "kernels": {k: {"avg_us": round(v["avg_us"], 2), "launches_per_step": v["launches"] / args.steps} for k, v in kern.items()},
        "render_pair_fps_end_to_end": pair_fps,
        "render_frames_fps_end_to_end": frames_fps,
        "decoder_loop": decoder_loop,
        "render_fps_note": PAIR_NOTE,
    }
    if cold is not None:
        res["cold_ground_truth"] = cold
        if "ms_per_step_cold" in cold:
            res["ms_per_step_cold"] = cold["ms_per_step_cold"]
    # which side bounds the step, measured: what the host blocks per step at the step's two waits (the plan's counts, the renders'
    # counters) is the GPU's lead over it (gsvc_amd.train.Trainer._update_bound); ~0 = the host is the bound
    if getattr(trainer, "_blocked_ema", None) is not None:
        from gsvc_amd import generate as _gen
        res["host_blocked_ms_per_step"] = 1e3 * trainer._blocked_ema
        hint = trainer._ctx.gpu_bound_hint
        res["gpu_bound_measured"] = bool(hint) if hint is not None else None
        res["gpu_bound_flips"] = [{"step": s_, "gpu_bound": g_, "blocked_ms_ema": round(m_, 3)} for s_, g_, m_ in getattr(trainer, "bound_log", [])]
    if comm is not None:
        res["gradient_exchange"] = comm
    if world == 1:
        # stream codec round trip of the fitted model and the decoder's frame rate INCLUDING the entropy decode
        import copy
        from gsvc_amd.stream_codec import conduct_stream_decoding, conduct_stream_encoding
        torch.cuda.synchronize()
        tc0 = time.perf_counter()
        pack = conduct_stream_encoding(pc)
        torch.cuda.synchronize()
        target = copy.deepcopy(pc)
        _lib.profile_enable(True)
        tc1 = time.perf_counter()
        dec = conduct_stream_decoding(target, pack)
        torch.cuda.synchronize()
        tc2 = time.perf_counter()
        kprof = _lib.profile_collect()
        _lib.profile_enable(False)
        n_dec = sum(1 for _ in render_frames(frames_e2e, dec, pipe, trainer.background))
        torch.cuda.synchronize()
        tc3 = time.perf_counter()
        bits = pack.bits()
        from gsvc_amd import anchor_codec
        anchor_codec.decode_anchors_gpu(pack.anchor_stream, dev)
        torch.cuda.synchronize()
        ta0 = time.perf_counter()
        anchor_codec.decode_anchors_gpu(pack.anchor_stream, dev)          # on the GPU (csrc/anchor.hip)
        torch.cuda.synchronize()
        anchor_decode_s = time.perf_counter() - ta0
        res["stream_codec"] = {"encode_ms": (tc1 - tc0) * 1e3, "decode_ms": (tc2 - tc1) * 1e3, "slabs": len(pack.slabs),
                               "anchor_geometry_decode_ms": anchor_decode_s * 1e3, "anchor_bits_per_anchor": bits["bit_anchor"] / max(pack.n, 1),
                               "stream_decode_fps_incl_anchor_geometry": n_dec / (tc3 - tc1 + anchor_decode_s),
                               "anchors_coded": pack.n, "megabytes": {k[4:]: round(v / 8 / 2 ** 20, 4) for k, v in bits.items()},
                               "ans_decode_kernel": {"launches": kprof.get("k_ans_decode", (0, 0.0))[0],
                                                     "sum_ms": kprof.get("k_ans_decode", (0, 0.0))[1],
                                                     "note": "two launches: masks + hash tables, then every attribute stream of every slab (gsvc_ans_decode_many)"},
                               "stream_decode_fps": n_dec / (tc3 - tc1),
                               "note": f"entropy decode of the whole model + {n_dec} two-view frames rendered from it"}
        del dec, pack
    if world == 1:
        # adjust_anchor (anchor growing / pruning + optimizer-state surgery, every update_interval = 100 iterations
        # from iteration 1500 in the reference's schedule: outside the timed steps above) on the statistics those steps gathered,
        # with the reference's thresholds scaled to this short run (check_interval = the steps taken so far / 4 observations each)
        try:
            def timed_adjust(grad_threshold):
                for _ in range(12):                       # new observations for the statistics the last call consumed
                    step()
                torch.cuda.synchronize()
                a0 = int(pc._anchor.shape[0])
                ta = time.perf_counter()
                pc.adjust_anchor(check_interval=3, success_threshold=opt.success_threshold, grad_threshold=grad_threshold,
                                 min_opacity=opt.min_opacity)
                torch.cuda.synchronize()
                return 1e3 * (time.perf_counter() - ta), a0, int(pc._anchor.shape[0])
            timed_adjust(opt.densify_grad_threshold)      # first call: torch loads its sort / unique / isin kernels (~0.1-0.3 s, once per process)
            ms, a0, a1 = timed_adjust(opt.densify_grad_threshold)
            ms_g, g0, g1 = timed_adjust(1e-7)             # every offset slot with enough observations is a candidate: a growing call
            res["adjust_anchor"] = {"ms_per_call": ms, "anchors_before": a0, "anchors_after": a1,
                                    "growing_call": {"ms_per_call": ms_g, "anchors_before": g0, "anchors_after": g1,
                                                     "grad_threshold": 1e-7},
                                    "amortised_ms_per_step": ms / float(opt.update_interval),
                                    "amortised_ms_per_step_growing": ms_g / float(opt.update_interval),
                                    "share_of_a_step": ms / float(opt.update_interval) / res["ms_per_step"],
                                    "note": f"the reference calls it every update_interval = {opt.update_interval} iterations "
                                            f"(pipeline/train.py:567-569): a step carries ms_per_call / {opt.update_interval}; timed after "
                                            "an untimed first call, on 12 steps' statistics (check_interval 3); the growing call lowers "
                                            "the gradient threshold so that anchors are added"}
        except Exception as e:  # noqa: BLE001
            res["adjust_anchor"] = {"error": f"{type(e).__name__}: {e}"}
    ref_cpu = reference_step_cpu("configs[3]" if args.cfg3 else "configs[2]")
    if ref_cpu is not None:
        res["cpu_baseline_reference_python"] = ref_cpu
    if world == 1 and not args.no_cpu_baseline:
        import oracle
        oracle.build()
        r = out.renders[0]
        gs = r.generated_gaussians
        fr = cube.get_dummy_frame(out.frame_idx)
        st = oracle.make_settings(H, W, fr.x_min, fr.y_min, fr.scale, mp_.threshold, fr.view_matrix.permute(1, 0).contiguous().numpy())
        arrs = [t.detach().cpu().numpy() for t in (gs.xyz, gs.color, gs.opacity, gs.scaling, gs.rot)]
        cores = _all_cores()
        t0 = time.perf_counter()
        oracle.raster_forward(st, *arrs, num_threads=cores)                       # (thread start-up, page faults of the scratch)
        reps = 0
        t0 = time.perf_counter()
        while True:                 # bounded sample: whole passes until ~10 s of wall time on all cores
            fwd = oracle.raster_forward(st, *arrs, num_threads=cores)
            oracle.raster_backward(st, *arrs, fwd, np.ones((3, H, W), np.float32), num_threads=cores)
            reps += 1
            tc = time.perf_counter() - t0
            if tc > 10.0 or reps >= 200:
                break
        res["cpu_baseline"] = {"value": reps * float((fwd.radii > 0).sum()) / tc, "unit": "Gaussians/s", "cores": cores, "kind": "port",
                               "sample": f"{reps} passes of the rasterizer forward + backward of ONE of the timed step's 4 renders, same "
                                         f"Gaussians, every stage on {cores} OpenMP threads (preprocess per Gaussian, per-tile sort, blend "
                                         "and backward per tile; the instance emission and two prefix sums are serial); the MLPs, hash "
                                         "grid, rate and image losses of the step are NOT in the CPU sample (it does less work per Gaussian)",
                               "seconds": round(tc, 3)}
        _gpu_cores()
    return res


def run_train_step_late(args, dev, total=3000, stop_frac=0.6, anchors=100_000, steps=40):
    """What a step costs LATE in a fit (the headline model is 200 steps old): a 100 k-anchor model (the reference's anchor count) fitted
    through a schedule scaled to `total` steps up to the middle of its entropy-constrained phase, then `steps` timed steps.  By then the
    Gaussians have grown (tens of tiles each, faint): the rasterizer is most of the step, and what the round's last changes are for."""
    import numpy as np
    import torch
    from gsvc_amd.arguments import cfg_20240919
    from gsvc_amd.frame import SyntheticFrameCube
    from gsvc_amd.model import GaussianModel
    from gsvc_amd.train import Trainer
    from gsvc_amd import rasterizer as _rz
    _rz._capacity_hint.clear()       # (another section's instance capacity is not this model's: buffers several times the need)
    H, W, T = args.height, args.width, args.train_frames
    mp_, opt, pipe = cfg_20240919()
    cube = SyntheticFrameCube(H, W, T, seed=1234, device=dev).materialize()
    mp_.threshold = 8.0 / cube.scale
    s = total / 40_000.0
    opt.iterations = total
    opt.full_precision_training_total, opt.quantized_training_total = int(10_000 * s), int(5_000 * s)
    opt.entropy_constrained_train_total = int(20_000 * s)
    opt.ste_entropy_constrained_train_total = total - int(35_000 * s)
    opt.start_stat, opt.update_from, opt.update_until = int(500 * s), int(1_500 * s), int(25_000 * s)
    opt.update_interval, opt.pause_densification = max(20, int(100 * s)), int(1_000 * s)
    for name in dir(opt):
        if name.endswith("_max_steps"):
            setattr(opt, name, total)
    torch.manual_seed(0)
    np.random.seed(0)
    pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                       mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                       log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device=dev)
    lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
    pc.create_from_points(np.random.default_rng(0).uniform(lim, -lim, (anchors, 3)), spatial_lr_scale=1.0)
    pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
    pc.training_setup(opt)
    stop = int(stop_frac * total)
    with Trainer(pc, cube, opt, pipe, mp_, seed=0) as trainer:
        it = 0
        while it < stop:
            it += 1
            trainer.step(it)
        active = torch.zeros((), device=dev, dtype=torch.float64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            it += 1
            out = trainer.step(it)
            active += out.active_gaussians
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        listed = sum(r.raster_state.listed_instances() for r in out.renders) / 4.0
        api = sum(r.num_rendered for r in out.renders) / 4.0
        mode = trainer.controller.render_mode.name
    n_act = float(active.item())
    return {"workload": f"a fitting step {stop} iterations into a {total}-step fit of the {T}-frame {H}x{W} synthetic video (schedule of the reference scaled, "
                        f"{int(pc._anchor.shape[0])} anchors x K=10, 16-frame slab), phase {mode}; {steps} timed steps",
            "ms_per_step": 1e3 * el / steps, "value": n_act / el, "unit": "Gaussians/s (active: radius > 0)",
            "active_per_render": n_act / (4.0 * steps), "instances_per_render": listed, "num_rendered_per_render": api,
            "tiles_per_active_gaussian": api / max(n_act / (4.0 * steps), 1.0),
            "note": "instances_per_render = entries of the tile lists (tight binning), num_rendered_per_render = the 3-sigma rectangles' tiles (last timed step)"}


def run_train_step_light(args, dev, anchors, steps, pretrain, warmup=5):
    """The configs[2] fitting step at another model size: step time and the per-render counts only (no side measurements).
    Used for the second operating point of the headline line: ~500 k ACTIVE Gaussians per render (radius > 0), where the
    headline workload has ~500 k submitted and ~140 k active."""
    import numpy as np
    import torch
    from gsvc_amd.arguments import cfg_20240919
    from gsvc_amd.frame import SyntheticFrameCube
    from gsvc_amd.model import GaussianModel
    from gsvc_amd.train import Trainer
    H, W, T = args.height, args.width, args.train_frames
    from gsvc_amd import rasterizer as _rz
    _rz._capacity_hint.clear()       # the instance capacity another section of the bench grew to is not this model's (buffers 4x the need)
    mp_, opt, pipe = cfg_20240919()
    cube = SyntheticFrameCube(H, W, T, seed=1234, device=dev).materialize()
    mp_.threshold = 8.0 / cube.scale
    opt.full_precision_training_total, opt.quantized_training_total = 0, 0
    opt.entropy_constrained_train_total = 10 ** 9
    opt.start_stat, opt.update_until, opt.pause_densification = 0, 10 ** 9, 0
    torch.manual_seed(0)
    np.random.seed(0)
    pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                       mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                       log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device=dev)
    rng = np.random.default_rng(0)
    lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
    pc.create_from_points(rng.uniform(lim, -lim, (anchors, 3)), spatial_lr_scale=1.0)
    pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
    pc.training_setup(opt)
    trainer = Trainer(pc, cube, opt, pipe, mp_, seed=0)
    it = 0
    for _ in range(pretrain + warmup):
        it += 1
        trainer.step(it)
    active = torch.zeros((), device=dev, dtype=torch.float64)
    submitted = inst = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        it += 1
        out = trainer.step(it)
        active += out.active_gaussians
        submitted += sum(int(r.radii.shape[0]) for r in out.renders)
        inst += sum(r.num_rendered for r in out.renders)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    n_act = float(active.item())
    return {"workload": f"train_step at {int(pc._anchor.shape[0])} anchors x K=10 (otherwise the headline's: {H}x{W}, {T} frames, 16-frame slab, "
                        f"TRAINING_ENTROPY), {pretrain} untimed fitting steps + {warmup} warm-up, {steps} timed steps",
            "ms_per_step": 1e3 * el / steps, "value": n_act / el, "unit": "Gaussians/s (active: radius > 0)",
            "active_per_render": n_act / (4 * steps), "gaussians_per_render": submitted / (4.0 * steps),
            "active_fraction": n_act / max(submitted, 1), "instances_per_render": inst / (4.0 * steps),
            "repeated_steps": int(getattr(trainer, "repeated_steps", 0))}


def run_train_step_phases(args, dev, anchors=245_000, steps=30, pretrain=100, warmup=6):
    """The fitting step in each phase of the reference's schedule (utils/train_util.py:8-92: FULL_PRECISION 10 k iterations,
    QUANTIZED 5 k, TRAINING_ENTROPY 20 k, STE_ENTROPY 5 k) on one fresh model of the headline's shape, in schedule order (the
    controller's entropy flag is sticky, as the reference's).  The headline `value` is the TRAINING_ENTROPY phase."""
    import numpy as np
    import torch
    from gsvc_amd.arguments import cfg_20240919
    from gsvc_amd.frame import SyntheticFrameCube
    from gsvc_amd.model import GaussianModel
    from gsvc_amd.train import Trainer
    from gsvc_amd import rasterizer as _rz
    _rz._capacity_hint.clear()       # (another section's instance capacity is not this model's: buffers several times the need)
    H, W, T = args.height, args.width, args.train_frames
    mp_, opt, pipe = cfg_20240919()
    share = {"TRAINING_FULL_PRECISION": opt.full_precision_training_total, "TRAINING_QUANTIZED": opt.quantized_training_total,
             "TRAINING_ENTROPY": opt.entropy_constrained_train_total, "STE_ENTROPY": opt.ste_entropy_constrained_train_total}
    cube = SyntheticFrameCube(H, W, T, seed=1234, device=dev).materialize()
    mp_.threshold = 8.0 / cube.scale
    opt.start_stat, opt.update_until, opt.pause_densification, opt.update_from = 0, 10 ** 9, 0, 10 ** 9
    B = 10 ** 9
    (opt.full_precision_training_total, opt.quantized_training_total, opt.entropy_constrained_train_total,
     opt.ste_entropy_constrained_train_total) = B, 0, 0, 0
    torch.manual_seed(0)
    np.random.seed(0)
    pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                       mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                       log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device=dev)
    rng = np.random.default_rng(0)
    lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
    pc.create_from_points(rng.uniform(lim, -lim, (anchors, 3)), spatial_lr_scale=1.0)
    pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
    pc.training_setup(opt)
    trainer = Trainer(pc, cube, opt, pipe, mp_, seed=0)
    it = 0
    for _ in range(pretrain):
        it += 1
        trainer.step(it)
    out = {}
    for name, totals in (("TRAINING_FULL_PRECISION", (B, 0, 0, 0)), ("TRAINING_QUANTIZED", (0, B, 0, 0)),
                         ("TRAINING_ENTROPY", (0, 0, B, 0)), ("STE_ENTROPY", (0, 0, 0, B))):
        (opt.full_precision_training_total, opt.quantized_training_total, opt.entropy_constrained_train_total,
         opt.ste_entropy_constrained_train_total) = totals
        for _ in range(warmup):
            it += 1
            trainer.step(it)
        active = torch.zeros((), device=dev, dtype=torch.float64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            it += 1
            active += trainer.step(it).active_gaussians
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out[name] = {"ms_per_step": 1e3 * el / steps, "gaussians_per_s": float(active.item()) / el,
                     "iterations_in_the_reference_schedule": share[name]}
    total = sum(share.values())
    out["schedule_weighted_ms_per_step"] = sum(out[n]["ms_per_step"] * share[n] for n in share) / total
    out["note"] = (f"{anchors} anchors, {H}x{W}, {T} frames, 16-frame slab; {pretrain} untimed full-precision steps, then per phase {warmup} warm-up + "
                   f"{steps} timed steps; early plan and late row gather in every phase, generation once per (frame, anchor) in the two phases without per-render noise")
    return out


# ------------------------------------------------------------------------------------------------ raster
def cpu_baseline_raster(sc, workload):
    """Oracle (CPU port of the same algorithm) on the host cores of this box, same scene, bounded sample."""
    import numpy as np
    import oracle
    oracle.build()
    s = sc["settings"]
    st = oracle.make_settings(s["H"], s["W"], s["x_min"], s["y_min"], s["scale"], s["threshold"], s["viewmatrix"],
                              bg=s["bg"], scale_modifier=s["scale_modifier"])
    cores = _all_cores()
    t0 = time.perf_counter()
    fwd = oracle.raster_forward(st, sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"],
                                num_threads=cores)
    t_f = time.perf_counter() - t0
    n_vis = int((fwd.radii > 0).sum())
    reps = 1
    while t_f < 10.0 and reps < 64:       # bounded sample: repeat the pass until ~10 s of CPU work have been timed
        t0 = time.perf_counter()
        oracle.raster_forward(st, sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"], num_threads=cores)
        t_f += time.perf_counter() - t0
        reps += 1
    sample = (f"{reps} forward passes of the same {s['H']}x{s['W']} scene ({sc['means3D'].shape[0]} Gaussians); preprocess, "
              f"per-tile sort and blend over {cores} OpenMP threads")
    units, t = n_vis * reps, t_f
    if workload == "raster_fwdbwd":      # one forward+backward pass = mean forward time + one scalar backward
        dL = np.ones((3, s["H"], s["W"]), np.float32)
        t0 = time.perf_counter()
        oracle.raster_backward(st, sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"], fwd, dL, num_threads=cores)
        units, t = n_vis, t_f / reps + (time.perf_counter() - t0)
        sample += f" (mean) + 1 backward pass on {cores} threads"
    _gpu_cores()
    return {"value": units / t, "unit": "Gaussians/s", "cores": cores, "kind": "port", "sample": sample, "seconds": round(t, 3)}


def run_raster(args, rank, world, dev, workload, cpu_baseline):
    """BASELINE.json configs[1] (and its forward+backward variant): rank r renders its own frame of the same video."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from gsvc_amd import _lib, rasterizer, synthetic

    H, W, T, P = args.height, args.width, args.frames, args.gaussians
    frame_id = T // 2 + rank
    sc = synthetic.raster_scene(P, H=H, W=W, T=T, seed=2026 + rank, window_frames=16, frame_id=frame_id, sigma_px=tuple(args.sigma_px))
    s = sc["settings"]
    rs = rasterizer.GaussianRasterizationSettings(
        image_height=H, image_width=W, x_min=s["x_min"], y_min=s["y_min"], scale=s["scale"], threshold=s["threshold"],
        bg=torch.zeros(3), scale_modifier=1.0, viewmatrix=torch.tensor(s["viewmatrix"]), sh_degree=0,
        campos=torch.tensor([0.0, 0.0, s["z_cam"]]), prefiltered=False, debug=False)
    cs = rasterizer.settings_to_c(rs)
    d = {k: torch.tensor(sc[k], device=dev) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    d["opacities"] = d["opacities"].view(-1).contiguous()
    dL = torch.randn(3, H, W, device=dev)

    # sizing pass (synchronising) -> instance capacity and the workload's counts
    _, radii, st0 = rasterizer.raster_forward(cs, d["means3D"], d["colors"], d["opacities"], d["scales"], d["rotations"])
    n_inst, _, n_vis, max_tile = st0.counters()
    cap = int(n_inst * 1.1) + 1024
    grads = [torch.empty(P, 3, device=dev), torch.empty(P, 3, device=dev), torch.empty(P, 3, device=dev),
             torch.empty(P, device=dev), torch.empty(P, 3, device=dev), torch.empty(P, 4, device=dev)]
    scratch = torch.empty(rasterizer.backward_scratch_floats(P, cap), device=dev)
    L = _lib.lib()

    def step():
        image, radii, st = rasterizer.raster_forward(cs, d["means3D"], d["colors"], d["opacities"], d["scales"],
                                                     d["rotations"], max_instances=cap, sync=False)
        if workload == "raster_fwdbwd":
            _lib.check(L.gsvc_raster_backward(
                C.byref(cs), P, cap, _lib.ptr(d["means3D"]), _lib.ptr(d["colors"]), _lib.ptr(d["opacities"]),
                _lib.ptr(d["scales"]), _lib.ptr(d["rotations"]), _lib.ptr(radii), _lib.ptr(st.geom), _lib.ptr(st.binning),
                _lib.ptr(st.image_state), _lib.ptr(dL), *[_lib.ptr(g) for g in grads], _lib.ptr(scratch),
                _lib.current_stream(dev)), "gsvc_raster_backward")
        return image

    def step_pair():
        return rasterizer.raster_forward(cs, d["means3D"], d["colors"], d["opacities"], d["scales"], d["rotations"],
                                         max_instances=cap, sync=False, pair=True)[0]

    for _ in range(args.warmup):
        step()
    elapsed = timed(torch, dist, world, step, args.steps)
    total_units, elapsed = reduce_sum_max(torch, dist, world, dev, float(n_vis) * args.steps, elapsed)

    # two-view frames (the reference's fps definition: view + opposite view + flip + average) from the fused pass
    for _ in range(3):
        step_pair()
    tp = timed(torch, dist, world, step_pair, args.steps)
    _, tp = reduce_sum_max(torch, dist, world, dev, 0.0, tp)
    pair_fps = args.steps * world / tp

    # the same two-view frames pipelined over two HIP streams (a decoder renders frame after frame: the latency-bound
    # binning kernels of frame i+1 overlap the compositing of frame i); reported beside, never instead of, the
    # single-stream numbers
    pipelined_fps = pipelined_single_fps = None
    if workload == "raster_fwd":
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        k = [0]

        def pipelined():
            k[0] += 1
            with torch.cuda.stream(streams[k[0] & 1]):
                step_pair()
        for _ in range(4):
            pipelined()
        tq = timed(torch, dist, world, pipelined, args.steps)
        _, tq = reduce_sum_max(torch, dist, world, dev, 0.0, tq)
        pipelined_fps = args.steps * world / tq

        def pipelined_single():
            k[0] += 1
            with torch.cuda.stream(streams[k[0] & 1]):
                step()
        for _ in range(4):
            pipelined_single()
        ts = timed(torch, dist, world, pipelined_single, args.steps)
        _, ts = reduce_sum_max(torch, dist, world, dev, 0.0, ts)
        pipelined_single_fps = args.steps * world / ts

    # the same step replayed from a captured HIP graph (every C-ABI call is capturable: no allocation, no synchronisation inside):
    # what the launch gaps between the pipeline's dependent kernels cost.  Reported beside the eager numbers.
    graph_ms = None
    try:
        side = torch.cuda.Stream(device=dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            step()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side):
                graph_image = step()
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        tg = timed(torch, dist, world, g.replay, args.steps)
        _, tg = reduce_sum_max(torch, dist, world, dev, 0.0, tg)
        graph_ms = 1e3 * tg / args.steps
        if not torch.equal(graph_image, step()):
            graph_ms = None          # a replay must give the eager image bit for bit
        del g, graph_image
    except Exception as e:          # capture is an extra: never fail the bench line over it
        print(f"[bench] graph capture skipped: {e}", file=sys.stderr)

    _lib.profile_enable(True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    _lib.profile_enable(False)
    if rank != 0:
        return None
    HW = H * W
    alg = {  # algorithmic bytes per launch (SURVEY.md section 8d / BASELINE.md section 3)
        "k_preprocess": 60 * P + 44 * n_vis, "k_blend": 40 * n_inst + 20 * HW, "k_blend_bwd": 40 * n_inst + 20 * HW,
        "k_gaussian_bwd": 88 * n_vis + 124 * P,
    }
    kern = {k: {"launches": n, "avg_us": 1e3 * ms / max(n, 1)} for k, (n, ms) in prof.items()}
    dom = max(kern, key=lambda k: kern[k]["avg_us"] * kern[k]["launches"])
    dom_bytes = alg.get(dom, 0)
    achieved = dom_bytes / (kern[dom]["avg_us"] * 1e-6) / 1e9 if dom_bytes else 0.0
    pipe_bytes = 60 * P + 44 * n_vis + 40 * n_inst + 20 * HW
    if workload == "raster_fwdbwd":
        pipe_bytes += 40 * n_inst + 20 * HW + 88 * n_vis + 124 * P
    kernel_us = sum(v["avg_us"] * v["launches"] for v in kern.values()) / args.steps
    traffic, traffic_src = pmc_traffic(workload, dom)
    cfg_name = ("BASELINE.json configs[1]" if (H, W, P, tuple(args.sigma_px)) == (1080, 1920, 200_000, (0.5, 4.0))
                else f"raster set of BASELINE.md section 2, sigma {args.sigma_px[0]}..{args.sigma_px[1]} px")
    out = {
        "metric": METRIC, "value": total_units / elapsed, "unit": "Gaussians/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{workload} ({cfg_name}): {H}x{W} single frame of a {T}-frame cube, {P} Gaussians in a "
                               f"16-frame z-slab; frames sharded 1 per rank",
                   "gaussians": P, "visible": n_vis, "instances": n_inst, "max_tile_list": max_tile,
                   "parallelism": f"frame-shard x{world}"},
        "rccl_ranks": (dist.get_world_size() if world > 1 else 1),
        "render_fps": args.steps * world / elapsed,
        "graph_replay": (None if graph_ms is None else
                         {"ms_per_step": graph_ms, "render_fps": 1e3 * world / graph_ms,
                          "note": "the same step captured once into a HIP graph and replayed (image bit-identical to the eager step)"}),
        "render_fps_two_view": pair_fps,
        "render_fps_two_view_2streams": pipelined_fps,
        "render_fps_2streams": pipelined_single_fps,
        "render_fps_note": PAIR_NOTE,
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "note": "this kernel's binding unit is the vector ALU, not HBM: traffic_source.valu holds the SQ counters of the "
                             "same command (share of the kernel's cycles its SIMDs spent executing vector instructions)",
                     "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_us": kern[dom]["avg_us"]},
        "roofline_pipeline": {"algorithmic_bytes_per_step": pipe_bytes, "kernel_us_per_step": kernel_us,
                              "achieved": pipe_bytes / (kernel_us * 1e-6) / 1e9, "unit": "GB/s",
                              "achieved_on_wall_ms_per_step": pipe_bytes / (elapsed / args.steps) / 1e9},
        "kernels": {k: {"avg_us": round(v["avg_us"], 2), "launches_per_step": v["launches"] / args.steps}
                    for k, v in kern.items()},
    }
    if cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_raster(sc, workload)
    del d, dL, grads, scratch
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("GSVC_HANG_DUMP"):        # diagnostics: every thread's stack to stderr after that many seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["GSVC_HANG_DUMP"]), exit=True)
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with --nproc-per-node {args.gpus}\n")
        sys.exit(2)
    import torch
    if os.environ.get("GSVC_SHARE_GPU"):        # test knob: every rank on device 0 (needs GSVC_DIST_BACKEND=gloo; RCCL wants one GPU per rank)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # one rank per GPU, on the CPU cores of that GPU's NUMA node (gsvc_amd/hostbind.py: the small-shape steps are host-bound and
    # 10-15 % slower from the far socket); the CPU baseline legs take the whole machine back (_all_cores)
    global _AFFINITY0, _AFFINITY_GPU
    _AFFINITY0 = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    from gsvc_amd.hostbind import bind_to_device
    if bind_to_device(local_rank):
        _AFFINITY_GPU = os.sched_getaffinity(0)
    if _dp_on(world):
        import torch.distributed as dist
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29571"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)          # a forced one-rank group outside torch.distributed.run
        backend = os.environ.get("GSVC_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        from gsvc_amd.dist import log_ranks
        log_ranks()
    cpu = world == 1 and not args.no_cpu_baseline
    if args.workload in ("headline", "train_step"):
        res = run_train_step(args, rank, world, dev)
        if args.workload == "headline" and not args.no_side and world == 1:
            # second term of the metric: render fps at 1080p on BASELINE.json configs[1]; a failure here must not take
            # the train-step line down
            try:
                import copy
                a2 = copy.copy(args)
                a2.steps, a2.warmup = max(args.steps, 100), max(args.warmup, 10)
                torch.cuda.empty_cache()
                side = run_raster(a2, rank, world, dev, "raster_fwd", cpu_baseline=False)
            except Exception as e:  # noqa: BLE001
                side = {"error": f"{type(e).__name__}: {e}"}
            if rank == 0:
                keep = ("value", "unit", "ms_per_step", "steps", "config", "render_fps", "render_fps_two_view",
                        "render_fps_two_view_2streams", "render_fps_2streams", "graph_replay", "render_fps_note", "roofline", "roofline_pipeline", "kernels", "error")
                res["raster_fwd"] = {k: side[k] for k in keep if k in side}
                if "render_fps" in side:
                    res["render_fps"] = side["render_fps"]
                    res["render_fps_two_view"] = side["render_fps_two_view"]
            # the headline's shape with a LIVE fit (the default mode's float atomics order the fit's sums differently run by run: the
            # model, and with it the active count, differs between runs): what rounds 1-5 reported as the headline, kept as a side entry
            if rank == 0 and not os.environ.get("GSVC_BENCH_NO_LIVE"):
                try:
                    torch.cuda.empty_cache()
                    res["train_step_live_fit"] = run_train_step_light(args, dev, anchors=args.anchors, steps=30, pretrain=args.pretrain)
                    res["train_step_live_fit"]["note"] = ("same shape as the headline, untimed fit in the DEFAULT mode: active_per_render moves between "
                                                          "~135 k and ~175 k from run to run (the fit is chaotic), the frozen headline scene does not")
                except Exception as e:  # noqa: BLE001
                    res["train_step_live_fit"] = {"error": f"{type(e).__name__}: {e}"}
            # second operating point: ~500 k ACTIVE Gaussians per render (the headline has ~500 k submitted, ~29 % of them active)
            if rank == 0 and not os.environ.get("GSVC_BENCH_NO_500K"):
                try:
                    torch.cuda.empty_cache()
                    res["train_step_500k_active"] = run_train_step_light(args, dev, anchors=870_000, steps=10, pretrain=40)
                except Exception as e:  # noqa: BLE001
                    res["train_step_500k_active"] = {"error": f"{type(e).__name__}: {e}"}
            # a step late in a fit (Gaussians grown to tens of tiles each): where most of a 40 000-iteration fit's time goes
            if rank == 0 and not os.environ.get("GSVC_BENCH_NO_LATE"):
                try:
                    torch.cuda.empty_cache()
                    res["train_step_late"] = run_train_step_late(args, dev)
                except Exception as e:  # noqa: BLE001
                    res["train_step_late"] = {"error": f"{type(e).__name__}: {e}"}
            # the other phases of the schedule (the headline is the TRAINING_ENTROPY phase: 20 k of the 40 k iterations)
            if rank == 0 and not os.environ.get("GSVC_BENCH_NO_PHASES"):
                try:
                    torch.cuda.empty_cache()
                    res["train_step_by_phase"] = run_train_step_phases(args, dev, anchors=args.anchors)
                except Exception as e:  # noqa: BLE001
                    res["train_step_by_phase"] = {"error": f"{type(e).__name__}: {e}"}
            # stream_decode fps at the BASELINE.json configs[4] shape (4K, ~2 M Gaussians per frame); same rule for failures
            if rank == 0 and not os.environ.get("GSVC_BENCH_NO_4K"):
                try:
                    del side
                    torch.cuda.empty_cache()
                    res["stream_decode_4k"] = run_stream_decode(args, dev)
                except Exception as e:  # noqa: BLE001
                    res["stream_decode_4k"] = {"error": f"{type(e).__name__}: {e}"}
    elif args.workload == "stream_decode":
        sd = run_stream_decode(args, dev)
        res = {"metric": "stream_decode fps (entropy decode of the whole model + two-view frames)", "value": sd["stream_decode_fps"],
               "unit": "frames/s", "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": sd["decode_ms"], "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": sd["workload"]}, **sd}
    else:
        res = run_raster(args, rank, world, dev, args.workload, cpu_baseline=cpu)
    if rank == 0:
        # a line that claims N GPUs must come from N ranks of one process group (RCCL unless the test knob says otherwise)
        if res.get("n_gpus") != args.gpus or (world > 1 and res.get("rccl_ranks") != args.gpus):
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but the line reports n_gpus={res.get('n_gpus')} rccl_ranks={res.get('rccl_ranks')}\n")
            print(json.dumps(res), flush=True)
            sys.exit(3)
        print(json.dumps(res), flush=True)
    if _dp_on(world):
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
