#!/usr/bin/env python3
"""bench.py — headline benchmark of the GSVC hot path on MI355X (contract: see the task statement).

    python bench.py --gpus N --steps K --warmup W [--workload raster_fwd|raster_fwdbwd]

A "step" is one pass of the hot path over one synthetic UVG-shaped 1080p frame resident in HBM.
  raster_fwd     BASELINE.json configs[1]: 1080p single frame, 200k Gaussians, forward raster only
  raster_fwdbwd  same scene, forward + backward of the rasterizer (dL/dimage random)
N > 1: launched by torch.distributed.run, one rank per GPU; frames shard across ranks (each rank rasterizes
its own frame of the same video; no data-path collective in these workloads) -> weak scaling.

One JSON line on rank 0.  `value` = Gaussians rasterized per second, whole job (sum over ranks of Gaussians
with radius > 0 per step, x steps, / max-over-ranks wall time).  `roofline` is for the dominant kernel,
timed with HIP events on its own stream inside this process (second pass of K steps with the library's
per-kernel event hooks on).  `cpu_baseline` = the CPU oracle (a port, not the product) on the same scene.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="raster_fwd", choices=["raster_fwd", "raster_fwdbwd"])
    ap.add_argument("--gaussians", type=int, default=200_000)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(sc, workload):
    """Oracle (CPU port of the same algorithm) on the host cores of this box, same scene, one pass."""
    import oracle
    oracle.build()
    s = sc["settings"]
    st = oracle.make_settings(s["H"], s["W"], s["x_min"], s["y_min"], s["scale"], s["threshold"], s["viewmatrix"],
                              bg=s["bg"], scale_modifier=s["scale_modifier"])
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    fwd = oracle.raster_forward(st, sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"],
                                num_threads=cores)
    t_f = time.perf_counter() - t0
    n_vis = int((fwd.radii > 0).sum())
    sample = f"1 forward pass of the same {s['H']}x{s['W']} scene ({sc['means3D'].shape[0]} Gaussians), OpenMP blend"
    t = t_f
    used = cores
    if workload == "raster_fwdbwd":
        dL = np.ones((3, s["H"], s["W"]), np.float32)
        t0 = time.perf_counter()
        oracle.raster_backward(st, sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"], fwd, dL)
        t += time.perf_counter() - t0
        sample += " + 1 scalar backward pass"
    return {"value": n_vis / t, "unit": "Gaussians/s", "cores": used, "kind": "port", "sample": sample,
            "seconds": round(t, 3)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    from gsvc_amd import _lib, rasterizer, synthetic

    H, W, T, P = args.height, args.width, args.frames, args.gaussians
    # frames shard across ranks: rank r renders its own frame of the same video
    frame_id = T // 2 + rank
    sc = synthetic.raster_scene(P, H=H, W=W, T=T, seed=2026 + rank, window_frames=16, frame_id=frame_id)
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
    scratch = torch.empty(P * 16, device=dev)
    L = _lib.lib()
    import ctypes as C

    def step():
        image, radii, st = rasterizer.raster_forward(cs, d["means3D"], d["colors"], d["opacities"], d["scales"],
                                                     d["rotations"], max_instances=cap, sync=False)
        if args.workload == "raster_fwdbwd":
            _lib.check(L.gsvc_raster_backward(
                C.byref(cs), P, cap, _lib.ptr(d["means3D"]), _lib.ptr(d["colors"]), _lib.ptr(d["opacities"]),
                _lib.ptr(d["scales"]), _lib.ptr(d["rotations"]), _lib.ptr(radii), _lib.ptr(st.geom), _lib.ptr(st.binning),
                _lib.ptr(st.image_state), _lib.ptr(dL), *[_lib.ptr(g) for g in grads], _lib.ptr(scratch),
                _lib.current_stream(dev)), "gsvc_raster_backward")
        return image

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    units = torch.tensor([float(n_vis) * args.steps, elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        import torch.distributed as dist
        tmax = units[1:2].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = units[0:1].clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed, total_units = float(tmax.item()), float(tot.item())
    else:
        total_units = float(units[0].item())

    # per-kernel pass: same K steps with HIP events around every launch on the launch stream
    _lib.profile_enable(True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    _lib.profile_enable(False)

    if rank == 0:
        HW = H * W
        alg = {  # algorithmic bytes per launch (SURVEY.md section 8d / BASELINE.md section 3)
            "k_preprocess": 60 * P + 44 * n_vis,
            "k_blend": 40 * n_inst + 20 * HW,
            "k_blend_bwd": 40 * n_inst + 20 * HW,
            "k_gaussian_bwd": 88 * n_vis + 124 * P,
        }
        kern = {k: {"launches": n, "avg_us": 1e3 * ms / max(n, 1)} for k, (n, ms) in prof.items()}
        dom = max(kern, key=lambda k: kern[k]["avg_us"] * kern[k]["launches"])
        dom_bytes = alg.get(dom, 0)
        achieved = dom_bytes / (kern[dom]["avg_us"] * 1e-6) / 1e9 if dom_bytes else 0.0
        pipe_bytes = 60 * P + 44 * n_vis + 40 * n_inst + 20 * HW
        if args.workload == "raster_fwdbwd":
            pipe_bytes += 40 * n_inst + 20 * HW + 88 * n_vis + 124 * P
        kernel_us = sum(v["avg_us"] * v["launches"] for v in kern.values()) / args.steps
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            try:
                traffic = json.load(open(pmc_path)).get(args.workload, {}).get(dom)
            except Exception:
                traffic = None
        out = {
            "metric": "train-step Gaussians/sec + render fps @1080p",
            "value": total_units / elapsed,
            "unit": "Gaussians/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {H}x{W} single frame of a {T}-frame cube, {P} Gaussians in a "
                                   f"16-frame z-slab (BASELINE.json configs[1]); frames sharded 1 per rank",
                       "gaussians": P, "visible": n_vis, "instances": n_inst, "max_tile_list": max_tile,
                       "parallelism": f"frame-shard x{world}"},
            "render_fps": args.steps * world / elapsed,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_us": kern[dom]["avg_us"]},
            "roofline_pipeline": {"algorithmic_bytes_per_step": pipe_bytes, "kernel_us_per_step": kernel_us,
                                  "achieved": pipe_bytes / (kernel_us * 1e-6) / 1e9, "unit": "GB/s"},
            "kernels": {k: {"avg_us": round(v["avg_us"], 2), "launches_per_step": v["launches"] / args.steps}
                        for k, v in kern.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, args.workload)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
