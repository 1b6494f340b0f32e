"""Worker of tests/test_train_gpu.py::test_four_rank_dry_run_at_the_configs3_shape (run under torch.distributed.run; the ranks share
device 0 over gloo — this pool allows at most 6 processes on a card, so the 8-rank form of BASELINE.json configs[3] is rehearsed
with 4): the reference's own configuration (cfgs/cfg_20240919.yaml: 100 k anchors, 600-frame 1080p cube, threshold .05, lambda
.004), frames sharded over the ranks, 20 TRAINING_ENTROPY steps that contain one anchor densification and one step every rank
repeats because ONE rank's rasterizer instance buffer overflowed.  Rank 0 prints, per step, the bytes it handed to the collectives
and whether the per-anchor gradients went as rows or dense; at the end every rank must hold the same parameters and anchors."""
import faulthandler
import os as _os
faulthandler.dump_traceback_later(int(_os.environ.get("GSVC_HANG_DUMP", "500")), exit=True)      # a deadlocked rank prints its stacks and exits
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from gsvc_amd import dist as gd
    rank, world, _ = gd.init_from_env(os.environ.get("GSVC_DIST_BACKEND", "gloo"))
    import gsvc_amd.rasterizer as RZ
    from gsvc_amd.arguments import cfg_20240919
    from gsvc_amd.frame import SyntheticFrameCube
    from gsvc_amd.model import GaussianModel
    from gsvc_amd.train import Trainer
    mp_, opt, pipe = cfg_20240919()
    cube = SyntheticFrameCube(1080, 1920, 600, seed=1234, device=dev)          # frames are generated on first use (a rank touches ~40)
    opt.full_precision_training_total, opt.quantized_training_total = 0, 0
    opt.entropy_constrained_train_total = 10 ** 9
    opt.start_stat, opt.update_from, opt.update_interval, opt.update_until, opt.pause_densification = 0, 5, 10, 10 ** 9, 0
    opt.densify_grad_threshold, opt.success_threshold = 1e-7, 0.2                # so that the one adjust_anchor call does grow anchors
    torch.manual_seed(0)
    np.random.seed(0)
    pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                       mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                       log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device=dev)
    lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
    pc.create_from_points(np.random.default_rng(0).uniform(lim, -lim, (100_000, 3)), spatial_lr_scale=1.0)
    pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
    pc.training_setup(opt)
    gd.broadcast_parameters(pc)
    tr = Trainer(pc, cube, opt, pipe, mp_, seed=0)
    assert (tr.lo, tr.hi) == gd.frame_shard(600)
    # step 14 on the LAST rank only: a 200-instance buffer for its four forwards -> every rank repeats the step
    real, small = RZ.raster_forward, [0]

    def tiny(cs, *a, **k):
        if small[0] > 0 and k.get("max_instances") is None:
            small[0] -= 1
            k["max_instances"] = 200
        return real(cs, *a, **k)
    RZ.raster_forward = tiny
    import gsvc_amd.ortho_gaussian_renderer.renderer as RR
    if hasattr(RR, "raster_forward"):
        RR.raster_forward = tiny
    a0 = int(pc._anchor.shape[0])
    log = []
    for it in range(1, 21):
        if it == 14 and rank == world - 1:
            small[0] = 4
        out = tr.step(it)
        assert np.isfinite(float(out.loss)), it
        zb = int(tr._zown.bytes_sent) if tr._zown is not None else 0          # GSVC_DP_ZOWN=1: halo rows of gradients out + parameter rows back
        log.append((it, int(tr.reducer.bytes_sent) + zb, "owned" if tr._zown is not None else tr.reducer._sparse is not None,
                    int(pc._anchor.shape[0]), int(getattr(tr, "repeated_steps", 0))))
    torch.cuda.synchronize()
    tr.sync_replicas()          # z-range ownership: a replica is whole again only after this (no-op otherwise)
    sig = torch.tensor([float(pc._anchor.shape[0])] + [float(p.detach().double().sum()) for p in pc.parameters()], dtype=torch.float64)
    lo, hi = sig.clone(), sig.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    rep = torch.tensor([float(getattr(tr, "repeated_steps", 0))], dtype=torch.float64)
    dist.all_reduce(rep, op=dist.ReduceOp.MIN)
    if rank == 0:
        for it, sent, sparse, anchors, reps in log:
            print(f"DRYRUN step {it}: {sent} bytes to the collectives, per-anchor gradients {'to their z-range owners' if sparse == 'owned' else 'as rows' if sparse else 'dense'}, "
                  f"{anchors} anchors, repeated steps so far {reps}", flush=True)
        assert torch.equal(lo, hi), "replicas differ"
        assert log[-1][3] != a0, "adjust_anchor did not change the anchor set"
        assert int(rep.item()) >= 1, "the overflowing step was not repeated on every rank"
        print(f"DP_DRYRUN_OK ranks={world} anchors {a0} -> {log[-1][3]} repeated={int(rep.item())}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
