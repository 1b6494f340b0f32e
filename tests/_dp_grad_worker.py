"""Worker of tests/test_train_gpu.py::test_two_rank_step_averages_gradients (run under torch.distributed.run, 2 ranks):
one data-parallel fitting step — each rank on a frame pair of its own shard, gradients exchanged by the overlapped
reducer (gsvc_amd/dist.py) — must leave on every rank the MEAN of the two single-process gradients of those two pairs.
Backend from GSVC_DIST_BACKEND: "nccl" (= RCCL, one GPU per rank) or "gloo" with GSVC_SHARE_GPU=1 (both ranks on device 0).
GSVC_DP_FORCE=1 with ONE rank: the same collectives as the identity on a one-rank communicator (RCCL on a single-GPU box).
GSVC_DP_ZOWN=1: the per-anchor tensors are owned by z-range — their gradients are compared on the rows this rank owns."""
import faulthandler
import os as _os
faulthandler.dump_traceback_later(int(_os.environ.get("GSVC_HANG_DUMP", "300")), exit=True)      # a deadlocked rank prints its stacks and exits
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _setup_configs3():
    import numpy as np
    from gsvc_amd.arguments import cfg_20240919
    from gsvc_amd.frame import SyntheticFrameCube
    from gsvc_amd.model import GaussianModel
    from gsvc_amd.train import Trainer
    mp_, opt, pipe = cfg_20240919()
    assert mp_.threshold == 0.05 and opt.init_anchor_num == 100_000
    cube = SyntheticFrameCube(1080, 1920, 600, seed=1234, device="cuda")
    torch.manual_seed(0)
    pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                       mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                       log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device="cuda")
    rng = np.random.default_rng(0)
    lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
    pc.create_from_points(rng.uniform(lim, -lim, (opt.init_anchor_num, 3)), spatial_lr_scale=1.0)
    pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
    return pc, cube, opt, pipe, mp_, Trainer


def main():
    backend = os.environ.get("GSVC_DIST_BACKEND", "nccl")
    local = 0 if os.environ.get("GSVC_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    from test_train_gpu import _setup
    from gsvc_amd import dist as gd
    if os.environ.get("GSVC_DP_SHAPE") == "configs3":
        # BASELINE.json configs[3]: reference cfgs/cfg_20240919.yaml as is (100 000 anchors, 600 frames of 1080p, threshold .05)
        pc, cube, opt, pipe, mp, Trainer = _setup_configs3()
    else:
        pc, cube, opt, pipe, mp, Trainer = _setup(anchors=3000)
    opt.full_precision_training_total = 1000          # no quantisation noise: the step is a deterministic function of the frames
    if os.environ.get("GSVC_DP_MODE") == "ste":
        # the entropy-constrained loss (rate over EVERY visible anchor, hash-table bits, the mask regulariser that touches every
        # row of _mask) in its deterministic form: STE quantisation, no sampling
        import gsvc_amd.generate as G
        G.SAMPLE_RATE = 2.0
        opt.full_precision_training_total = opt.quantized_training_total = opt.entropy_constrained_train_total = 0
        opt.ste_entropy_constrained_train_total = 1000
    pc.training_setup(opt)
    gd.broadcast_parameters(pc)
    tr = Trainer(pc, cube, opt, pipe, mp, seed=3)
    captured = {}

    per_anchor = {}

    def capture_instead_of_update():
        captured.clear()
        for g in pc.optimizer.param_groups:
            for i, p in enumerate(g["params"]):
                if p.grad is not None:
                    captured[f"{g['name']}.{i}"] = p.grad.detach().clone()
                    if any(p is getattr(pc, n) for n in gd.PER_ANCHOR):
                        per_anchor[f"{g['name']}.{i}"] = True

    def rows(k, t):
        """What the exchange promises of tensor k: under z-range ownership the rows this rank owns."""
        return t[tr._zown.own_idx] if (tr._zown is not None and k in per_anchor) else t
    pc.optimizer.step = capture_instead_of_update      # parameters stay as they are: every step below sees the same model

    mine = torch.tensor([tr.lo], device="cuda" if backend == "nccl" else "cpu")
    los = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(los, mine)
    los = [int(t.item()) for t in los]
    assert len(set(los)) == len(los), los               # every rank has its own frames

    tr.step(1, frame_idx=tr.lo)
    dp = dict(captured)
    assert len(dp) >= 8, sorted(dp)

    tr.reducer.enabled = False                          # single-process reference: the same pairs one after the other
    ref = {}
    for f in los:
        tr.step(1, frame_idx=f)
        for k, v in captured.items():
            ref[k] = ref.get(k, 0) + v / len(los)
    worst = 0.0
    for k, v in ref.items():
        scale = float(v.abs().max())
        if scale == 0.0:
            assert float(rows(k, dp[k]).abs().max()) == 0.0, k
            continue
        err = float((rows(k, dp[k]) - rows(k, v)).abs().max()) / scale
        worst = max(worst, err)
        assert err < 2e-4, (k, err, scale)
    # a second data-parallel step, this one on the frame pair the trainer drew itself with the step plan it queued at the end of
    # the last step: the per-anchor gradients then travel as (row index, row) lists of the rank's distinct visible anchors
    # (GradReducer.set_sparse) instead of dense all-reduces — same mean of the ranks' gradients
    tr.reducer.enabled = True
    planned = tr._plan is not None and tr._plan_idx is not None
    tr.step(2)
    dp2 = dict(captured)
    sparse_used = tr.reducer._sparse is not None
    mine = torch.tensor([tr._last_idx], device=mine.device)
    idxs = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(idxs, mine)
    idxs = [int(t.item()) for t in idxs]
    tr.reducer.enabled = False
    ref2 = {}
    for f in idxs:
        tr.step(2, frame_idx=f)
        for k, v in captured.items():
            ref2[k] = ref2.get(k, 0) + v / len(idxs)
    worst2 = 0.0
    for k, v in ref2.items():
        scale = float(v.abs().max())
        if scale == 0.0:
            assert float(rows(k, dp2[k]).abs().max()) == 0.0, k
            continue
        err = float((rows(k, dp2[k]) - rows(k, v)).abs().max()) / scale
        worst2 = max(worst2, err)
        assert err < 2e-4, ("planned step", k, err, scale)
    if dist.get_rank() == 0:
        print(f"DP_GRAD_OK backend={dist.get_backend()} ranks={dist.get_world_size()} tensors={len(ref)} worst={worst:.2e} "
              f"planned={planned} sparse={sparse_used} worst_planned={worst2:.2e} bytes_sent={tr.reducer.bytes_sent} "
              f"zown={tr._zown is not None} zown_bytes={tr._zown.bytes_sent if tr._zown is not None else 0} per_anchor={len(per_anchor)}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
