"""Worker of tests/test_train_gpu.py::test_two_ranks_through_every_phase_at_the_headline_shape: data-parallel ranks through the four
phases of the schedule at BASELINE.json configs[2] size (where the early plan from inside the backward and the once-per-frame
generation are in use): no deadlock, finite losses, replicas stay identical.  Launch:  GSVC_DIST_BACKEND=gloo GSVC_SHARE_GPU=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tests/_dp_phases_worker.py"""
import os, sys, time
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
backend = os.environ.get("GSVC_DIST_BACKEND", "nccl")
local = 0 if os.environ.get("GSVC_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev) if backend == "nccl" else dist.init_process_group(backend)
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
from gsvc_amd.train import Trainer
from gsvc_amd import dist as gdist
mp_, opt, pipe = cfg_20240919()
cube = SyntheticFrameCube(1080, 1920, 64, seed=1234, device=dev).materialize()
mp_.threshold = 8.0 / cube.scale
opt.start_stat, opt.update_until, opt.pause_densification, opt.update_from = 0, 10 ** 9, 0, 10 ** 9
B = 10 ** 9
(opt.full_precision_training_total, opt.quantized_training_total, opt.entropy_constrained_train_total,
 opt.ste_entropy_constrained_train_total) = B, 0, 0, 0
torch.manual_seed(0); np.random.seed(0)
pc = GaussianModel(mp_, mp_.anchor_feature_dim, mp_.n_offsets, mp_.voxel_size, mp_.update_depth, mp_.update_init_factor,
                   mp_.update_hierarchy_factor, mp_.use_feat_bank, n_features_per_level=mp_.grid_feature_dim,
                   log2_hashmap_size=mp_.log2, log2_hashmap_size_2D=mp_.log2_2D, device=dev)
rng = np.random.default_rng(0)
lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
pc.create_from_points(rng.uniform(lim, -lim, (245_000, 3)), spatial_lr_scale=1.0)
pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
pc.training_setup(opt)
gdist.broadcast_parameters(pc)
tr = Trainer(pc, cube, opt, pipe, mp_, seed=0)
it = 0
STEPS = int(os.environ.get("GSVC_DP_PHASE_STEPS", "4"))
for name, totals in (("FULL", (B, 0, 0, 0)), ("QUANT", (0, B, 0, 0)), ("ENTROPY", (0, 0, B, 0)), ("STE", (0, 0, 0, B))):
    (opt.full_precision_training_total, opt.quantized_training_total, opt.entropy_constrained_train_total,
     opt.ste_entropy_constrained_train_total) = totals
    e0 = getattr(tr, "early_steps", 0)
    t0 = time.perf_counter()
    losses = []
    for _ in range(STEPS):
        it += 1
        losses.append(float(tr.step(it).loss))
    torch.cuda.synchronize()
    tr.sync_replicas()      # GSVC_DP_ZOWN=1: a replica is fresh inside its block + halo only until it is made whole (no-op otherwise)
    # replicas identical: the checksum of every parameter agrees across ranks
    cs = torch.stack([p.detach().double().sum() for p in pc.parameters()])
    lo, hi = cs.clone(), cs.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    same = bool(torch.equal(lo, hi))
    if dist.get_rank() == 0:
        print(f"DP_PHASE {name}: losses {['%.4f' % l for l in losses]} early_steps {getattr(tr, 'early_steps', 0) - e0} replicas_identical {same} "
              f"sparse {tr.reducer._sparse is not None} zown {tr._zown is not None} checksum {float(cs.sum()).hex()} "
              f"{(time.perf_counter() - t0) / STEPS:.2f} s/step", flush=True)
    if dist.get_rank() == 0 and os.environ.get("GSVC_DP_PHASE_CS"):
        print("DP_CS " + name + " " + " ".join(f"{n}={float(v).hex()}" for (n, _), v in zip(pc.named_parameters(), cs)), flush=True)
    assert all(np.isfinite(losses)) and same
if dist.get_rank() == 0:
    print("DP_PHASES_OK", flush=True)
dist.barrier()
dist.destroy_process_group()
