"""Worker of tests/test_train_gpu.py::test_two_ranks_keep_identical_anchors_through_densification (run under
torch.distributed.run, 2 ranks on device 0, gloo): fitting steps across several adjust_anchor calls; every rank must end
with the same anchors, features and Adam moments although each one steps on its own frames."""
import faulthandler
import os as _os
faulthandler.dump_traceback_later(int(_os.environ.get("GSVC_HANG_DUMP", "300")), exit=True)      # a deadlocked rank prints its stacks and exits
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from test_train_gpu import _setup
    from gsvc_amd import dist as gd
    pc, cube, opt, pipe, mp, Trainer = _setup(anchors=3000)
    opt.full_precision_training_total = 1000
    opt.start_stat, opt.update_from, opt.update_interval, opt.update_until, opt.pause_densification = 2, 6, 5, 40, 0
    opt.densify_grad_threshold, opt.success_threshold = 1e-7, 0.2
    pc.training_setup(opt)
    gd.broadcast_parameters(pc)
    tr = Trainer(pc, cube, opt, pipe, mp, seed=3)
    a0 = pc._anchor.shape[0]
    counts = []
    for it in range(1, 24):
        out = tr.step(it)                      # frame pairs from the rank's own shard
        assert np.isfinite(float(out.loss))
        counts.append(pc._anchor.shape[0])
    torch.cuda.synchronize()
    st = pc.optimizer.state[pc._anchor_feat]
    sig = torch.tensor([float(pc._anchor.shape[0]), float(pc._anchor.double().sum()), float(pc._anchor_feat.double().abs().sum()),
                        float(pc._offset.double().abs().sum()), float(st["exp_avg_sq"].double().sum()),
                        float(pc.mlp_opacity.linear1.weight.double().abs().sum())], dtype=torch.float64)
    gathered = [torch.zeros_like(sig) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, sig)
    if dist.get_rank() == 0:
        assert len(set(counts)) > 1 and counts[-1] != a0, counts
        for g in gathered[1:]:
            assert torch.allclose(g, gathered[0], rtol=1e-9, atol=0.0), (gathered[0].tolist(), g.tolist())
        print("DP_DENSIFY_OK", counts[0], "->", counts[-1], flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
