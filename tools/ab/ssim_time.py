"""Kernel durations of the fused SSIM + L1 pair loss (csrc/ssim.hip) at 1080p, forward and backward, HIP events on the launch stream
(gsvc_profile_enable).  GSVC_LIB_PATH selects a variant library for an A/B on the same box:  python tools/ab/ssim_time.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsvc_amd import _lib
from gsvc_amd.loss_utils import ssim_l1_pair

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g = torch.Generator(device="cuda").manual_seed(0)
f = torch.rand(3, 1080, 1920, device="cuda", generator=g, requires_grad=True)
b = torch.rand(3, 1080, 1920, device="cuda", generator=g, requires_grad=True)
gt = torch.rand(3, 1080, 1920, device="cuda", generator=g)
for _ in range(5):
    s, l, _ = ssim_l1_pair(f, b, gt)
    (s + l).backward()
torch.cuda.synchronize()
_lib.profile_enable(True)
for _ in range(reps):
    s, l, _ = ssim_l1_pair(f, b, gt)
    (s + l).backward()
torch.cuda.synchronize()
prof = _lib.profile_collect()
_lib.profile_enable(False)
print(os.environ.get("GSVC_LIB_PATH", "default library"), {k: round(1e3 * ms / max(n, 1), 2) for k, (n, ms) in prof.items() if "ssim" in k},
      "checksum", float(s), float(l), float(f.grad.double().sum()))
