import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsvc_amd import loss_utils as LU
torch.manual_seed(0)
f = torch.rand(3, 1080, 1920, device="cuda", requires_grad=True); b = torch.rand(3, 1080, 1920, device="cuda", requires_grad=True)
g = torch.rand(3, 1080, 1920, device="cuda")
def run():
    s, l, a = LU.ssim_l1_pair(f, b, g)
    (s + l).backward()
for _ in range(5): run()
torch.cuda.synchronize()
from gsvc_amd import _lib
_lib.profile_enable(True)
for _ in range(20): run()
torch.cuda.synchronize()
for k, (n, ms) in _lib.profile_collect().items():
    print(k, n, f"{1e3 * ms / n:.1f} us")
