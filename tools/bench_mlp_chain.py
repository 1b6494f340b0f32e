"""Times the generation MLPs (3 GeneratorNets + mlp_deform) forward + backward at M rows: whole-network chain kernels
(gsvc_amd.mlp.generate_all, csrc/mlp_chain.hip) against the layer-by-layer path, with the library's per-kernel events."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gsvc_amd import _lib, mlp
from gsvc_amd.model import GeluSequential, GeneratorNet, Linear

M = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
torch.manual_seed(0)
gens = [GeneratorNet(50, 10, 100, 66, out_act=torch.nn.Tanh()).cuda(), GeneratorNet(50, 30, 100, 66, out_act=torch.nn.Sigmoid()).cuda(),
        GeneratorNet(50, 70, 100, 66).cuda()]
deform = GeluSequential(Linear(116, 100), torch.nn.GELU(), Linear(100, 100), torch.nn.GELU(), Linear(100, 100), torch.nn.GELU(),
                        Linear(100, 100), torch.nn.GELU(), Linear(100, 30)).cuda()
lin = list(deform)[0::2]
feat = (torch.randn(M, 50, device="cuda") * 2).requires_grad_(True)
cond = torch.randn(M, 66, device="cuda")
gs = [torch.randn(M, n, device="cuda") for n in (10, 30, 70, 30)]
params = [p for net in gens for p in net.parameters()] + list(deform.parameters())


def run(chain):
    feat.grad = None
    for p in params:
        p.grad = None
    if chain:
        outs = mlp.generate_all(gens, lin, feat, cond)
    else:
        films = [g.film_nets(cond) for g in gens]
        outs = [g(feat, cond, film=f) for g, f in zip(gens, films)] + [deform(torch.cat([feat, cond], 1))]
    torch.autograd.backward(outs, gs)


for chain in (False, True):
    for _ in range(5):
        run(chain)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run(chain)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20 * 1e3
    _lib.profile_enable(True)
    for _ in range(5):
        run(chain)
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    _lib.profile_enable(False)
    tot = sum(ms for _, ms in prof.values()) / 5
    print(f"{'chain' if chain else 'layer'}: M={M} wall {wall:.3f} ms/pass, library kernels {tot:.3f} ms/pass, launches {sum(n for n, _ in prof.values()) / 5:.0f}")
    for k, (n, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        print(f"    {k:28s} {n / 5:5.1f} x {1e3 * ms / n:8.1f} us")
