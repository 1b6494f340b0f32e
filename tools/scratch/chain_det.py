import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gsvc_amd import mlp
from gsvc_amd.model import GeluSequential, GeneratorNet, Linear
M = 4096
torch.manual_seed(M)
gens = [GeneratorNet(50, 10, 100, 66, out_act=torch.nn.Tanh()).cuda(), GeneratorNet(50, 30, 100, 66, out_act=torch.nn.Sigmoid()).cuda(),
        GeneratorNet(50, 70, 100, 66).cuda()]
deform = GeluSequential(Linear(116, 100), torch.nn.GELU(), Linear(100, 100), torch.nn.GELU(), Linear(100, 100), torch.nn.GELU(),
                        Linear(100, 100), torch.nn.GELU(), Linear(100, 30)).cuda()
lin = list(deform)[0::2]
feat = (torch.randn(M, 50, device="cuda") * 2).requires_grad_(True)
cond = torch.randn(M, 66, device="cuda")
gs = [torch.randn(M, n, device="cuda") for n in (10, 30, 70, 30)]
names = ["feat"] + [f"gen{g}.{n}" for g in range(3) for n, _ in gens[g].named_parameters()] + [f"deform.{n}" for n, _ in deform.named_parameters()]
params = [p for net in gens for p in net.parameters()] + list(deform.parameters())
def run(chain):
    feat.grad = None
    for p in params: p.grad = None
    if chain:
        outs = mlp.generate_all(gens, lin, feat, cond)
    else:
        outs = [g(feat, cond) for g in gens] + [deform(torch.cat([feat, cond], 1))]
    sum((o * g).sum() for o, g in zip(outs, gs)).backward()
    return [feat.grad.clone()] + [p.grad.clone() for p in params]
for chain in (True, False):
    a, b, c = run(chain), run(chain), run(chain)
    for n, x, y, z in zip(names, a, b, c):
        if not torch.equal(x, y) or not torch.equal(x, z):
            print("chain" if chain else "layer", n, tuple(x.shape), (x - y).abs().max().item(), (x - z).abs().max().item())
print("done")
