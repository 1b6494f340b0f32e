import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsvc_amd import mlp, _lib
torch.manual_seed(0)
M = 4096
nets = [torch.nn.Sequential(torch.nn.Linear(192, 50), torch.nn.GELU(), torch.nn.Linear(50, 1)).cuda() for _ in range(3)]
x = torch.randn(M, 192, device="cuda")
params = [p.detach() for n in nets for p in (n[0].weight, n[0].bias, n[2].weight, n[2].bias)]
class Ctx:
    def save_for_backward(self, *a): self.saved = a
    def set_materialize_grads(self, v): pass
ctx = Ctx()
q = mlp._QuantStepNets.forward(ctx, x, *params)
xs, buf, *_ = ctx.saved
Mh, H, per = ctx.geom
for i in range(3):
    z = buf[i * per:i * per + M * H].view(M, H); a = buf[(3 + i) * per:(3 + i) * per + M * H].view(M, H)
    zr = torch.nn.functional.linear(x, nets[i][0].weight, nets[i][0].bias); ar = torch.nn.functional.gelu(zr)
    qr = nets[i][2](ar)
    print(i, "z err", float((z - zr).abs().max()), "a err", float((a - ar).abs().max()), "q err", float((q[i] - qr).abs().max()), float(q[i][0]), float(qr[0]))
z = buf[0:M * H].view(M, H)
zr = torch.nn.functional.linear(x, nets[0][0].weight, nets[0][0].bias)
print("z[0,:6]", z[0, :6].tolist()); print("zr[0,:6]", zr[0, :6].tolist()); print("b1[:6]", nets[0][0].bias[:6].tolist())
print("z[17,44:50]", z[17, 44:50].tolist()); print("zr[17,44:50]", zr[17, 44:50].tolist())
print("frac of rows close", float(((z - zr).abs().max(dim=1).values < 1e-4).float().mean()), "cols close", ((z - zr).abs().max(dim=0).values < 1e-4).tolist())
print("x[0,:4]", x[0, :4].tolist(), "W1[0,:4]", nets[0][0].weight[0, :4].tolist(), "W1[1,:4]", nets[0][0].weight[1, :4].tolist())
print("z[0,:4] (x frag)", z[0, :4].tolist(), "z[0,16:20] (w frag)", z[0, 16:20].tolist(), "z[1,16:20]", z[1, 16:20].tolist())
print("z[0,32:36] (RB rb M stride)", z[0, 32:36].tolist(), "z[0,48:50] direct x", z[0, 48:50].tolist(), "z[20,32:36]", z[20, 32:36].tolist())
