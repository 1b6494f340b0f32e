"""Microbenchmark of the hash-grid kernels at the train-step shapes."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsvc_amd.encodings import GridEncoder
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 181585
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for D, res, log2 in ((3, (18, 24, 33, 44, 59, 80, 108, 148, 201, 275, 376, 514), 13), (2, (130, 258, 514, 1026), 15)):
    enc = GridEncoder(num_dim=D, n_features=8, resolutions_list=res, log2_hashmap_size=log2, ste_binary=True).to(dev)
    x = torch.rand(N, D, device=dev, requires_grad=True)
    y = enc(x)
    g = torch.randn_like(y)
    t_f = timeit(lambda: enc(x))
    def fb():
        x.grad = None
        enc.zero_grad()
        enc(x).backward(g)
    t_fb = timeit(fb)
    print(f"D={D} L={len(res)} N={N}: fwd {t_f:.1f} us, fwd+bwd {t_fb:.1f} us")
