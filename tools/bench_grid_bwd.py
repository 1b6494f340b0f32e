"""Microbenchmark of the hash-grid table backward (csrc/grid.hip k_grid_bwd_lds) at the fitting step's shapes:
3-D grid 12 levels x 2^13 rows x 8 features, 2-D grids 4 levels x 2^15 rows x 8 features."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gsvc_amd.encodings import GridEncoder
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 115000
res3 = (16, 23, 32, 46, 64, 92, 128, 184, 256, 368, 514, 736)[:12]
res2 = (130, 258, 514, 1026)


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for D, res, log2 in ((3, res3, 13), (2, res2, 15)):
    enc = GridEncoder(num_dim=D, n_features=8, resolutions_list=res, log2_hashmap_size=log2).to(dev)
    x = torch.rand(N, D, device=dev)
    out = enc(x)
    g = torch.randn_like(out)
    def run():
        enc.params.grad = None
        out.backward(g, retain_graph=True)
    print(f"D={D} N={N} levels={len(res)} fwd+bwd call {timeit(run):7.1f} us", flush=True)
