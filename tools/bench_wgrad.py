"""Microbenchmark: dW = G^T X through gsvc_linear_wgrad vs the library (plain and M-split batched)."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsvc_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 181585
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
st = _lib.current_stream(dev)
for (K, N) in [(50, 100), (100, 100), (66, 66), (66, 100), (100, 10), (100, 70), (116, 100), (192, 150), (150, 100), (192, 50), (100, 30), (192, 192), (51, 37), (100, 1)]:
    x = torch.randn(M, K, device=dev); g = torch.randn(M, N, device=dev)
    ref = (g.double().t() @ x.double()).float(); refb = g.double().sum(0).float()
    gw = torch.zeros(N, K, device=dev); gb = torch.zeros(N, device=dev)
    wsn = int(L.gsvc_linear_wgrad_workspace(N, K)); ws = torch.empty(wsn, device=dev)
    _lib.check(L.gsvc_linear_wgrad(_lib.ptr(g), _lib.ptr(x), _lib.ptr(gw), _lib.ptr(gb), M, N, K, _lib.ptr(ws), wsn, st), "wgrad")
    err = ((gw - ref).abs().max() / ref.abs().max()).item(); errb = ((gb - refb).abs().max() / refb.abs().max()).item()
    def mine():
        L.gsvc_linear_wgrad(_lib.ptr(g), _lib.ptr(x), _lib.ptr(gw), _lib.ptr(gb), M, N, K, _lib.ptr(ws), wsn, st)
    S = M // 4096
    def split():
        main = S * 4096
        r = torch.bmm(g[:main].view(S, 4096, -1).transpose(1, 2), x[:main].view(S, 4096, -1)).sum(dim=0) + g[main:].t() @ x[main:]
        b = g.sum(0)
    print(f"K={K:4d} N={N:4d} mine {timeit(mine):7.1f} us   bmm-split+sum {timeit(split):7.1f} us   ({M * (K + N) * 4 / 1e3 / timeit(mine) / 1e3:5.2f} TB/s) err {err:.1e} {errb:.1e}")
