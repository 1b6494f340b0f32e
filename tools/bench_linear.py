"""Microbenchmark: library F.linear vs k_linear_fwd vs k_linear_ws on the MLP shapes of the train step."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsvc_amd import _lib
import torch.nn.functional as F
L = _lib.lib()
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 196608
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
st = _lib.current_stream(dev)
for (K, N) in [(50, 100), (100, 100), (66, 66), (66, 100), (100, 10), (100, 70), (116, 100), (192, 150), (150, 100), (192, 50), (100, 30), (10, 100), (70, 100), (150, 192), (192, 192)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev); y2 = torch.empty(M, N, device=dev)
    ref = F.linear(x, w, b)
    _lib.check(L.gsvc_linear_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, K, N, 0, 0, st), "ws")
    wt = w.t().contiguous()
    _lib.check(L.gsvc_linear_forward(_lib.ptr(x), _lib.ptr(wt), _lib.ptr(b), _lib.ptr(y2), M, K, N, 1, 1, st), "ws")
    err = ((y - ref).abs().max() / ref.abs().max()).item()
    err2 = ((y2 - ref.clamp_min(0)).abs().max() / ref.abs().max()).item()
    t_lib = timeit(lambda: F.linear(x, w, b))
    t_old = float('nan')
    t_ws = timeit(lambda: L.gsvc_linear_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, K, N, 0, 0, st))
    gb = M * (K + N) * 4 / 1e9
    print(f"K={K:4d} N={N:4d} lib {t_lib:7.1f}  old {t_old:7.1f}  ws {t_ws:7.1f} us  ({gb / t_ws * 1e6 / 1e3:5.2f} TB/s, {2 * M * K * N / t_ws / 1e6:5.1f} TF)  err {err:.1e} {err2:.1e}")
