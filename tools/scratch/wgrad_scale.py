import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsvc_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); st = _lib.current_stream(dev)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (K, N) in [(100, 100), (66, 66), (100, 10)]:
    row = []
    for M in (50_000, 100_000, 200_000, 400_000, 800_000):
        x = torch.randn(M, K, device=dev); g = torch.randn(M, N, device=dev)
        wsn = int(L.gsvc_linear_wgrad_workspace(N, K)); ws = torch.empty(wsn, device=dev)
        slots = __import__("ctypes").c_int32(0)
        def mine():
            L.gsvc_linear_wgrad_partial(_lib.ptr(g), _lib.ptr(x), 1, M, N, K, _lib.ptr(ws), wsn, __import__("ctypes").byref(slots), st)
        t = timeit(mine)
        row.append(f"M={M} {t:6.1f}us {M*(K+N)*4/t/1e6:4.2f}TB/s")
    print(f"K={K} N={N}: " + " | ".join(row))
