"""Microbenchmark A/B of csrc/linear.hip: plain forward, every epilogue program and the weight gradient on the MLP shapes of
the fitting step.  GSVC_LIB_PATH selects the library (an older build for comparison)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gsvc_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
has_ex = hasattr(L, "gsvc_linear_forward_ex")
st = _lib.current_stream(dev)


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


P = _lib.ptr
for (K, N) in [(50, 100), (100, 100), (66, 66), (66, 100), (100, 10), (100, 70), (100, 30), (116, 100), (192, 150), (150, 100), (192, 50), (10, 100), (70, 100), (30, 100), (100, 66)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
    y, y2, y3, a1, a2 = (torch.randn(M, N, device=dev) for _ in range(5))
    t_plain = timeit(lambda: L.gsvc_linear_forward(P(x), P(w), P(b), P(y), M, K, N, 0, 0, st))
    line = f"K={K:4d} N={N:4d} plain {t_plain:6.1f}"
    if has_ex:
        for mode, name in [(2, "gelu2"), (5, "xgelu'"), (6, "xrelu'"), (7, "film"), (8, "filmg"), (3, "tanh")]:
            t = timeit(lambda: L.gsvc_linear_forward_ex(P(x), P(w), P(b), P(y), M, K, N, 0, mode, P(a1), P(a2), P(y2), P(y3), st))
            line += f"  {name} {t:6.1f}"
    g = torch.randn(M, N, device=dev)
    wsf = int(L.gsvc_linear_wgrad_workspace(N, K))
    ws = torch.empty(wsf, device=dev); dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    t_wg = timeit(lambda: L.gsvc_linear_wgrad(P(g), P(x), P(dw), P(db), M, N, K, P(ws), wsf, st))
    ref = g.t() @ x
    err = ((dw - ref).abs().max() / ref.abs().max()).item()
    line += f"  | wgrad {t_wg:6.1f} (err {err:.0e})"
    print(line, flush=True)
