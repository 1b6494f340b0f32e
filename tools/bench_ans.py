"""Micro-benchmark of the entropy decoder kernel (diagnostic): python tools/bench_ans.py [n] [sigma]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gsvc_amd import _lib, codec
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_500_000
sig = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
mu = torch.randn(n, device=dev, generator=g) * 3.0
sigma = torch.full((n,), sig, device=dev)
sym = torch.round(mu + sigma * torch.randn(n, device=dev, generator=g)).to(torch.int32)
smin, smax = int(sym.min()), int(sym.max())
stream = codec.ans_encode(sym, mu, sigma, smin, smax)
ps = codec.prepare_streams([stream], dev)[0]
checks = codec.DeferredChecks()
for _ in range(2):
    out = codec.ans_decode(ps, mu, sigma, defer=checks)
torch.cuda.synchronize()
_lib.profile_enable(True)
for _ in range(5):
    out = codec.ans_decode(ps, mu, sigma, defer=checks)
torch.cuda.synchronize()
prof = _lib.profile_collect()
ok = bool(torch.equal(out, sym))
print(f"n={n} sigma={sig} seg_len={ps.seg_len} bits/sym={8 * len(stream) / n:.3f} exact={ok}", {k: round(ms / c, 3) for k, (c, ms) in prof.items()})
