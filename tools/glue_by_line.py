"""Diagnostic: device time of the PyTorch (non-gsvc) operators of one fitting step, grouped by the gsvc_amd source line that
dispatched them (forward) or by autograd node (backward)."""
import os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
from gsvc_amd.train import Trainer
dev = torch.device("cuda")
mp_, opt, pipe = cfg_20240919()
cube = SyntheticFrameCube(1080, 1920, 64, device=dev).materialize()
mp_.threshold = 8.0 / cube.scale
opt.full_precision_training_total = opt.quantized_training_total = 0
opt.entropy_constrained_train_total = 10 ** 9
opt.start_stat, opt.update_until, opt.pause_densification = 0, 10 ** 9, 0
pc = GaussianModel(mp_, 50, 10, 0.001, 3, 16, 4, False, n_features_per_level=8, log2_hashmap_size=13, log2_hashmap_size_2D=15, device=dev)
rng = np.random.default_rng(0)
lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
pc.create_from_points(rng.uniform(lim, -lim, (245000, 3)), 1.0)
pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
pc.training_setup(opt)
tr = Trainer(pc, cube, opt, pipe, mp_)
for i in range(30):
    tr.step(i + 1)
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    for i in range(N):
        tr.step(40 + i)
    torch.cuda.synchronize()
_src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gsvc_amd", "train.py")).read().splitlines()
BACKWARD_LINES = [i + 1 for i, l in enumerate(_src) if "loss.backward()" in l or "return self._step_body(" in l]
by = defaultdict(lambda: [0.0, 0])
tot = 0.0
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.self_device_time_total <= 0:
        continue
    fr = [f for f in ev.stack if "gsvc_amd" in f and "torch/" not in f]
    where = fr[0].split("gsvc_amd/")[-1].split(":")[0].strip() if fr else "(autograd engine)"
    k = (where, ev.name)
    by[k][0] += ev.self_device_time_total
    by[k][1] += 1
    tot += ev.self_device_time_total
print(f"aten device time per step: {tot / N / 1e3:.2f} ms")
line = defaultdict(float)
for (w, n), (t, c) in by.items():
    line[w] += t
for w, t in sorted(line.items(), key=lambda kv: -kv[1])[:45]:
    ops = sorted(((n, tt, c) for (ww, n), (tt, c) in by.items() if ww == w), key=lambda x: -x[1])[:5]
    print(f"{t / N:8.1f} us  {w:34s} " + ", ".join(f"{n[6:]} {tt / N:.0f}us x{c // N}" for n, tt, c in ops))
# the backward's own launches (gradient accumulation at fan-outs, reductions inside autograd nodes): by operator and shape
bw = defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.self_device_time_total <= 0:
        continue
    fr = [f for f in ev.stack if "gsvc_amd" in f and "torch/" not in f]
    if fr and not any(f"train.py({ln})" in fr[0] for ln in BACKWARD_LINES) and "backward" not in fr[0]:
        continue
    shp = str([list(x) for x in (ev.input_shapes or []) if x])[:70]
    k = (ev.name, shp, (fr[0].split("gsvc_amd/")[-1].split(":")[0].strip() if fr else ""))
    bw[k][0] += ev.self_device_time_total
    bw[k][1] += 1
# by shape for the three most expensive source lines
top = [w for w, _ in sorted(line.items(), key=lambda kv: -kv[1])[:3]]
ts = defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.self_device_time_total <= 0:
        continue
    fr = [f for f in ev.stack if "gsvc_amd" in f and "torch/" not in f]
    where = fr[0].split("gsvc_amd/")[-1].split(":")[0].strip() if fr else "(autograd engine)"
    if where in top and where != "(autograd engine)":
        shp = str([list(x) for x in (ev.input_shapes or []) if x])[:70]
        ts[(where, ev.name, shp)][0] += ev.self_device_time_total
        ts[(where, ev.name, shp)][1] += 1
print("most expensive lines, by operator and shape:")
for (w, n, shp), (t, c) in sorted(ts.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{t / N:8.1f} us x{c / N:4.1f}  {w:22s} {n[6:]:14s} {shp}")
print("backward-side aten launches by shape:")
for (n, shp, w), (t, c) in sorted(bw.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{t / N:8.1f} us x{c / N:4.1f}  {n[6:]:18s} {shp:72s} {w}")
# zero fills / copies by shape and source line
fl = defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if ev.name not in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::cat") or ev.self_device_time_total <= 0:
        continue
    fr = [f for f in ev.stack if "gsvc_amd" in f and "torch/" not in f]
    w = fr[0].split("gsvc_amd/")[-1].split(":")[0].strip() if fr else "(autograd engine)"
    shp = str([list(x) for x in (ev.input_shapes or []) if x])[:60]
    fl[(ev.name, shp, w)][0] += ev.self_device_time_total
    fl[(ev.name, shp, w)][1] += 1
print("fills / copies / cats by shape:")
for (n, shp, w), (t, c) in sorted(fl.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{t / N:8.1f} us x{c / N:4.1f}  {n[6:]:8s} {shp:62s} {w}")
# backward-side launches by the autograd node that issued them (gradient accumulation at a fan-out shows up under the node
# whose output is accumulated: "evaluate_function: XBackward" with an aten::add)
nd = defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.self_device_time_total <= 0:
        continue
    p, node = ev.cpu_parent, None
    while p is not None:
        if "evaluate_function" in p.name or p.name.endswith("Backward") or "Backward" in p.name:
            node = p.name
        p = p.cpu_parent
    if node is None:
        continue
    shp = str([list(x) for x in (ev.input_shapes or []) if x])[:48]
    nd[(node.replace("autograd::engine::evaluate_function: ", "")[:44], ev.name[6:], shp)][0] += ev.self_device_time_total
    nd[(node.replace("autograd::engine::evaluate_function: ", "")[:44], ev.name[6:], shp)][1] += 1
print("backward launches by autograd node:")
for (node, n, shp), (t, c) in sorted(nd.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"{t / N:8.1f} us x{c / N:4.1f}  {node:44s} {n:16s} {shp}")
