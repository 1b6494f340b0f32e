"""Diagnostic: launches per fitting step by kernel (rocprofv3 kernel_trace.csv), last full step."""
import csv, re, sys
from collections import Counter
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# one step = from one positional-embedding launch (one per step, near its start) to the next
marks = [i for i, e in enumerate(ev) if 'k_embed_pe' in e[2]]
# fitting steps only (the bench renders frames afterwards): intervals that contain a weight-gradient launch
steps = [(m0, m1) for m0, m1 in zip(marks[:-1], marks[1:]) if any('k_linear_wgrad' in e[2] for e in ev[m0:m1])]
if len(steps) >= 2:
    a, b = steps[-2][0] - 1, steps[-2][1] - 1
else:
    adam = [i for i, e in enumerate(ev) if 'k_adam' in e[2]]
    ends = [i for j, i in enumerate(adam) if j + 1 == len(adam) or ev[adam[j + 1]][0] - ev[i][1] > 5_000_000]
    a, b = ends[-3], ends[-2]
seg = ev[a + 1:b + 1]
def short(n):
    m = re.search(r'(gsvc::k_\w+)', n)
    if m: return m.group(1)
    m = re.search(r'at::native::(?:\(anonymous namespace\)::)?(\w+)<[^>]*?at::native::(?:\(anonymous namespace\)::)?(\w+)', n)
    if m: return m.group(1)[:24] + ':' + m.group(2)[:30]
    return re.sub(r'void ', '', n)[:50]
cnt, tim = Counter(), Counter()
for s, e, n in seg:
    k = short(n); cnt[k] += 1; tim[k] += e - s
print("kernels", len(seg), "busy ms", sum(tim.values()) / 1e6)
for k, c in cnt.most_common(90):
    print(f"{c:5d} {tim[k] / 1e3:9.1f} us  {k}")
# where the GPU waits for the host: idle time between consecutive kernels of that step, largest gaps first
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(seg[:-1], seg[1:]):
    if s1 > e0:
        gaps.append((s1 - e0, short(n0), short(n1), (e0 - seg[0][0]) / 1e6))
tot = sum(g[0] for g in gaps)
print(f"step span {(seg[-1][1] - seg[0][0]) / 1e6:.2f} ms, idle between kernels {tot / 1e6:.2f} ms in {len(gaps)} gaps; gaps > 5 us: "
      f"{sum(g[0] for g in gaps if g[0] > 5000) / 1e6:.2f} ms")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"  {g[0] / 1e3:7.1f} us at +{g[3]:6.2f} ms  after {g[1][:40]:40s} before {g[2][:40]}")
# idle per millisecond of the step
import collections
per = collections.Counter()
for g in gaps:
    per[int(g[3])] += g[0]
print("idle us per ms of the step:", " ".join(f"{int(per[k] / 1e3)}" for k in range(int((seg[-1][1] - seg[0][0]) / 1e6) + 1)))
# optional second argument: a regular expression — every launch of the matching kernels in that step, in order, with its duration
if len(sys.argv) > 2:
    print(f"launches matching '{sys.argv[2]}':")
    for s, e, n in seg:
        if re.search(sys.argv[2], n):
            print(f"  +{(s - seg[0][0]) / 1e6:6.2f} ms  {(e - s) / 1e3:8.1f} us  {short(n)}")
