"""Timeline of one fitting step from a rocprofv3 kernel trace: every kernel with its start (ms from the step's first kernel), duration,
queue (= HIP stream) and how many other kernels ran beside it; the step's span, the time at least one kernel ran (union), the idle
time, and the time per kernel family during which it was the ONLY family running (its exposed share).
usage: python tools/step_timeline.py <kernel_trace.csv> [step_from_the_end=2] [--list]"""
import csv, re, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 2
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in rows)


def short(n):
    m = re.search(r'gsvc::(k_\w+)', n)
    if m:
        return m.group(1)
    m = re.search(r'at::native::(?:\(anonymous namespace\)::)?(\w+)', n)
    if m:
        return 'torch:' + m.group(1)[:28]
    return re.sub(r'void ', '', n)[:36]


# a step ends with its FULL Adam launch (every parameter: the longest k_adam launches; the guarded early updates of _scaling / _mask
# from inside the backward are several times shorter).  Round 6: the old rule (a gap of 1.5 ms to the next k_adam) cut steps in two.
adam = [i for i, e in enumerate(ev) if 'k_adam' in e[2]]
longest = max(ev[i][1] - ev[i][0] for i in adam)
ends = [i for i in adam if ev[i][1] - ev[i][0] > 0.5 * longest]
a, b = ends[-back - 1], ends[-back]
seg = ev[a + 1:b + 1]
t0 = seg[0][0]
span = seg[-1][1] - t0
# sweep: union busy, exclusive time per family
pts = sorted([(s, 1, short(n)) for s, e, n, q in seg] + [(e, -1, short(n)) for s, e, n, q in seg])
active = defaultdict(int)
last = t0
union = 0
alone = defaultdict(int)
for t, d, name in pts:
    fams = [k for k, v in active.items() if v > 0]
    if fams:
        union += t - last
        if len(fams) == 1:
            alone[fams[0]] += t - last
    last = t
    active[name] += d
queues = sorted({q for *_, q in seg})
print(f"step: {len(seg)} kernels on {len(queues)} queues, span {span / 1e6:.3f} ms, some kernel running {union / 1e6:.3f} ms, idle {(span - union) / 1e6:.3f} ms, "
      f"sum of durations {sum(e - s for s, e, *_ in seg) / 1e6:.3f} ms")
tot = defaultdict(lambda: [0, 0])
for s, e, n, q in seg:
    tot[short(n)][0] += e - s
    tot[short(n)][1] += 1
print("family                         launches   sum_us  alone_us  overlapped_us")
for k, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"{k:30s} {c:8d} {d / 1e3:8.1f} {alone[k] / 1e3:9.1f} {(d - alone[k]) / 1e3:10.1f}")
print(f"alone_us summed over the families = the part of the step's {union / 1e6:.3f} busy ms during which ONE family had the chip: "
      f"{sum(alone.values()) / 1e6:.3f} ms; the rest ({(union - sum(alone.values())) / 1e6:.3f} ms) is shared by two or more families")
if '--list' in sys.argv:
    for s, e, n, q in seg:
        print(f"{(s - t0) / 1e6:8.3f} ms {(e - s) / 1e3:8.1f} us  q{queues.index(q)}  {short(n)}")
