"""From a rocprofv3 kernel trace: do the small-work kernels (k_linear_ws, k_rate_sample...) run concurrently with the compositing
kernels?  usage: python tools/scratch/overlap_trace.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def name(r): return r.get("Kernel_Name") or r.get("Name")
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name(r), r.get("Queue_Id"), r.get("Stream_Id")) for r in rows]
ks.sort()
ks = ks[len(ks) // 2:]                     # the second half of the run (steady steps)
big = [(s, e) for s, e, n, q, st in ks if "k_blend" in n]
small = [(s, e, n, q, st) for s, e, n, q, st in ks if "k_linear_ws" in n or "k_rate_sample" in n or "k_regs" in n or "k_optical" in n]
def overlapped(s, e):
    return any(bs < e and s < be for bs, be in big)
tot = sum(e - s for s, e, *_ in small)
ov = sum(e - s for s, e, *_ in small if overlapped(s, e))
print(f"small kernels {len(small)}, total {tot / 1e3:.0f} us, of which concurrent with a compositing kernel {ov / 1e3:.0f} us ({100.0 * ov / max(tot, 1):.0f} %)")
qs = collections.Counter((n.split('(')[0][-28:], q, st) for s, e, n, q, st in small)
for k, v in sorted(qs.items(), key=lambda kv: -kv[1])[:8]: print("  ", k, v)
qb = collections.Counter((q, st) for s, e, n, q, st in ks if "k_blend" in n)
print("compositing kernels by (queue, stream):", dict(qb))
qm = collections.Counter((q, st) for s, e, n, q, st in ks if "k_trunk" in n)
print("trunk kernels by (queue, stream):", dict(qm))
