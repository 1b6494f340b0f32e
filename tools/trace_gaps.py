"""Diagnostic: GPU busy/idle per fitting step from a rocprofv3 kernel trace (kernel_trace.csv)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
adam = [i for i, e in enumerate(ev) if 'FusedAdam' in e[2] or 'k_adam' in e[2]]
ends = [i for j, i in enumerate(adam) if j + 1 == len(adam) or ev[adam[j + 1]][0] - ev[i][1] > 5_000_000]
def short(n):
    m = re.search(r'(gsvc::k_\w+)', n)
    if m: return m.group(1)
    m = re.search(r'at::native::(?:\(anonymous namespace\)::)?(\w+)<[^>]*?at::native::(?:\(anonymous namespace\)::)?(\w+)', n)
    if m: return m.group(1)[:20] + ':' + m.group(2)[:28]
    return re.sub(r'void ', '', n)[:40]
for a, b in zip(ends[-4:-1], ends[-3:]):
    seg = ev[a + 1:b + 1]
    t0 = seg[0][0]
    span = seg[-1][1] - t0
    busy = sum(e[1] - e[0] for e in seg)
    print(f"step: span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, kernels {len(seg)}")
    # idle per 2-ms window
    W = 2_000_000
    nb = span // W + 1
    idle = [0] * nb
    for i in range(len(seg) - 1):
        g = seg[i + 1][0] - seg[i][1]
        if g > 0: idle[(seg[i][1] - t0) // W] += g
    print("  idle ms per 2-ms window:", [round(x / 1e6, 2) for x in idle])
    marks = {}
    for s, e, n in seg:
        k = short(n)
        if k.startswith('gsvc::') and k not in marks: marks[k] = round((s - t0) / 1e6, 2)
    print("  first occurrence (ms):", marks)
    gaps = sorted(((seg[i + 1][0] - seg[i][1], round((seg[i][1] - t0) / 1e6, 2), short(seg[i][2]) + ' -> ' + short(seg[i + 1][2])) for i in range(len(seg) - 1)), reverse=True)[:10]
    print("  top gaps (us, at ms, between):", [(g // 1000, t, n) for g, t, n in gaps])
