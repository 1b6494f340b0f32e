"""Diagnostic: launches per fitting step by kernel (rocprofv3 kernel_trace.csv), last full step."""
import csv, re, sys
from collections import Counter
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
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
