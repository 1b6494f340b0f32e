"""Means of the SQ counters of one kernel's dispatches, variant A against variant B, from a rocprofv3 counter_collection.csv: the probe
scripts of this directory end with N launches of variant A followed by N of variant B (earlier dispatches of the same kernel — the
fit that builds the scene — are ignored).  usage: pmc_split.py <csv> <kernel substring> <N>"""
import csv, sys
from collections import defaultdict
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
N = int(sys.argv[3])
by = defaultdict(list)
for r in rows:
    by[int(r["Dispatch_Id"])].append(r)
ids = sorted(by)[-2 * N:]
for tag, part in (("variant A", ids[:N]), ("variant B", ids[N:])):
    acc = defaultdict(float)
    for i in part:
        for r in by[i]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(tag, len(part), "dispatches, per dispatch:", "  ".join(f"{k} {v / max(len(part), 1):.4g}" for k, v in sorted(acc.items())))
