"""Per-kernel summary of a rocprofv3 counter_collection extract (rows of gsvc kernels): launches, mean / min / max of every
counter — the committed form of the large per-dispatch CSVs (profiles/rNN/*_pmc_*_summary.csv)."""
import csv, re, sys
from collections import defaultdict
acc = defaultdict(list)
with open(sys.argv[1], newline="") as f:
    for row in csv.DictReader(f):
        m = re.search(r"(gsvc::(?:\(anonymous namespace\)::)?k_\w+(<[^>(]*>)?)", row["Kernel_Name"])
        if m:
            acc[(m.group(1), row["Counter_Name"])].append(float(row["Counter_Value"]))
w = csv.writer(sys.stdout)
w.writerow(["Kernel", "Counter", "Launches", "Mean", "Min", "Max"])
for (k, c), v in sorted(acc.items()):
    w.writerow([k, c, len(v), f"{sum(v) / len(v):.1f}", f"{min(v):.1f}", f"{max(v):.1f}"])
