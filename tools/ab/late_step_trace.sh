#!/bin/bash
# Kernel time per step LATE in a fit without the library's per-kernel events (which serialise the step's streams): a rocprofv3 kernel trace of
# a 6 000-step run of tools/ab/late_stage_profile.py's fit (LATE_NO_EVENTS=1: plain steps), of which only the last 0.2 s are summed per kernel.
#   bash tools/ab/late_step_trace.sh   ->  gpurun_out/late_step_trace.txt
export TMPDIR=/tmp LATE_NO_EVENTS=1
OUT=/tmp/late_trace; rm -rf $OUT
REPO=$PWD
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/tools/ab/late_stage_profile.py 10000 6100 > $REPO/gpurun_out/late_step_trace.log 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $REPO/gpurun_out/late_step_trace.txt <<'PY'
import csv, sys, collections
rows = []
with open(sys.argv[1]) as fh:
    r = csv.DictReader(fh)
    for d in r:
        rows.append((int(d["Start_Timestamp"]), int(d["End_Timestamp"]), d["Kernel_Name"]))
end = max(e for _, e, _ in rows)
win = [x for x in rows if x[0] >= end - 200_000_000]
steps = sum(1 for x in win if "k_training_statis" in x[2]) or 1
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in win:
    k = k.split("(")[0].split("<")[0].replace("gsvc::", "").replace("(anonymous namespace)::", "")
    agg[k][0] += 1; agg[k][1] += e - s
span = (max(e for _, e, _ in win) - min(s for s, _, _ in win)) / 1e6
print(f"last 0.2 s of the trace: {len(win)} dispatches, {steps} steps, {span / steps:.3f} ms per step (wall), kernel time summed {sum(v[1] for v in agg.values()) / 1e6 / steps:.3f} ms per step")
for k, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
    print(f"{ns / 1e3 / steps:9.1f} us/step  x{n / steps:5.1f}  {k}")
PY
rm -rf $OUT
cat $REPO/gpurun_out/late_step_trace.txt
