"""Matrix-pipe utilisation of the MLP kernels from a rocprofv3 SQ pass (tools/profile_round.sh: SQ_VALU_MFMA_BUSY_CYCLES,
SQ_INSTS_VALU_MFMA_F32, SQ_BUSY_CYCLES, ... with the dispatches' own timestamps).  Per kernel: launches, mean duration, fp32 MFMA
wave-instructions per launch, and two estimates of how busy the 1 024 matrix pipes were:
  issue   = MFMA instructions x 32 cycles (v_mfma_f32_16x16x4_f32: 32 cycles per SIMD, MI355X_MICROARCH.md) / (1024 SIMDs x duration x clock)
  counter = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CYCLES)      (busy cycles summed over the 4 SIMDs of a CU against its CU-cycles...
            the counter's unit is established by the `issue` column: the two agree when it is read this way)
clock = SQ_BUSY_CYCLES / 32 shader-engine instances / duration.   usage: python tools/mfma_util.py <extract.csv>"""
import csv, re, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
with open(sys.argv[1], newline="") as f:
    for row in csv.DictReader(f):
        m = re.search(r"gsvc::(?:\(anonymous namespace\)::)?(k_\w+)", row["Kernel_Name"])
        if not m:
            continue
        k = m.group(1)
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        acc[k]["dur_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "launches", "mean_us", "mfma_f32_insts_per_launch", "clock_MHz", "mfma_issue_share", "SQ_VALU_MFMA_BUSY_CYCLES_per_inst",
            "mfma_busy_over_4xSQ_BUSY", "valu_insts_per_mfma", "wait_inst_any_frac", "wait_any_frac", "lds_bank_conflict_per_wave_cycle"])
mean = lambda v: sum(v) / len(v) if v else 0.0  # noqa: E731
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1]["dur_ns"])):
    n = len(c["SQ_BUSY_CYCLES"]) or 1
    dur = mean(c["dur_ns"])
    mf, busy, mb = mean(c["SQ_INSTS_VALU_MFMA_F32"]), mean(c["SQ_BUSY_CYCLES"]), mean(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    mhz = busy / 32.0 / (dur * 1e-3) if dur else 0.0
    issue = mf * 32.0 / (1024.0 * dur * 1e-3 * mhz) if dur and mhz else 0.0
    wc = mean(c["SQ_WAVE_CYCLES"]) or 1.0
    w.writerow([k, n, f"{dur / 1e3:.1f}", f"{mf:.0f}", f"{mhz:.0f}", f"{issue:.3f}", f"{mb / mf:.2f}" if mf else "", f"{mb / (4 * busy):.3f}" if busy else "",
                f"{mean(c['SQ_INSTS_VALU']) / mf:.2f}" if mf else "", f"{mean(c['SQ_WAIT_INST_ANY']) / wc:.3f}", f"{mean(c['SQ_WAIT_ANY']) / wc:.3f}",
                f"{mean(c['SQ_LDS_BANK_CONFLICT']) / wc:.4f}"])
