"""Turn rocprofv3 PMC passes into profiles/pmc_latest.json (the `roofline.traffic` source of bench.py).

Usage: python tools/pmc_extract.py <workload> <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> [tag [SQ csv]]
       python tools/pmc_extract.py --stats <kernel_stats.csv> [rows]     (shorten the kernel names of a --stats summary)

Per MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes, are
reported in KiB per dispatch, and FETCH_SIZE under-counts by 2x on gfx950 (64-byte requests counted as 32) -> doubled.
HBM bytes per launch = (2 * FETCH_KiB + WRITE_KiB) * 1024, averaged over the launches of each gsvc kernel.
"""
import csv, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


ALIAS = {"k_blend_bwd_tile": "k_blend_bwd", "k_scatter_lds": "k_scatter"}      # kernel function name -> the name bench.py's HIP-event profiler uses


def short(name):
    m = re.search(r"gsvc::(k_\w+)(<[^>]*>)?", name)
    if not m:
        return None
    k = m.group(1)
    if k == "k_blend" and m.group(2) == "<true>":
        return "k_blend_pair"                    # two-view composite: different traffic from the single view
    return ALIAS.get(k, k)


def averages(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            if k and row["Counter_Name"] == counter:
                acc[k][0] += float(row["Counter_Value"])
                acc[k][1] += 1
    return {k: s / n for k, (s, n) in acc.items() if n}


def shorten_stats(path, rows):
    """rocprofv3 --stats summary with the template / argument lists cut from the kernel names (every column kept)."""
    w = csv.writer(sys.stdout)
    with open(path, newline="") as f:
        for i, row in enumerate(csv.reader(f)):
            if i > rows:
                break
            n = row[0]
            m = re.search(r"(gsvc::(?:\(anonymous namespace\)::)?k_\w+(<[^>(]*>)?)", n)
            row[0] = m.group(1).replace("(anonymous namespace)::", "") if m else re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:110]
            w.writerow(row)


def valu_summary(sq_csv):
    """Per kernel: vector wave-instructions per launch and the share of the chip's vector-issue slots they fill.  One wave64
    vector instruction occupies its SIMD's issue port for 4 cycles (MI355X_MICROARCH.md, vector-instruction ISSUE cost; 8 for
    transcendentals, so this is a lower bound): issue share = 4 * SQ_INSTS_VALU / (1024 SIMDs * duration * 2.4 GHz peak clock),
    duration from the same dispatches.  1.0 = the kernel can only get faster by issuing fewer vector instructions."""
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    with open(sq_csv, newline="") as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            if k:
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
                if row["Counter_Name"] == "SQ_INSTS_VALU":
                    dur[k][0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); dur[k][1] += 1
    out = {}
    for k, c in acc.items():
        avg = {n: s / m for n, (s, m) in c.items() if m}
        if "SQ_INSTS_VALU" in avg and dur[k][1]:
            ns = dur[k][0] / dur[k][1]
            out[k] = {"valu_insts_per_launch": int(avg["SQ_INSTS_VALU"]), "launch_us_in_this_pass": round(ns / 1e3, 1),
                      "valu_issue_share_at_2p4GHz": round(4.0 * avg["SQ_INSTS_VALU"] / (1024.0 * ns * 2.4), 3),
                      "wave_cycles_waiting_frac": round(avg.get("SQ_WAIT_ANY", 0.0) / max(avg.get("SQ_WAVE_CYCLES", 1.0), 1.0), 3)}
    return out


def csrc_fingerprint():
    """sha256 (first 16 hex digits) over the kernel sources the library is built from — the same function as bench.py's."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gsvc_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def main():
    if sys.argv[1] == "--stats":
        return shorten_stats(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 60)
    workload, fetch_csv, write_csv = sys.argv[1:4]
    tag = sys.argv[4] if len(sys.argv) > 4 else "unknown"
    sq_csv = sys.argv[5] if len(sys.argv) > 5 else None
    fetch, write = averages(fetch_csv, "FETCH_SIZE"), averages(write_csv, "WRITE_SIZE")
    out_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    data["_note"] = ("HBM bytes per launch from rocprofv3 PMC (separate FETCH_SIZE and WRITE_SIZE passes, values in KiB; "
                     "FETCH doubled per MI355X_MICROARCH.md HBM section; gather-heavy kernels are uncalibrated)")
    data[workload] = {k: int(round((2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024)) for k in sorted(set(fetch) | set(write))}
    if workload == "raster_fwdbwd":              # the forward kernels of the same launches
        data["raster_fwd"] = {k: v for k, v in data[workload].items() if "bwd" not in k}
    data.setdefault("_binary", {})[workload] = tag
    data.setdefault("_csrc_sha16", {})[workload] = csrc_fingerprint()      # bench.py refuses the numbers of another kernel source
    data.setdefault("_source", {})[workload] = (f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) -- python3 bench.py "
                                                f"--workload {workload}; CSVs: {os.path.basename(fetch_csv)}, {os.path.basename(write_csv)}")
    if workload == "raster_fwdbwd":
        data["_binary"]["raster_fwd"], data["_source"]["raster_fwd"] = data["_binary"][workload], data["_source"][workload]
    if sq_csv and os.path.exists(sq_csv):
        data.setdefault("valu", {})[workload] = valu_summary(sq_csv)
        if workload == "raster_fwdbwd":
            data["valu"]["raster_fwd"] = {k: v for k, v in data["valu"][workload].items() if "bwd" not in k}
    raw = data.setdefault("raw_KiB", {})
    if not isinstance(raw.get(workload), dict) or "FETCH_SIZE" in raw.get(workload, {}):
        raw = data["raw_KiB"] = {k: v for k, v in raw.items() if isinstance(v, dict) and "FETCH_SIZE" not in v}
    raw[workload] = {k: {"FETCH_SIZE": fetch.get(k), "WRITE_SIZE": write.get(k)} for k in sorted(set(fetch) | set(write))}
    json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(data[workload], indent=1))


if __name__ == "__main__":
    main()
