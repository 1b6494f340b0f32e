"""Turn rocprofv3 PMC passes into profiles/pmc_latest.json (the `roofline.traffic` source of bench.py).

Usage: python tools/pmc_extract.py <workload> <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv>

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


def main():
    workload, fetch_csv, write_csv = sys.argv[1:4]
    tag = sys.argv[4] if len(sys.argv) > 4 else "unknown"
    fetch, write = averages(fetch_csv, "FETCH_SIZE"), averages(write_csv, "WRITE_SIZE")
    out_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    data["_note"] = ("HBM bytes per launch from rocprofv3 PMC (separate FETCH_SIZE and WRITE_SIZE passes, values in KiB; "
                     "FETCH doubled per MI355X_MICROARCH.md HBM section; gather-heavy kernels are uncalibrated)")
    data[workload] = {k: int(round((2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024)) for k in sorted(set(fetch) | set(write))}
    if workload == "raster_fwdbwd":              # the forward kernels of the same launches
        data["raster_fwd"] = {k: v for k, v in data[workload].items() if "bwd" not in k}
    data.setdefault("_binary", {})[workload] = tag
    data.setdefault("_source", {})[workload] = (f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) -- python3 bench.py "
                                                f"--workload {workload}; CSVs: {os.path.basename(fetch_csv)}, {os.path.basename(write_csv)}")
    if workload == "raster_fwdbwd":
        data["_binary"]["raster_fwd"], data["_source"]["raster_fwd"] = data["_binary"][workload], data["_source"][workload]
    raw = data.setdefault("raw_KiB", {})
    if not isinstance(raw.get(workload), dict) or "FETCH_SIZE" in raw.get(workload, {}):
        raw = data["raw_KiB"] = {k: v for k, v in raw.items() if isinstance(v, dict) and "FETCH_SIZE" not in v}
    raw[workload] = {k: {"FETCH_SIZE": fetch.get(k), "WRITE_SIZE": write.get(k)} for k in sorted(set(fetch) | set(write))}
    json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(data[workload], indent=1))


if __name__ == "__main__":
    main()
