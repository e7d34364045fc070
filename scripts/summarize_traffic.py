#!/usr/bin/env python3
"""Digest of scripts/traffic_profile.sh (gpurun_out/traffic/<workload>/) into profiles/<tag>_<workload>_traffic.json:
per launch of the solve kernel its duration (rocprofv3 --kernel-trace --stats), FETCH_SIZE and WRITE_SIZE
(separate --pmc passes), the HBM-side bytes they stand for (MI355X_MICROARCH.md, HBM section: both counters are in
KiB; on gfx950 FETCH_SIZE tallies 64 B per 128-B request, so it is doubled - exact for 16-B-per-lane streaming
reads, an upper bound for this kernel's mix of LDS-DMA rows (16 B per lane) and 8-B-per-lane row reads), and their
ratio to the algorithmic bytes the workload script printed.  usage: summarize_traffic.py <tag> <workload> ..."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, workloads = sys.argv[1], sys.argv[2:]
KERNEL = "pantr_solve_"


def newest(pattern):
    files = glob.glob(pattern)
    return max(files, key=os.path.getmtime) if files else None


for w in workloads:
    src = os.path.join(ROOT, "gpurun_out", "traffic", w)
    line = [l for l in open(os.path.join(src, "stats.log")) if l.startswith("{")][-1]
    run = json.loads(line)
    out = {"tag": tag, "workload": w, "command": f"python3 scripts/traffic_workload.py {w}", "run": run}
    for r in csv.DictReader(open(newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv")))):
        if KERNEL in r["Name"]:
            out["kernel_trace"] = {"kernel": r["Name"], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                   "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6}
    pmc = {}
    for name in ("pmc_fetch", "pmc_write"):
        f = newest(os.path.join(src, name, "*", "*_counter_collection.csv"))
        rows = [r for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"]]
        vals = [float(r["Counter_Value"]) for r in rows]
        pmc[rows[0]["Counter_Name"]] = {"per_launch_KiB": sum(vals) / len(vals), "launches": len(vals)}
    out["pmc"] = pmc
    rd, wr = 2 * pmc["FETCH_SIZE"]["per_launch_KiB"] * 1024, pmc["WRITE_SIZE"]["per_launch_KiB"] * 1024
    ms = out["kernel_trace"]["avg_ms"]
    alg = run["algorithmic_bytes_per_launch"]
    out["hbm_traffic_bytes_per_launch"] = {"read_corrected": rd, "write": wr, "total": rd + wr,
                                           "note": "KiB -> bytes; FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request)"}
    out["traffic_over_algorithmic"] = (rd + wr) / alg
    out["algorithmic_GBps"] = alg / (ms * 1e-3) / 1e9
    out["traffic_GBps"] = (rd + wr) / (ms * 1e-3) / 1e9
    out["frac_of_8TBps_algorithmic"] = alg / (ms * 1e-3) / 8e12
    out["frac_of_8TBps_traffic"] = (rd + wr) / (ms * 1e-3) / 8e12
    out["bytes_per_product"] = (rd + wr) / run["n_hess_psi_prod_total"]
    dst = os.path.join(ROOT, "profiles", f"{tag}_{w}_traffic.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(w, f"kernel {ms:.1f} ms, read {rd/1e9:.1f} GB + write {wr/1e9:.1f} GB = {(rd+wr)/1e9:.1f} GB per launch "
             f"({out['traffic_GBps']:.0f} GB/s), algorithmic {alg/1e9:.1f} GB ({out['algorithmic_GBps']:.0f} GB/s): ratio {out['traffic_over_algorithmic']:.2f}, "
             f"{out['bytes_per_product']/1e3:.0f} KB per product")
