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
KERNELS = ("pantr_solve_", "pantr_alm_")  # the solve kernels of the hanging chain / of the quadcopter family


def newest(pattern):
    files = glob.glob(pattern)
    return max(files, key=os.path.getmtime) if files else None


for w in workloads:
    src = os.path.join(ROOT, "gpurun_out", "traffic", w)
    line = [l for l in open(os.path.join(src, "stats.log")) if l.startswith("{")][-1]
    run = json.loads(line)
    out = {"tag": tag, "workload": w, "command": f"python3 scripts/traffic_workload.py {w}", "run": run}
    # the timed launches: the solve kernel at the workload's batch size (one workgroup of 64 or 128 threads per instance) - a
    # workload that first records its closed loops launches the same kernel many times on single instances
    grids = {run["instances"] * 64, run["instances"] * 128}
    timed = lambda name, grid: any(k in name for k in KERNELS) and "eval" not in name and int(grid) in grids
    rows = [r for r in csv.DictReader(open(newest(os.path.join(src, "stats", "*", "*_kernel_trace.csv"))))
            if timed(r["Kernel_Name"], r["Grid_Size_X"])]
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    # launch_order=auto ranks the instances from the solve before: the first launch has nothing to go by and is left
    # out of the durations where the later ones were ranked (the bytes do not depend on the order)
    src_ = run.get("launch_order", [])
    if len(src_) == len(durs) and len(durs) > 1 and src_[0] != src_[-1]:
        durs = [d for d, s_ in zip(durs, src_) if s_ == src_[-1]]
        out["launch_order"] = src_
    out["kernel_trace"] = {"kernel": rows[0]["Kernel_Name"], "calls": len(durs), "avg_ms": sum(durs) / len(durs),
                           "min_ms": min(durs), "max_ms": max(durs), "source": "rocprofv3 --kernel-trace: the launches at the workload's grid size"}
    pmc = {}
    for name in ("pmc_fetch", "pmc_write"):
        f = newest(os.path.join(src, name, "*", "*_counter_collection.csv"))
        rows = [r for r in csv.DictReader(open(f)) if timed(r["Kernel_Name"], r["Grid_Size"])]
        vals = [float(r["Counter_Value"]) for r in rows]
        pmc[rows[0]["Counter_Name"]] = {"per_launch_KiB": sum(vals) / len(vals), "launches": len(vals)}
    # the occupancy / issue counters, where the passes were taken (SQ=1 scripts/traffic_profile.sh ...)
    for name in ("pmc_sq", "pmc_sq2"):
        f = newest(os.path.join(src, name, "*", "*_counter_collection.csv"))
        if not f:
            continue
        rows = [r for r in csv.DictReader(open(f)) if timed(r["Kernel_Name"], r["Grid_Size"])]
        for cname in sorted({r["Counter_Name"] for r in rows}):
            vals = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == cname]
            # (one row per dispatch and counter; rocprofv3 sums a counter over the chip's shader engines)
            pmc[cname] = {"per_launch": sum(vals) / len(vals), "launches": len(vals)}
    out["pmc"] = pmc
    if "SQ_INSTS_VALU" in pmc and "SQ_BUSY_CYCLES" in pmc:
        ms_ = out["kernel_trace"]["avg_ms"]
        valu, waves = pmc["SQ_INSTS_VALU"]["per_launch"], pmc["SQ_WAVES"]["per_launch"]
        out["valu_issue"] = {
            "valu_instructions_per_wavefront": valu / waves,
            # a 64-wide VALU instruction occupies its SIMD for 4 cycles; 256 CUs x 4 SIMDs; clock from the wave-cycle counters is not
            # available here, so the fraction is stated against the kernel's duration at 2.4 GHz
            "simd_issue_slots_used_frac": 4.0 * valu / (256 * 4 * 2.4e9 * ms_ * 1e-3),
            "wait_any_over_busy": pmc["SQ_WAIT_ANY"]["per_launch"] / pmc["SQ_BUSY_CYCLES"]["per_launch"] if "SQ_WAIT_ANY" in pmc else None,
        }
        if "SQ_WAVE_CYCLES" in pmc:
            out["valu_issue"]["sq_wave_cycles_over_insts_valu_in_counter_units"] = pmc["SQ_WAVE_CYCLES"]["per_launch"] / valu
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
    out["bytes_per_product"] = (rd + wr) / max(run["n_hess_psi_prod_total"], 1)
    dst = os.path.join(ROOT, "profiles", f"{tag}_{w}_traffic.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(w, f"kernel {ms:.1f} ms, read {rd/1e9:.1f} GB + write {wr/1e9:.1f} GB = {(rd+wr)/1e9:.1f} GB per launch "
             f"({out['traffic_GBps']:.0f} GB/s), algorithmic {alg/1e9:.1f} GB ({out['algorithmic_GBps']:.0f} GB/s): ratio {out['traffic_over_algorithmic']:.2f}, "
             f"{out['bytes_per_product']/1e3:.0f} KB per product")
