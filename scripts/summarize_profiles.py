#!/usr/bin/env python3
"""Copies the rocprofv3 summaries of scripts/profile.sh from gpurun_out/prof into profiles/
(tracked) and writes profiles/<tag>_summary.json with per-launch numbers for the solve kernel.

Units/corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE counts 64 B per 128-B read request, so it is doubled before use
(an upper bound for this kernel's 8-B-per-lane row reads, which are uncalibrated).
"""
import csv, glob, json, os, shutil, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa


def newest(pattern):
    files = glob.glob(pattern)
    return max(files, key=os.path.getmtime) if files else None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
KERNEL = "pantr_solve_"  # pantr_solve_kernel<ChainModel, 0> (one wavefront per instance) | pantr_solve_spec_kernel<ChainModel> (two)
INSTANCES = 600           # the timed launches: 600 instances, one workgroup each


def timed_grid():
    """Grid size (threads) of the timed launches: 600 workgroups of 64 or - with speculation - 128 threads."""
    t = newest(os.path.join(src, "stats", "*", "*_kernel_trace.csv"))
    for r in csv.DictReader(open(t)):
        if KERNEL in r["Kernel_Name"] and int(r["Grid_Size_X"]) in (INSTANCES * 64, INSTANCES * 128):
            return int(r["Grid_Size_X"]), r["Kernel_Name"]
    raise SystemExit("no timed launch of the solve kernel in the trace")


def gpu_agent():
    """Compute units, SIMDs per CU and the engine clock (MHz) of the profiled GPU, from the trace's agent table."""
    f = newest(os.path.join(src, "stats", "*", "*_agent_info.csv"))
    for r in csv.DictReader(open(f)):
        if r["Agent_Type"] == "GPU":
            return int(r["Cu_Count"]), int(r["Simd_Per_Cu"]), float(r["Max_Engine_Clk_Fcompute"])
    raise SystemExit("no GPU agent in the trace")


GRID, KERNEL_FULL = timed_grid()
CUS, SIMDS_PER_CU, CLK_MHZ = gpu_agent()
summary = {"tag": tag, "command": "python3 bench.py --steps 3 --warmup 1 --headline-only",
           "kernel": KERNEL_FULL, "wavefronts_per_instance": GRID // (INSTANCES * 64),
           "gpu": {"compute_units": CUS, "simds_per_cu": SIMDS_PER_CU, "engine_clock_mhz": CLK_MHZ},
           # the code these numbers describe: bench.py prints roofline.traffic only while csrc/ still hashes to this
           "csrc_sha256": pa.source_hash()}
stats = newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
for r in csv.DictReader(open(stats)):
    if KERNEL in r["Name"]:
        summary["kernel_trace"] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                   "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6,
                                   "percentage": float(r["Percentage"])}
trace = newest(os.path.join(src, "stats", "*", "*_kernel_trace.csv"))
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(trace))
        if KERNEL in r["Kernel_Name"] and int(r["Grid_Size_X"]) == GRID]
summary["kernel_trace_timed_launches"] = {"grid_size": GRID, "calls": len(durs), "avg_ms": sum(durs) / max(len(durs), 1),
                                          "min_ms": min(durs, default=0.0), "max_ms": max(durs, default=0.0)}
for r in csv.DictReader(open(trace)):
    if KERNEL in r["Kernel_Name"] and int(r["Grid_Size_X"]) == GRID:
        summary["dispatch"] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size",
                                                  "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count",
                                                  "SGPR_Count") if k in r}
        break
pmc = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    f = newest(os.path.join(src, name, "*", "*_counter_collection.csv"))
    if not f:
        continue
    agg = {}
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"] and int(r.get("Grid_Size", GRID)) == GRID:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc[k] = sum(v) / len(v)
summary["pmc_per_launch"] = pmc
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    summary["hbm_traffic_bytes_per_launch"] = {
        "read_corrected": 2 * pmc["FETCH_SIZE"] * 1024, "write": pmc["WRITE_SIZE"] * 1024,
        "total": (2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024,
        "note": "KiB -> bytes; FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request)"}
if "SQ_INSTS_VALU" in pmc and "SQ_WAVE_CYCLES" in pmc and summary.get("kernel_trace_timed_launches", {}).get("avg_ms"):
    # what actually bounds the kernel: VALU issue of lone wavefronts (MI355X_MICROARCH.md: SQ_WAVE_CYCLES counts
    # quad-cycles; a 64-wide fp64 / DPP instruction occupies a SIMD for 4 cycles; CUs, SIMDs and clock from the trace's agent table)
    ms = summary["kernel_trace_timed_launches"]["avg_ms"]
    simd_cycles = CUS * SIMDS_PER_CU * ms * 1e-3 * CLK_MHZ * 1e6
    wave_cycles = 4 * pmc["SQ_WAVE_CYCLES"]
    summary["valu_issue"] = {
        "valu_wave_instructions_per_launch": pmc["SQ_INSTS_VALU"],
        "simd_issue_slots_used_frac": 4 * pmc["SQ_INSTS_VALU"] / simd_cycles,
        "wave_slot_occupancy_frac": wave_cycles / simd_cycles,
        "cycles_per_valu_instruction_of_a_wavefront": wave_cycles / pmc["SQ_INSTS_VALU"],
        "mean_wavefront_ms": wave_cycles / pmc["SQ_WAVES"] / (CLK_MHZ * 1e6) * 1e3,
        "note": "one wavefront per instance: the pass lasts as long as its slowest instance, so most SIMD cycles idle"}
if "SQ_LDS_IDX_ACTIVE" in pmc and summary.get("kernel_trace_timed_launches", {}).get("avg_ms"):
    ms = summary["kernel_trace_timed_launches"]["avg_ms"]
    cu_cycles = CUS * ms * 1e-3 * CLK_MHZ * 1e6
    summary["lds"] = {
        "lds_busy_frac_of_cu_cycles": pmc["SQ_LDS_IDX_ACTIVE"] / cu_cycles,
        "bank_conflict_frac_of_lds_cycles": pmc.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(pmc["SQ_LDS_IDX_ACTIVE"], 1.0),
        "lds_instructions_per_launch": pmc.get("SQ_INSTS_LDS"),
        "note": "SQ_LDS_IDX_ACTIVE / (CUs x kernel time x engine clock): averaged over the whole pass, including its "
                "tail when few instances are left; while all 600 run the fraction is about pass / mean-solve-time higher"}
json.dump(summary, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
