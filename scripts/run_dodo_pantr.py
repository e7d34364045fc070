#!/usr/bin/env python3
"""The PANTR rows of new-benchmarks-paper/dodo.py on the MI355X solver (BASELINE.json configs[3]).

1. Runs every task of mpc_benchmark(problem) (dodo.py:69-111, PANTR only) through the C++ CLI
   clone `benchmark-mpc-driver` exactly as run-single-mpc-exp.sh:19 does: one closed loop per
   (horizon, warm/cold), results written as <outdir>/<name>.py (+ .txt with stdout/stderr), the
   files perfplot-cold-warm.py / benchmark_util.py:4-9 consume.
2. Replays the whole sweep as batches (one per horizon: its 60 warm + 60 cold recorded steps),
   checks that the batched solutions equal the closed-loop ones, and reports steps/s;
3. then launches all horizons at once, each on its own HIP stream (hanging_chain).

    python scripts/run_dodo_pantr.py --outdir output [--problem hanging_chain] [--num-sim 60]
                                     [--method pantr|panoc|strucpanoc]   (the comparators: dodo.py:34-40)
"""
import argparse
import importlib.util
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa  # noqa: E402
from pantr_amd import workloads  # noqa: E402

DRIVER = os.path.join(ROOT, "pantr-cdc2023-experiments_amd", "bin", "benchmark-mpc-driver")


def load_results(path, name):  # benchmark_util.py:4-9
    spec = importlib.util.spec_from_file_location("", os.path.join(path, f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.results


def first_guess(batch, num_dist, u_dist):
    """Warm start of the first timed step: the unrecorded first solve's solution, shifted."""
    x = batch.initial_state()[None, :]
    for _ in range(num_dist):
        x = batch.simulate_dynamics(x, np.asarray(u_dist, float)[None, :])
    u, _ = batch.solve(x, batch.initial_guess()[None, :])
    return np.concatenate([u[0, batch.nu:], u[0, -batch.nu:]])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--outdir", default="output")
    ap.add_argument("--problem", default="hanging_chain")
    ap.add_argument("--num-sim", type=int, default=workloads.NUM_SIM)
    ap.add_argument("--horizons", type=int, nargs="*", default=list(workloads.HORIZONS))
    ap.add_argument("--method", default="pantr", choices=sorted(pa.DODO_OPTS))
    args = ap.parse_args()
    if args.problem == "hanging_chain":
        # step 3 keeps its inputs in torch tensors: torch must bring its HIP runtime into the
        # process before the solver library does (two runtimes in one process do not share the GPU)
        import torch
        torch.cuda.is_available()
    os.makedirs(args.outdir, exist_ok=True)
    tasks = workloads.dodo_tasks(args.problem, args.method, args.horizons, args.num_sim)
    summary = {"tasks": [], "problem": args.problem, "method": args.method}

    # ---- 1. closed loops through the CLI, one process per task as doit runs them ----
    al = args.problem != "hanging_chain"  # general constraints: the CLI replays each task itself
    for t in tasks:
        t0 = time.perf_counter()
        with open(os.path.join(args.outdir, t["name"] + ".txt"), "w") as log:
            rc = subprocess.run([DRIVER] + t["argv"] + (["replay=true"] if al else []), cwd=args.outdir,
                                stdout=log, stderr=subprocess.STDOUT).returncode
        if rc != 0:
            raise SystemExit(f"{t['name']}: driver exited with {rc}")
        res = load_results(args.outdir, t["name"])
        rt = np.abs(res["runtimes"])
        summary["tasks"].append({"name": t["name"], "horizon": t["horizon"], "warm": t["warm"],
                                 "avg_runtime_s": float(rt.mean()), "p95_runtime_s": float(np.quantile(rt, .95)),
                                 "converged": int((res["runtimes"] > 0).sum()), "num_sim": len(rt),
                                 "hess_psi_prod": int(res["evaluations"]["hess_ψ_prod"]),
                                 "wall_s": time.perf_counter() - t0})
        line = f"{t['name']}: avg {rt.mean()*1e3:.2f} ms/step, {summary['tasks'][-1]['converged']}/{len(rt)} converged"
        if al:
            rep = [l for l in open(os.path.join(args.outdir, t["name"] + ".txt")) if l.startswith("replay:")]
            summary["tasks"][-1]["replay"] = rep[-1].strip() if rep else None
            line += " | " + (rep[-1].strip() if rep else "no replay line")
        print(line)
    if al:
        with open(os.path.join(args.outdir, "summary.json"), "w") as f:
            json.dump(summary, f, indent=1)
        return

    # ---- 2. the same steps as batches, one launch per horizon ----
    total_steps, total_kernel_s = 0, 0.0
    replay_inputs = {}
    for N in args.horizons:
        warm = load_results(args.outdir, workloads.problem_name(args.problem, True, args.method, N, args.num_sim))
        cold = load_results(args.outdir, workloads.problem_name(args.problem, False, args.method, N, args.num_sim))
        ns = args.num_sim
        batch = pa.PantrBatch(args.problem, N, 2 * ns, opts=pa.DODO_OPTS[args.method])
        sol_w, sol_c = warm["inputs"].T, cold["inputs"].T
        guess_w = np.vstack([first_guess(batch, 5, (-1.0, 1.0, 1.0))[None, :],
                             np.hstack([sol_w[:-1, batch.nu:], sol_w[:-1, -batch.nu:]])])
        x0 = np.vstack([warm["states"].T[:ns], cold["states"].T[:ns]])
        guess = np.vstack([guess_w, np.zeros_like(sol_c)])
        u, st = batch.solve(x0, guess)
        ms = batch.last_kernel_ms()
        batch.close()
        # (np.vstack of transposed views may come out Fortran-ordered: the device path takes raw pointers)
        replay_inputs[N] = tuple(np.ascontiguousarray(a) for a in (x0, guess, np.vstack([sol_w, sol_c])))
        same = np.array_equal(u, np.vstack([sol_w, sol_c]))
        print(f"N={N}: {2*ns} steps in one launch, {ms:.1f} ms, {int((st['status']==1).sum())} converged, "
              f"solutions {'identical to' if same else 'DIFFERENT from'} the closed loops")
        if not same:
            raise SystemExit("replay batch does not reproduce the closed loops")
        total_steps += 2 * ns
        total_kernel_s += ms * 1e-3
    summary["replay"] = {"steps": total_steps, "kernel_s": total_kernel_s,
                         "steps_per_s": total_steps / total_kernel_s}
    print(f"sweep replay: {total_steps} steps in {total_kernel_s*1e3:.1f} ms of kernels "
          f"= {total_steps/total_kernel_s:.0f} steps/s")

    # ---- 3. the whole sweep at once: one launch per horizon, each on its own HIP stream ----
    # (inputs resident in HBM; the horizons' kernels share the chip instead of queueing)
    import torch
    dev = torch.device("cuda", 0)
    jobs = []
    for N, (x0, guess, ref) in replay_inputs.items():
        batch = pa.PantrBatch(args.problem, N, len(x0), opts=pa.DODO_OPTS[args.method])
        jobs.append(dict(N=N, batch=batch, ref=ref, x0=torch.from_numpy(x0).to(dev),
                         guess=torch.from_numpy(guess).to(dev), u=torch.empty(guess.shape, dtype=torch.float64, device=dev),
                         stats=torch.zeros(len(x0) * pa.STATS_DTYPE.itemsize, dtype=torch.uint8, device=dev),
                         stream=torch.cuda.Stream(dev)))
    jobs.sort(key=lambda j: -j["N"])  # longest first
    # at most four streams (the runtime's hardware queues per process: launches on streams that share one run one after
    # the other): the fifth and sixth horizon go behind the shortest of the first four
    for k, j in enumerate(jobs[4:]):
        j["stream"] = jobs[3 - k % 4]["stream"]
    best = None
    for rep in range(3):
        for j in jobs:
            j["u"].copy_(j["guess"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in jobs:
            j["batch"].solve_device(len(j["ref"]), j["x0"].data_ptr(), j["u"].data_ptr(), j["stats"].data_ptr(),
                                    j["stream"].cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    for j in jobs:
        stj = np.frombuffer(j["stats"].cpu().numpy().tobytes(), dtype=pa.STATS_DTYPE)
        j["same"] = np.array_equal(j["u"].cpu().numpy(), j["ref"])
        if not j["same"]:
            bad = np.flatnonzero(np.any(j["u"].cpu().numpy() != j["ref"], axis=1))
            print(f"  N={j['N']}: {len(bad)} instances differ (first {bad[:5]}), statuses {np.unique(stj['status'])}")
    same = all(j["same"] for j in jobs)
    for j in jobs:
        j["batch"].close()
    print(f"sweep replay on {len(jobs)} concurrent streams: {total_steps} steps in {best*1e3:.1f} ms wall "
          f"= {total_steps/best:.0f} steps/s, solutions {'identical to' if same else 'DIFFERENT from'} the closed loops")
    if not same:
        raise SystemExit("concurrent replay does not reproduce the closed loops")
    summary["replay_concurrent_streams"] = {"steps": total_steps, "wall_s": best, "steps_per_s": total_steps / best}
    with open(os.path.join(args.outdir, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
