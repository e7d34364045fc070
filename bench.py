#!/usr/bin/env python3
"""Headline benchmark: MPC steps solved per second, hanging_chain horizon 60, cold + warm.

One "step" of this bench = one pass of the PANTR hot path over the replay batch of
BASELINE.json configs[1]: the 300 timed MPC time steps of one closed-loop run
(num_sim=300, num_dist=5, u_dist=-1,1,1; new-benchmarks-paper/dodo.py:12-33,94-95), each
solved twice — warm-started (previous solution shifted by nu, mpc-driver.cpp:168-175) and
cold-started (u = 0, mpc-driver.cpp:192-197) — i.e. 600 independent PANTR solves in one
kernel launch per GPU.  Inputs (x0, initial guesses) are resident in HBM when the timed
region starts; every timed pass resets the guesses with a device-to-device copy and runs
the complete solves (no cached outputs).

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver with torch.distributed.run, one rank per GPU; the path has
no exchange step, so ranks only meet in the barriers around the timed region and in the
MAX-reduction of the elapsed time ("weak" scaling: every rank solves the full batch).

Rank 0 prints one JSON line with `roofline` (algorithmic bytes of SURVEY.md section 8(d) /
HIP-event kernel time, against the 8 TB/s HBM peak) and, at N = 1, `cpu_baseline` (the CPU
oracle, a port of the algorithm, timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))

METRIC = "MPC-steps solved/sec, hanging_chain N=60 cold+warm, 1/2/4/8 MI355X"
HORIZON = 60
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# PANTR cold+warm combined on the reference's (unstated) CPU, 1 thread
BASELINE_STEPS_PER_S = 4.21
FIXTURE = os.path.join(ROOT, "tests", "golden", "mpc_hanging_chain_N60_warm300.npz")
# PMC summary of the same command (scripts/profile.sh + scripts/summarize_profiles.py)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r01_summary.json")


def measured_traffic(instances):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this workload, or None."""
    try:
        with open(PMC_SUMMARY) as f:
            d = json.load(f)
        if int(d["pmc_per_launch"]["SQ_WAVES"]) != instances:
            return None
        return float(d["hbm_traffic_bytes_per_launch"]["total"])
    except (OSError, KeyError, ValueError):
        return None


def load_workload():
    d = np.load(FIXTURE)
    x0 = np.vstack([d["x0"], d["x0"]])
    guess = np.vstack([d["guess"], np.zeros_like(d["guess"])])
    return np.ascontiguousarray(x0), np.ascontiguousarray(guess), d["sol"]


def cpu_baseline(x0, guess, cores):
    """Times the CPU oracle (kind "port") on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob  # the checker / CPU baseline; never on the product path
    idx = np.arange(len(x0))  # the whole workload: 300 warm + 300 cold solves
    t = time.perf_counter()
    _, stats = ob.solve_batch(HORIZON, x0[idx], guess[idx], ob.dodo_params(), nthreads=cores)
    dt = time.perf_counter() - t
    ok = sum(s["status"] == 1 for s in stats)
    return {
        "value": len(idx) / dt, "unit": "steps/s", "cores": cores, "kind": "port",
        "sample": f"the whole workload once ({len(idx) // 2} warm + {len(idx) // 2} cold solves, {ok} converged) "
                  f"over {cores} host threads, {dt:.2f} s wall; single-thread solve time summed: "
                  f"{sum(s['elapsed_s'] for s in stats):.2f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent passes in flight per GPU, each on its own HIP stream and solver handle "
                         "(default 1: one pass after the other, the configuration the roofline line describes)")
    args = ap.parse_args()

    import torch
    import pantr_amd as pa
    from pantr_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if distributed and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the PANTR solver has no CPU path")
    # Rehearsal aid for the multi-rank path on a box with fewer GPUs than ranks (several ranks share
    # a device, rendezvous over gloo): PANTR_BENCH_REHEARSAL=1.  Never set by the driver.
    rehearsal = os.environ.get("PANTR_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl"); used for barriers + one MAX reduction only
        dist.init_process_group(backend="gloo" if rehearsal else "nccl")

    x0_h, guess_h, sol_warm = load_workload()
    B, n = guess_h.shape
    dev = torch.device("cuda", local_rank)
    x0 = torch.from_numpy(x0_h).to(dev)
    guess = torch.from_numpy(guess_h).to(dev)
    S = max(1, args.streams)
    us = [torch.empty_like(guess) for _ in range(S)]
    stats_ts = [torch.zeros(B * pa.STATS_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in range(S)]
    batches = [pa.PantrBatch("hanging_chain", HORIZON, B, device=local_rank, opts=pa.DODO_PANTR_OPTS)
               for _ in range(S)]
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(dev) for _ in range(S - 1)]
    u, stats_t, batch = us[0], stats_ts[0], batches[0]

    def one_pass(i=0):
        k = i % S
        with torch.cuda.stream(streams[k]):
            us[k].copy_(guess)  # device-to-device reset of the in/out guesses
            batches[k].solve_device(B, x0.data_ptr(), us[k].data_ptr(), stats_ts[k].data_ptr(),
                                    streams[k].cuda_stream)  # m = 0: no y

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(max(args.warmup, S if S > 1 else 0)):
        one_pass(i)
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_pass(i)
        if S == 1:
            kernel_ms.append(batch.last_kernel_ms())  # HIP events on the launch stream (synchronises)
    barrier()
    elapsed = time.perf_counter() - t0
    if S > 1:  # the last launch of every stream (durations overlap: they are latencies, not shares)
        kernel_ms = [b.last_kernel_ms() for b in batches[:min(S, args.steps)]]
    elapsed = sharding.max_over_ranks(elapsed, None if rehearsal else dev)  # MAX over ranks (identity at N = 1)

    stats = np.frombuffer(stats_t.cpu().numpy().tobytes(), dtype=pa.STATS_DTYPE)
    converged = int(np.sum(stats["status"] == 1))
    u_h = u.cpu().numpy()
    max_dev = float(np.max(np.abs(u_h[:B // 2] - sol_warm)))  # warm half vs the recorded solutions

    if rank == 0:
        total_steps = world * B * args.steps
        alg_bytes = pa.algorithmic_bytes(stats, HORIZON, batch.nx, batch.nu)
        k_ms = float(np.mean(kernel_ms))
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        value = total_steps / elapsed
        line = {
            "metric": METRIC, "value": value, "unit": "steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": value / BASELINE_STEPS_PER_S, "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "hanging_chain horizon=60 replay batch: 300 MPC time steps of one closed "
                            "loop (num_sim=300 num_dist=5 u_dist=-1,1,1), each solved warm and cold = "
                            "600 PANTR solves per GPU per pass, tolerance 1e-8, dodo.py PANTR options",
                "instances_per_gpu": B, "horizon": HORIZON, "n": n, "streams": S,
                "converged": converged, "max_abs_dev_from_recorded_warm_solutions": max_dev,
                "vs_baseline_ref": "BASELINE.md 1.1: 4.21 steps/s (alpaqa PANTR cold+warm, unstated CPU, 1 thread)",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(B),
                "kernel": "pantr_solve_kernel<ChainModel, 0>", "kernel_ms": k_ms,
                "algorithmic_bytes_per_launch": int(alg_bytes),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            # the GPU box's CPU share for one GPU is 16 hardware threads
            cores = min(16, len(os.sched_getaffinity(0)))
            line["cpu_baseline"] = cpu_baseline(x0_h, guess_h, cores)
        print(json.dumps(line), flush=True)
    for b in batches:
        b.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
