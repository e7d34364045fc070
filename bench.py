#!/usr/bin/env python3
"""Headline benchmark: MPC steps solved per second, hanging_chain horizon 60, cold + warm.

One "step" of this bench = one pass of the PANTR hot path over the replay batch of
BASELINE.json configs[1]: the 300 timed MPC time steps of one closed-loop run
(num_sim=300, num_dist=5, u_dist=-1,1,1; new-benchmarks-paper/dodo.py:12-33,94-95), each
solved twice — warm-started (previous solution shifted by nu, mpc-driver.cpp:168-175) and
cold-started (u = 0, mpc-driver.cpp:192-197) — i.e. 600 independent PANTR solves in one
kernel launch per GPU.  The batch is produced by the product's own closed loop on the device
(pantr_batch_mpc_record) before anything is timed; inputs (x0, initial guesses) are resident in
HBM when the timed region starts; every timed pass resets the guesses with a device-to-device
copy and runs the complete solves (no cached outputs).

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver with torch.distributed.run, one rank per GPU; the path has no
exchange step, so ranks only meet in the barriers around the timed region and in the
MAX-reduction of the elapsed time.  Default mode: every rank solves its own full batch
("weak": per-GPU work fixed).  `--workload sweep|stress --scaling strong` deals ONE batch
(BASELINE.json configs[3] / configs[4]) to the ranks longest-first (pantr_amd/sharding.py),
gathers the solutions and checks them against the unsharded solve.

Rank 0 prints ONE JSON line.  At N = 1 it carries, next to the headline: `roofline`
(algorithmic bytes of SURVEY.md 8(d) / HIP-event kernel time, against the 8 TB/s spec peak and
the measured device-copy peak), `cpu_baseline` (the CPU oracle, a port of the algorithm, on the
host cores), `warm` / `cold` separately, and `configs` — the other BASELINE.json configurations
(horizon sweep, realistic_quadcopter N=60, quadcopter N=12, N=120 stress), each with its own
steps/s, dominant kernel, roofline and CPU baseline on the same threads — and the serial
one-thread closed loop on the CPU next to the reference's published per-step times.
`--headline-only` skips everything but the headline (profiling runs).
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
HBM_COPY_GUIDE_GBS = 6290.0  # MI355X_MICROARCH.md:36: 6.29 TB/s measured with a float4 copy (79 % of the spec peak)
FIXTURE = os.path.join(ROOT, "tests", "golden", "mpc_hanging_chain_N60_warm300.npz")
PROFILE_TAG = "r05"
# PMC summary of the headline command (scripts/profile.sh + scripts/summarize_profiles.py)
PMC_SUMMARY = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_summary.json")
# published mean solver time per MPC step, alpaqa PANTR, unstated CPU, 1 thread
PUBLISHED_S_PER_STEP = {
    ("hanging_chain", 60): {"cold": 0.3454, "warm": 0.1301},
    ("realistic_quadcopter", 60): {"cold": 0.3162, "warm": 0.07881},
    ("quadcopter", 60): {"cold": 0.1379, "warm": 0.02738},
}


def measured_traffic(instances):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of the headline command, with
    where they come from.  None unless the summary was taken at exactly this csrc/ (hash) and batch."""
    import pantr_amd as pa
    src = {"file": os.path.relpath(PMC_SUMMARY, ROOT), "csrc_sha256": pa.source_hash()}
    try:
        with open(PMC_SUMMARY) as f:
            d = json.load(f)
        src["profiled_csrc_sha256"] = d.get("csrc_sha256")
        if d.get("csrc_sha256") != src["csrc_sha256"]:
            src["refused"] = "the kernels changed since the PMC passes were taken"
            return None, src
        if int(d["pmc_per_launch"]["SQ_WAVES"]) != instances * int(d.get("wavefronts_per_instance", 1)):
            src["refused"] = "the PMC passes were taken on another batch size"
            return None, src
        if "valu_issue" in d:  # what binds instead of HBM: VALU issue of lone wavefronts (same PMC passes)
            src["valu_issue"] = {k: d["valu_issue"][k] for k in ("simd_issue_slots_used_frac", "wave_slot_occupancy_frac",
                                                                 "cycles_per_valu_instruction_of_a_wavefront")}
        return float(d["hbm_traffic_bytes_per_launch"]["total"]), src
    except (OSError, KeyError, ValueError) as e:
        src["refused"] = f"no usable summary ({type(e).__name__})"
        return None, src


def measured_workload_traffic(name, instances):
    """HBM bytes per launch of a memory-bound workload from its committed rocprofv3 PMC passes
    (scripts/traffic_profile.sh + scripts/summarize_traffic.py -> profiles/<tag>_<name>_traffic.json), with where they
    come from; None unless the summary was taken at exactly this tree's sources (hash) and batch size."""
    import pantr_amd as pa
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{name}_traffic.json")
    src = {"file": os.path.relpath(path, ROOT), "csrc_sha256": pa.source_hash()}
    try:
        with open(path) as f:
            d = json.load(f)
        src["profiled_csrc_sha256"] = d["run"].get("source_hash")
        if src["profiled_csrc_sha256"] != src["csrc_sha256"]:
            src["refused"] = "the kernels changed since the PMC passes were taken"
            return None, src
        if int(d["run"]["instances"]) != instances:
            src["refused"] = "the PMC passes were taken on another batch size"
            return None, src
        src["traffic_over_algorithmic"] = d["traffic_over_algorithmic"]
        src["profiled_kernel_ms"] = d["kernel_trace"]["avg_ms"]
        if "valu_issue" in d:  # the occupancy / issue counters of the same command (SQ=1 scripts/traffic_profile.sh)
            src["valu_issue"] = d["valu_issue"]
        return float(d["hbm_traffic_bytes_per_launch"]["total"]), src
    except (OSError, KeyError, ValueError) as e:
        src["refused"] = f"no usable summary ({type(e).__name__})"
        return None, src


class Resident:
    """A replay batch resident in HBM with its solver handle: launch() = reset the guesses (device to
    device) + one solve kernel over the whole batch on `stream`."""

    def __init__(self, w, torch, pa, dev, local_rank, idx=None):
        self.w, self.torch, self.pa = w, torch, pa
        idx = np.arange(len(w["x0"])) if idx is None else np.asarray(idx)
        self.idx, self.B = idx, len(idx)
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a[idx])).to(dev)
        self.x0, self.u0 = to(w["x0"]), to(w["u0"])
        self.u = torch.empty_like(self.u0)
        self.y0 = to(w["y0"]) if w["y0"] is not None else None
        self.y = torch.empty_like(self.y0) if self.y0 is not None else None
        self.stats_t = torch.zeros(max(self.B, 1) * pa.STATS_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        self.batch = pa.PantrBatch(w["problem"], w["horizon"], max(self.B, 1), device=local_rank, opts=w["opts"])
        self.launched = False
        self.last_dev = None  # max |u - recorded closed loop| measured by the last check()

    def launch(self, stream=None):
        if self.B == 0:
            return
        torch = self.torch
        s = stream if stream is not None else torch.cuda.current_stream()
        with torch.cuda.stream(s):
            self.u.copy_(self.u0)
            if self.y is not None:
                self.y.copy_(self.y0)
            self.batch.solve_device(self.B, self.x0.data_ptr(), self.u.data_ptr(), self.stats_t.data_ptr(),
                                    s.cuda_stream, self.y.data_ptr() if self.y is not None else None)
        self.launched = True

    def kernel_ms(self):
        return self.batch.last_kernel_ms() if self.B else 0.0

    def kernel_name(self):
        """The device kernel this batch's solve launches (the library decides: method, speculation, batch size)."""
        return self.batch.kernel_name(max(self.B, 1))

    def stats(self):
        raw = self.stats_t.cpu().numpy().tobytes()
        return np.frombuffer(raw, dtype=self.pa.STATS_DTYPE)[:self.B]

    def solutions(self):
        return self.u.cpu().numpy()

    def check(self, what):
        """Every instance converged, and the replay reproduces the recorded closed loop bit for bit."""
        self.last_dev = None
        if self.B == 0:
            return 0.0
        st = self.stats()
        bad = int(np.sum(st["status"] != 1))
        if bad:
            raise SystemExit(f"bench.py: {bad} of {self.B} instances of '{what}' did not converge - refusing to report")
        dev = 0.0
        if self.w["sol"] is not None:  # rows beyond the recorded ones (the cold twins) have no reference
            have = self.idx < len(self.w["sol"])
            if np.any(have):
                dev = float(np.max(np.abs(self.solutions()[have] - self.w["sol"][self.idx[have]])))
                self.last_dev = dev
                if dev != 0.0:
                    raise SystemExit(f"bench.py: '{what}' replay deviates from the recorded closed loop by "
                                     f"{dev:.3e} - refusing to report")
        return dev

    def alg_bytes(self):
        b = self.batch
        return int(self.pa.algorithmic_bytes(self.stats(), self.w["horizon"], b.nx, b.nu, b.m))

    def close(self):
        self.batch.close()


def roofline_obj(alg_bytes, kernel_ms, kernel, peak_measured=None, traffic=None, traffic_source=None):
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kernel, "kernel_ms": kernel_ms,
         "algorithmic_bytes_per_launch": int(alg_bytes)}
    if peak_measured:
        r["peak_measured"] = peak_measured  # best 16-B-per-lane device copy of this library on this device, GB/s
        r["frac_of_measured_peak"] = achieved / peak_measured
        r["peak_measured_guide"] = HBM_COPY_GUIDE_GBS
    if traffic_source is not None:
        r["traffic_source"] = traffic_source
    return r


# ---- CPU legs (the oracle: checker and CPU baseline, never on the product path) --------------------
def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    ob.lib()  # built (for this host) and loaded before any timer starts
    return ob


def cpu_batch_baseline(w, cores, sample=None):
    """The CPU oracle (kind "port") on `sample` instances of workload w (default: all), one instance
    per thread over `cores` host threads."""
    ob = _oracle()
    idx = np.arange(len(w["x0"])) if sample is None else np.asarray(sample)

    def once():
        if w["problem"] == "hanging_chain":
            return ob.solve_batch(w["horizon"], w["x0"][idx], w["u0"][idx], ob.dodo_params(), nthreads=cores)[1]
        from concurrent.futures import ThreadPoolExecutor
        prm = ob.quad_dodo_params(w["problem"])
        # longest first (cold starts), so the pool does not end on a straggler it could have started early
        order = sorted(idx, key=lambda i: bool(w["warm"][i]))
        with ThreadPoolExecutor(cores) as ex:
            return list(ex.map(lambda i: ob.quad_solve(w["problem"], w["horizon"], w["x0"][i], w["u0"][i],
                                                       w["y0"][i], prm)[2], order))

    # a batch that takes milliseconds is repeated until half a second has passed (at most 64 times)
    reps, t = 0, time.perf_counter()
    while True:
        stats = once()
        reps += 1
        dt = time.perf_counter() - t
        if dt >= 0.5 or reps == 64:
            break
    ok = sum(s["status"] == 1 for s in stats)
    el = [s["elapsed_s"] for s in stats]
    times = "once" if reps == 1 else f"{reps} times"
    return {"value": reps * len(idx) / dt, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"{len(idx)} of the {len(w['x0'])} instances of this workload {times} ({ok} converged) over {cores} "
                      f"host threads, {dt:.2f} s wall; single-thread solve times: sum {sum(el):.2f} s, max {max(el):.3f} s"}


def cpu_serial_closed_loop():
    """SURVEY.md 8(d) CPU plan (i): the serial closed loop on ONE thread, hanging_chain N=60, 60 timed
    steps, per-step solver time as stats.elapsed_time (mpc-driver.cpp:199-201), next to BASELINE.md."""
    ob = _oracle()
    out = {"threads": 1, "kind": "port", "problem": "hanging_chain", "horizon": HORIZON, "num_sim": 60}
    for warm in (False, True):
        _, _, _, stats = ob.mpc(HORIZON, 60, warm)
        t = np.array([s["elapsed_s"] for s in stats])
        key = "warm" if warm else "cold"
        out[key] = {"avg_s": float(t.mean()), "median_s": float(np.median(t)), "p5_s": float(np.percentile(t, 5)),
                    "p95_s": float(np.percentile(t, 95)), "steps_per_s": float(len(t) / t.sum()),
                    "converged": int(sum(s["status"] == 1 for s in stats)),
                    "published_avg_s": PUBLISHED_S_PER_STEP[("hanging_chain", 60)][key]}
    out["note"] = ("published_avg_s = alpaqa PANTR on CasADi-generated code, unstated CPU (BASELINE.md 1.1); this record = "
                   "the CPU oracle (parity unpinned), which loops a stage function and replays a K/M tape")
    return out


# ---- the other BASELINE.json configurations (N = 1, rank 0) -----------------------------------------
def plan_streams(residents, torch, max_streams=4):
    """[(resident, stream)] in launch order for batches that run concurrently.  At most four streams: the runtime
    feeds the GPU through four hardware queues per process, and launches on streams that share a queue run one after
    the other - with one stream for each of the six horizons of the sweep, the pass ended with a short launch waiting
    behind the longest one (45.3 ms where the longest launch takes 40.7).  Longest first (by instances x horizon
    squared); a fifth and sixth batch go behind the streams with the least work."""
    order = sorted(range(len(residents)), key=lambda i: -residents[i].B * float(residents[i].w["horizon"]) ** 2)
    streams = [torch.cuda.Stream() for _ in order[:max_streams]]
    load = [0.0] * len(streams)
    plan = []
    for i in order:
        j = int(np.argmin(load))
        load[j] += residents[i].B * float(residents[i].w["horizon"]) ** 2
        plan.append((residents[i], streams[j]))
    return plan


def time_group(residents, torch, rounds=3):
    """Launch all residents concurrently (up to four HIP streams), `rounds` times; wall seconds per round
    (best), and the per-handle kernel durations of the last round."""
    plan = plan_streams(residents, torch)
    best, first = None, None
    for i in range(rounds + 1):          # first round untimed
        torch.cuda.synchronize()
        t = time.perf_counter()
        for r, s in plan:
            r.launch(s)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        if i == 0:
            first = dt
            continue
        best = dt if best is None else min(best, dt)
        if first > 1.0:                  # a pass of seconds (quadcopter cold starts): one timed round
            break
    return best, [r.kernel_ms() for r in residents]


def extra_configs(torch, pa, dev, local_rank, cores, peak_measured, with_cpu):
    from pantr_amd import workloads
    out = {}

    def one(name, ws, description, cpu_sample=None, traffic_name=None, with_cpu=with_cpu):
        rs = [Resident(w, torch, pa, dev, local_rank) for w in ws]
        wall, kms = time_group(rs, torch)
        for r in rs:
            r.check(name)
        total = sum(r.B for r in rs)
        dom = int(np.argmax(kms))                      # the launch that decides the wall time
        traffic, traffic_src = measured_workload_traffic(traffic_name, total) if traffic_name else (None, None)
        cfg = {"workload": description, "instances": total, "launches_per_pass": len(rs),
               "value": total / wall, "unit": "steps/s", "ms_per_pass": wall * 1e3,
               "roofline": roofline_obj(rs[dom].alg_bytes(), kms[dom], rs[dom].kernel_name(), peak_measured, traffic, traffic_src),
               "all_launches": {"algorithmic_bytes": int(sum(r.alg_bytes() for r in rs)),
                                "achieved_GBps": sum(r.alg_bytes() for r in rs) / wall / 1e9,
                                "kernel_ms": kms}}
        for w in ws:
            pub = PUBLISHED_S_PER_STEP.get((w["problem"], w["horizon"]))
            if pub and len(ws) == 1:
                cfg["published_cpu_1thread_s_per_step"] = pub
        if with_cpu:
            legs = [cpu_batch_baseline(w, cores, cpu_sample) for w in ws]
            ns = [len(w["x0"]) if cpu_sample is None else len(cpu_sample) for w in ws]
            # the groups one after the other: time of one pass = sum of (instances / measured rate) over the groups
            cfg["cpu_baseline"] = {"value": sum(ns) / sum(n / l["value"] for n, l in zip(ns, legs)), "unit": "steps/s",
                                   "cores": cores, "kind": "port", "sample": "; ".join(l["sample"] for l in legs)}
        for r in rs:
            r.close()
        out[name] = cfg

    # configs[3]: six horizons x {cold, warm} x 60 steps = 720 instances, one launch per horizon, concurrent
    one("sweep", [workloads.closed_loop_batch("hanging_chain", N, device=local_rank) for N in workloads.HORIZONS],
        "BASELINE.json configs[3]: hanging_chain horizons 10..60 x {warm, cold} x 60 MPC time steps = 720 replay "
        "instances, one launch per horizon, longest first, on four concurrent HIP streams (dodo.py:69-95)")
    # configs[2]
    one("realistic_quadcopter", [workloads.closed_loop_batch("realistic_quadcopter", 60, device=local_rank)],
        "BASELINE.json configs[2]: realistic_quadcopter N=60 (n=240, m=244), all 60 warm + 60 cold time steps as one "
        "batch of 120 ALM solves (dodo.py:96-105)", traffic_name="realistic_quadcopter")
    # configs[0]: the reference's CPU-runnable case, here as the replay of its one warm run
    one("quadcopter_N12", [workloads.closed_loop_batch("quadcopter", 12, starts=(True,), device=local_rank)],
        "BASELINE.json configs[0]: quadcopter N=12 (n=48, m=52), the 60 time steps of one warm-started MPC run "
        "(num_sim=60 warm=true num_dist=0 solver.max_iter=250) as one batch", traffic_name="quadcopter_N12")
    # configs[4] on ONE GPU: all 1024
    one("stress_N120", [workloads.stress_batch(1024, device=local_rank)],
        "BASELINE.json configs[4] on one GPU: hanging_chain N=120 (n=360), 1024 synthetic cold instances in one launch",
        cpu_sample=np.arange(0, 1024, 8), traffic_name="stress")
    # the memory-bound regime of the headline kernel family: the bench batch tiled eight times into ONE launch
    hw = workloads.headline_batch(device=local_rank, num_sim=300, horizon=HORIZON)
    tiled = dict(hw, x0=np.tile(hw["x0"], (8, 1)), u0=np.tile(hw["u0"], (8, 1)), sol=None, warm=np.tile(hw["warm"], 8))
    # A batch that is solved again and again: its instances are started longest first by the time each took in the
    # pass before (launch_order=auto, an option: the default starts them by index - that figure is attached as well).
    by_index = Resident(tiled, torch, pa, dev, local_rank)
    wall_idx, kms_idx = time_group([by_index], torch)
    by_index.check("tile8 by index")
    by_index.close()
    one("tile8", [dict(tiled, opts=tuple(tiled["opts"]) + ("launch_order=auto",))],
        "the 600-instance bench batch tiled 8 times: 4800 hanging_chain N=60 instances in one launch (one "
        "wavefront per instance, six resident per compute unit (LDS)), option launch_order=auto: started longest first by "
        "the time each took in the pass before: where the kernel is bound by the memory system", traffic_name="tile8",
        with_cpu=False)
    out["tile8"]["started_by_index"] = {"value": by_index.B / wall_idx, "unit": "steps/s", "ms_per_pass": wall_idx * 1e3,
                                        "kernel_ms": kms_idx[0], "note": "the same launch with the default launch_order=natural"}
    return out


# ---- strong scaling over ranks: one batch dealt longest-first --------------------------------------
def strong_mode(args, torch, pa, dev, local_rank, rank, world, distributed, rehearsal, emit=True):
    """One batch dealt to the ranks (BASELINE.json configs[3] | configs[4]); rank 0 prints the record (emit) or
    returns it (the multi-rank default run attaches it to the headline line as `strong`)."""
    from pantr_amd import sharding, workloads
    import torch.distributed as dist
    if args.workload == "sweep":
        ws = [workloads.closed_loop_batch("hanging_chain", N, device=local_rank) for N in workloads.HORIZONS]
        desc = "BASELINE.json configs[3]: 720 hanging_chain replay instances (horizons 10..60 x warm/cold x 60 steps)"
    else:
        ws = [workloads.stress_batch(1024, device=local_rank)]
        desc = "BASELINE.json configs[4]: hanging_chain N=120, 1024 synthetic cold instances"
    # global instance list: (workload k, row i); the same deterministic partition on every rank
    where = [(k, i) for k, w in enumerate(ws) for i in range(len(w["x0"]))]
    # what an instance costs: the Hessian-vector products its own recorded solve needed, times the horizon (a
    # product sweeps the horizon twice), where the replay batch holds that record; N^2.7 otherwise
    costs = [float(ws[k]["cost"][i]) if ws[k].get("cost") is not None
             else workloads.expected_cost(ws[k]["horizon"], bool(ws[k]["warm"][i])) for k, i in where]
    cost_source = ("recorded n_hess_psi_prod x horizon of every instance" if all(w.get("cost") is not None for w in ws)
                   else "horizon^2.7 x (cold: 2.5) - this batch has no recorded solves")
    mine = sharding.shard_indices(len(where), rank, world, costs)
    rs = []
    for k, w in enumerate(ws):
        rows = [where[g][1] for g in mine if where[g][0] == k]
        rs.append(Resident(w, torch, pa, dev, local_rank, rows))
    plan = plan_streams(rs, torch)

    def one_pass():
        for r, s in plan:
            r.launch(s)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        one_pass()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
        torch.cuda.synchronize()
    barrier()
    cdev = None if rehearsal else dev
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0, cdev)
    # a failure on one rank ends all ranks here, before the next collective
    sharding.raise_together(next((m for m in (sharding.local_failure(r.check, args.workload) for r in rs) if m), None), cdev)
    # gather every rank's solutions (first nu inputs of each instance: what the plant receives) and check
    # them on rank 0 against the unsharded solve of the same batch
    nu = rs[0].batch.nu
    local = np.vstack([r.solutions()[:, :nu] for r in rs if r.B]) if len(mine) else np.zeros((0, nu))
    local_idx = np.array([g for k in range(len(ws)) for g in mine if where[g][0] == k], dtype=np.int64)
    gathered = sharding.gather_rows(local, local_idx, len(where), None if rehearsal else dev)
    kms = [r.kernel_ms() for r in rs]
    alg = sum(r.alg_bytes() for r in rs if r.B)
    # what bounds a pass: the slowest instance's in-kernel time (over all ranks), and how full this rank's GPU is
    slowest_local = max((float(r.stats()["elapsed_s"].max()) for r in rs if r.B), default=0.0)
    slowest = sharding.max_over_ranks(slowest_local, cdev)
    cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
    failure, same = None, None
    if rank == 0:
        full = [Resident(w, torch, pa, dev, local_rank) for w in ws]
        for r in full:
            r.launch()
        torch.cuda.synchronize()
        ref = np.vstack([r.solutions()[:, :nu] for r in full])
        same = bool(np.array_equal(ref, gathered))
        for r in full:
            failure = failure or sharding.local_failure(r.check, args.workload + " (unsharded)")
            r.close()
        if not same:
            failure = failure or "bench.py: the sharded solutions differ from the unsharded solve"
    sharding.raise_together(failure, cdev)  # every rank: rank 0's verdict on the gathered result
    record = None
    if rank == 0:
        total = len(where) * args.steps
        dom = int(np.argmax(kms))
        record = {
            "metric": f"MPC-steps solved/sec, {args.workload} batch sharded over {world} MI355X", "value": total / elapsed,
            "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "instances": len(where), "instances_rank0": int(len(mine)),
                       "partition": "longest-processing-time-first (pantr_amd/sharding.py); cost = " + cost_source,
                       "gathered_equals_unsharded": same, "rehearsal_shared_gpu": rehearsal},
            "bound": {"slowest_instance_in_kernel_ms": slowest * 1e3, "ms_per_pass": elapsed / args.steps * 1e3,
                      "instances_per_cu_rank0": len(mine) / cus, "compute_units": cus,
                      "note": "a pass cannot be shorter than its slowest instance: sharding shortens a pass only while "
                              "the GPU is full (many instances per CU)"},
            "roofline": dict(roofline_obj(rs[dom].alg_bytes(), kms[dom], rs[dom].kernel_name()),
                             rank0_all_launches_algorithmic_bytes=int(alg), rank0_kernel_ms=kms),
        }
        if emit:
            print(json.dumps(record), flush=True)
    for r in rs:
        r.close()
    return record


def spawn_ranks(n):
    """`bench.py --gpus N` started bare (no WORLD_SIZE in the environment): start N fresh rank processes - before
    this process has made any torch.cuda / HIP call; it never makes one - with the environment torch.distributed.run
    would give them, relay rank 0's line (its stdout is this process's stdout), and leave with a failure if any
    rank failed.  Children are ended by their exact PIDs if one of them fails while the others still run."""
    import subprocess
    import tempfile
    # rendezvous through a file of this launch's own (PANTR_BENCH_STORE; main() hands it to init_process_group): a port
    # picked by bind-then-close could be taken by another process before the ranks meet
    store = os.path.join(tempfile.mkdtemp(prefix="pantr_bench_"), "rendezvous")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   PANTR_BENCH_STORE="file://" + store,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.pop("MASTER_PORT", None)
        out = None if r == 0 else subprocess.DEVNULL  # rank 0 prints the JSON line
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            try:
                code = p.wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                continue
            pending.remove(p)
            if code != 0:
                rc = rc or code
        if rc and pending:  # a rank failed: its peers would wait for it in the next collective
            time.sleep(5.0)  # (sharding.raise_together normally ends them by themselves)
            for p in pending:
                if p.poll() is None:
                    p.terminate()
            for p in pending:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
    import shutil
    shutil.rmtree(os.path.dirname(store), ignore_errors=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the headline workload (no per-config records, no CPU legs): profiling runs")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent passes in flight per GPU, each on its own HIP stream and solver handle "
                         "(default 1: one pass after the other, the configuration the roofline line describes)")
    ap.add_argument("--workload", choices=("headline", "sweep", "stress"), default="headline")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare: this process becomes the launcher (no GPU call before or after this point)
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import pantr_amd as pa
    from pantr_amd import sharding, workloads

    if not pa.is_product_build():
        raise SystemExit(f"bench.py: {pa.LIB_PATH} is not the product build ({pa.version()})")
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
    dev = torch.device("cuda", local_rank)
    torch.zeros(1, device=dev)  # HIP runtime up before the first call into the library
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl"); used for barriers + one MAX reduction (+ the result gather in strong mode) only
        store = os.environ.get("PANTR_BENCH_STORE")  # set by spawn_ranks; under torch.distributed.run: env:// as usual
        if store:
            dist.init_process_group(backend="gloo" if rehearsal else "nccl", init_method=store, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="gloo" if rehearsal else "nccl")

    if args.scaling == "strong" or args.workload != "headline":
        if args.workload == "headline":
            raise SystemExit("--scaling strong needs --workload sweep|stress (the headline batch does not shard usefully)")
        strong_mode(args, torch, pa, dev, local_rank, rank, world, distributed, rehearsal)
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- the headline workload, produced by the product's own closed loop on this GPU ----
    fixture_dev, source = None, "closed loop run on this GPU before the timed region (pantr_batch_mpc_record)"
    if args.headline_only and os.path.exists(FIXTURE):
        # profiling runs: the same closed loop as recorded in tests/golden (committed data), so that the
        # kernel trace holds nothing but the timed 600-instance launches
        d = np.load(FIXTURE)
        w = dict(problem="hanging_chain", horizon=HORIZON, x0=np.vstack([d["x0"], d["x0"]]),
                 u0=np.vstack([d["guess"], np.zeros_like(d["guess"])]), y0=None, sol=d["sol"],
                 warm=np.r_[np.ones(len(d["x0"]), bool), np.zeros(len(d["x0"]), bool)], opts=tuple(pa.DODO_PANTR_OPTS))
        source = "tests/golden/mpc_hanging_chain_N60_warm300.npz (--headline-only: no other launches in the trace)"
    else:
        w = workloads.headline_batch(device=local_rank, num_sim=300, horizon=HORIZON)
        if os.path.exists(FIXTURE):  # committed data: the same closed loop as recorded by the CPU oracle
            fixture_dev = float(np.max(np.abs(np.load(FIXTURE)["sol"] - w["sol"])))
            if fixture_dev != 0.0:
                raise SystemExit(f"bench.py: the device closed loop deviates from tests/golden by {fixture_dev:.3e}")
    B, n = w["u0"].shape
    S = max(1, args.streams)
    if S > 1:  # several passes in flight fill the SIMDs by themselves: one wavefront per instance
        w = dict(w, opts=tuple(w["opts"]) + ("speculation=off",))
    rs = [Resident(w, torch, pa, dev, local_rank) for _ in range(S)]
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(dev) for _ in range(S - 1)]

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(max(args.warmup, S if S > 1 else 0)):
        rs[i % S].launch(streams[i % S])
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        rs[i % S].launch(streams[i % S])
        if S == 1:
            kernel_ms.append(rs[0].kernel_ms())  # HIP events on the launch stream (synchronises)
    barrier()
    elapsed = time.perf_counter() - t0
    if S > 1:  # the last launch of every stream (durations overlap: they are latencies, not shares)
        kernel_ms = [r.kernel_ms() for r in rs[:min(S, args.steps)]]
    elapsed = sharding.max_over_ranks(elapsed, None if rehearsal else dev)  # MAX over ranks (identity at N = 1)
    # a failure on one rank ends all ranks together (none is left waiting in the closing barrier)
    sharding.raise_together(next((m for m in (sharding.local_failure(r.check, "headline") for r in rs if r.launched) if m),
                                 None), None if rehearsal else dev)

    if rank == 0:
        total_steps = world * B * args.steps
        stats = rs[0].stats()
        alg_bytes = rs[0].alg_bytes()
        k_ms = float(np.mean(kernel_ms))
        value = total_steps / elapsed
        extras = world == 1 and not args.headline_only
        peak_measured = pa.copy_bandwidth(local_rank) if extras else None
        traffic, traffic_src = measured_traffic(B)
        line = {
            "metric": METRIC, "value": value, "unit": "steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "hanging_chain horizon=60 replay batch: 300 MPC time steps of one closed "
                            "loop (num_sim=300 num_dist=5 u_dist=-1,1,1), each solved warm and cold = "
                            "600 PANTR solves per GPU per pass, tolerance 1e-8, dodo.py PANTR options",
                "workload_source": source,
                "instances_per_gpu": B, "horizon": HORIZON, "n": n, "streams": S,
                "converged": int(np.sum(stats["status"] == 1)),
                # measured by Resident.check() on this run's solutions (a nonzero value refuses to report)
                "max_abs_dev_from_recorded_closed_loop": rs[0].last_dev, "max_abs_dev_from_tests_golden": fixture_dev,
                "scaling_note": "weak = every rank solves its own 600-instance batch (replicas, no exchange step); "
                                "`--workload sweep|stress --scaling strong` shards one batch over the ranks",
                "published_reference": "BASELINE.md 1.1: 4.21 steps/s (alpaqa PANTR cold+warm, unstated CPU, 1 thread) - "
                                       "another machine and another code base: not a baseline for this metric",
                "parity": "unpinned (oracle = this project's restatement of un-vendored alpaqa)",
            },
            "roofline": roofline_obj(alg_bytes, k_ms, rs[0].kernel_name(), peak_measured, traffic,
                                     traffic_src),
        }
        if extras:
            affinity = len(os.sched_getaffinity(0))
            cores = min(16, affinity)  # the GPU box's CPU share for one GPU
            # warm and cold separately: the two halves of the batch as launches of their own
            for key, rows in (("warm", np.arange(B // 2)), ("cold", np.arange(B // 2, B))):
                r = Resident(w, torch, pa, dev, local_rank, rows)
                wall, kms = time_group([r], torch)
                r.check(key)
                line[key] = {"value": r.B / wall, "unit": "steps/s", "instances": r.B, "ms_per_pass": wall * 1e3,
                             "mean_iterations": float(r.stats()["iterations"].mean()),
                             "roofline": roofline_obj(r.alg_bytes(), kms[0], r.kernel_name(),
                                                      peak_measured)}
                r.close()
            # throughput mode: four independent passes in flight, every stream issuing its next pass as soon
            # as its previous one is queued (one pass fills 600 of the chip's wavefront slots and ends on
            # its slowest instance; overlapping passes fill the tail)
            # (speculation off: its second wavefront per instance uses SIMDs that four concurrent passes fill anyway)
            w_thr = dict(w, opts=tuple(w["opts"]) + ("speculation=off",))
            four = [Resident(w_thr, torch, pa, dev, local_rank) for _ in range(4)]
            four_streams = [torch.cuda.Stream(dev) for _ in four]
            rounds = 3
            for r, st4 in zip(four, four_streams):
                r.launch(st4)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            for _ in range(rounds):
                for r, st4 in zip(four, four_streams):
                    r.launch(st4)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t4
            for r in four:
                r.check("throughput mode")
            line["throughput_mode"] = {"passes_in_flight": 4, "passes": 4 * rounds, "value": 4 * rounds * B / wall,
                                       "unit": "steps/s", "ms_total": wall * 1e3,
                                       "last_kernel_ms": [r.kernel_ms() for r in four],
                                       "kernel": four[0].kernel_name(),
                                       "note": "same batch, four independent solver handles on four HIP streams, each "
                                               "running its passes back to back (speculation=off: one wavefront per "
                                               "instance); the headline stays one pass at a time"}
            for r in four:
                r.close()
            with_cpu = not args.no_cpu_baseline
            if with_cpu:
                line["cpu_baseline"] = cpu_batch_baseline(w, cores)
                line["cpu_baseline"]["host"] = {"os_cpu_count": os.cpu_count(), "affinity": affinity,
                                                "threads_used": cores}
                if affinity > cores:
                    # more of the host than the one-GPU share: every hardware thread this process may run on, half
                    # and a quarter of them (the oracle's throughput does not grow monotonically with the thread
                    # count on a two-socket host) - the best of the three and all three values
                    legs = [cpu_batch_baseline(w, t) for t in sorted({affinity, max(cores, affinity // 2),
                                                                      max(cores, affinity // 4)})]
                    best = max(legs, key=lambda l: l["value"])
                    line["cpu_baseline"]["all_cores"] = dict(best, tried={str(l["cores"]): l["value"] for l in legs})
                line["cpu_serial_closed_loop"] = cpu_serial_closed_loop()
            line["configs"] = extra_configs(torch, pa, dev, local_rank, cores, peak_measured, with_cpu)
    for r in rs:
        r.close()
    # Several ranks, default flags (what the driver's scaling run starts): the weak headline above is N replicas of
    # one batch and scales by construction; what shows whether sharding helps are the two batches that DO shard -
    # BASELINE.json configs[3] (the horizon sweep) and configs[4] (the N = 120 stress batch) dealt to the ranks, a
    # few passes each.  Attached as `strong`; the headline value stays the weak one.
    if distributed and not args.headline_only and S == 1:
        import copy
        strong = {}
        for wl in ("sweep", "stress"):
            a2 = copy.copy(args)
            a2.workload, a2.scaling, a2.steps, a2.warmup = wl, "strong", 3, 1
            rec = strong_mode(a2, torch, pa, dev, local_rank, rank, world, distributed, rehearsal, emit=False)
            if rank == 0:
                strong[wl] = {k: rec[k] for k in ("value", "unit", "ms_per_step", "steps", "scaling", "bound")}
                strong[wl]["instances"] = rec["config"]["instances"]
                strong[wl]["gathered_equals_unsharded"] = rec["config"]["gathered_equals_unsharded"]
                strong[wl]["roofline"] = {k: rec["roofline"].get(k) for k in ("kernel", "frac", "achieved", "unit")}
        if rank == 0:
            line["strong"] = strong
    if rank == 0:
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
