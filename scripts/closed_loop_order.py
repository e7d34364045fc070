#!/usr/bin/env python3
"""Does the start order of the solve before help a GENUINE closed loop (not a replayed batch)?  B hanging-chain
controllers (N = 60) in lock-step, each with its own disturbance during the first five steps; every time step is one
launch of B instances whose initial states and warm starts come from the step before.  Wall time of the whole loop
with launch_order=natural and launch_order=auto (ranked by the time each instance took a step earlier), same
disturbances; results must be identical.

    python scripts/closed_loop_order.py [B] [num_sim] > profiles/<tag>_closed_loop_order.txt"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4800
num_sim = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(2023)
u_dist = rng.uniform(-1.0, 1.0, size=(B, 3))
print(f"# {B} hanging_chain N=60 closed loops in lock-step, {num_sim} time steps, warm starts, disturbances U(-1,1)^3 during 5 steps")
print(f"# ({pa.version()}; scripts/closed_loop_order.py)")
ref = None
for rep in range(2):
    for mode in ("natural", "auto"):
        b = pa.PantrBatch("hanging_chain", 60, B, opts=tuple(pa.DODO_PANTR_OPTS) + (f"launch_order={mode}", "alm.max_time=5s"))
        t = time.perf_counter()
        states, inputs, stats = b.mpc(num_sim, True, u_dist)
        wall = time.perf_counter() - t
        conv = int((stats["status"] == 1).sum())
        per_step = stats["elapsed_s"].max(axis=1) * 1e3       # the slowest instance of every step (in-kernel time)
        if ref is None:
            ref = inputs
        print(f"launch_order={mode:8s} wall {wall*1e3:8.1f} ms for {num_sim * B} solves = {num_sim * B / wall:8.0f} steps/s; "
              f"converged {conv}/{num_sim * B}; identical inputs: {np.array_equal(inputs, ref)}; "
              f"mean iterations per step {stats['iterations'].mean(axis=1).round(1).tolist()}")
        # how well does a step's solve time forecast the next step's?  (rank correlation over the instances)
        if mode == "natural" and rep == 0:
            from scipy.stats import spearmanr
            rho = [spearmanr(stats["n_hess_psi_prod"][k], stats["n_hess_psi_prod"][k + 1]).statistic for k in range(num_sim - 1)]
            print(f"  rank correlation of an instance's products at step k and k+1: {np.round(rho, 2).tolist()}")
        b.close()
