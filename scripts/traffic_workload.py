#!/usr/bin/env python3
"""One memory-bound workload of the hanging chain, launched a few times and nothing else: the program the
rocprofv3 passes of scripts/traffic_profile.sh run (kernel trace, FETCH_SIZE, WRITE_SIZE in separate passes).

    traffic_workload.py stress [opt=value ...]   BASELINE.json configs[4]: N = 120, 1024 synthetic cold instances
    traffic_workload.py tile8  [opt=value ...]   the 600-instance bench batch tiled 8 times (4800 instances, N = 60)
    traffic_workload.py realistic_quadcopter     BASELINE.json configs[2]: N = 60, the 60 warm + 60 cold steps as one batch of 120
    traffic_workload.py quadcopter_N12           BASELINE.json configs[0]: N = 12, the 60 steps of one warm run as one batch

Extra arguments are solver options (e.g. residency=3).  Prints one JSON line with the launch's kernel time (HIP
events), steps/s, the algorithmic bytes of SURVEY.md 8(d) from the instances' own counters, and the kernel name."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa
from pantr_amd import workloads

what = sys.argv[1] if len(sys.argv) > 1 else "stress"
extra = tuple(a for a in sys.argv[2:] if "=" in a and not a.startswith("order="))
# order=prev: from the second launch on, workgroups start the instances longest first by the products each needed in
# the launch before (pantr_batch_set_launch_order; results do not depend on the order)
order_prev = "order=prev" in sys.argv[2:]
launches = 3
y0 = None
if what == "stress":
    w = workloads.stress_batch(1024)
    X0, U0, N = w["x0"], w["u0"], 120
    opts = tuple(w["opts"]) + extra
elif what in ("realistic_quadcopter", "quadcopter_N12"):
    w = (workloads.closed_loop_batch("realistic_quadcopter", 60) if what == "realistic_quadcopter"
         else workloads.closed_loop_batch("quadcopter", 12, starts=(True,)))
    X0, U0, y0, N = w["x0"], w["u0"], w["y0"], w["horizon"]
    opts = tuple(w["opts"]) + extra
else:
    R = int(what[4:]) if what.startswith("tile") else 8
    d = np.load(os.path.join(ROOT, "tests/golden/mpc_hanging_chain_N60_warm300.npz"))
    x0 = np.vstack([d["x0"], d["x0"]]); u0 = np.vstack([d["guess"], np.zeros_like(d["guess"])])
    X0, U0, N = np.tile(x0, (R, 1)), np.tile(u0, (R, 1)), 60
    # (a batch solved again and again: started longest first by the pass before, as bench.py's tile8 record - unless
    # the command line names an order)
    opts = tuple(pa.DODO_PANTR_OPTS) + ("alm.max_time=5s",) + (() if any(a.startswith("launch_order=") for a in extra) else ("launch_order=auto",)) + extra
B = len(X0)
problem = w["problem"] if y0 is not None else "hanging_chain"
batch = pa.PantrBatch(problem, N, B, opts=opts)
ms, sources = [], []
for _ in range(launches):
    if y0 is not None:
        u, y, st = batch.solve_al(X0, U0, y0)
    else:
        u, st = batch.solve(X0, U0)
    ms.append(batch.last_kernel_ms())
    sources.append(batch.last_launch_order()[0])
    if order_prev:
        from pantr_amd import sharding
        batch.set_launch_order(sharding.launch_order(st["n_hess_psi_prod"].astype(float)))
if y0 is not None:
    alg = int(pa.algorithmic_bytes(st, N, 9, 4, m=batch.m))
else:
    alg = int(pa.algorithmic_bytes(st, N, 57, 3))
k = min(ms[1:])
print(json.dumps({"workload": what, "options": list(extra) + (["order=prev"] if order_prev else []), "instances": B, "horizon": N, "launches": launches,
                  "kernel": batch.kernel_name(B), "kernel_ms": ms, "launch_order": sources, "steps_per_s": B / (k * 1e-3),
                  "converged": int((st["status"] == 1).sum()), "algorithmic_bytes_per_launch": alg,
                  "algorithmic_GBps": alg / (k * 1e-3) / 1e9, "frac_of_8TBps": alg / (k * 1e-3) / 8e12,
                  "n_hess_psi_prod_total": int(st["n_hess_psi_prod"].sum()), "source_hash": pa.source_hash()}))
batch.close()
