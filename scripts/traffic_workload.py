#!/usr/bin/env python3
"""One memory-bound workload of the hanging chain, launched a few times and nothing else: the program the
rocprofv3 passes of scripts/traffic_profile.sh run (kernel trace, FETCH_SIZE, WRITE_SIZE in separate passes).

    traffic_workload.py stress [opt=value ...]   BASELINE.json configs[4]: N = 120, 1024 synthetic cold instances
    traffic_workload.py tile8  [opt=value ...]   the 600-instance bench batch tiled 8 times (4800 instances, N = 60)

Extra arguments are solver options (e.g. residency=3).  Prints one JSON line with the launch's kernel time (HIP
events), steps/s, the algorithmic bytes of SURVEY.md 8(d) from the instances' own counters, and the kernel name."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa
from pantr_amd import workloads

what = sys.argv[1] if len(sys.argv) > 1 else "stress"
extra = tuple(a for a in sys.argv[2:] if "=" in a)
launches = 3
if what == "stress":
    w = workloads.stress_batch(1024)
    X0, U0, N = w["x0"], w["u0"], 120
    opts = tuple(w["opts"]) + extra
else:
    R = int(what[4:]) if what.startswith("tile") else 8
    d = np.load(os.path.join(ROOT, "tests/golden/mpc_hanging_chain_N60_warm300.npz"))
    x0 = np.vstack([d["x0"], d["x0"]]); u0 = np.vstack([d["guess"], np.zeros_like(d["guess"])])
    X0, U0, N = np.tile(x0, (R, 1)), np.tile(u0, (R, 1)), 60
    opts = tuple(pa.DODO_PANTR_OPTS) + ("alm.max_time=5s",) + extra
B = len(X0)
batch = pa.PantrBatch("hanging_chain", N, B, opts=opts)
ms = []
for _ in range(launches):
    u, st = batch.solve(X0, U0)
    ms.append(batch.last_kernel_ms())
alg = int(pa.algorithmic_bytes(st, N, 57, 3))
k = min(ms[1:])
print(json.dumps({"workload": what, "options": list(extra), "instances": B, "horizon": N, "launches": launches,
                  "kernel": batch.kernel_name(B), "kernel_ms": ms, "steps_per_s": B / (k * 1e-3),
                  "converged": int((st["status"] == 1).sum()), "algorithmic_bytes_per_launch": alg,
                  "algorithmic_GBps": alg / (k * 1e-3) / 1e9, "frac_of_8TBps": alg / (k * 1e-3) / 8e12,
                  "n_hess_psi_prod_total": int(st["n_hess_psi_prod"].sum()), "source_hash": pa.source_hash()}))
batch.close()
