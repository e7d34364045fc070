#!/usr/bin/env python3
"""One launch of each kernel of rows f1 / f4 on a fixed workload (for `rocprofv3 --kernel-trace --stats`):
quadcopter and realistic_quadcopter N=60: steps 10..59 of a cold closed loop replayed as one batch (PANTR);
hanging_chain N=60, the first 60 warm + 60 cold instances of the bench workload with PANOC+ and structured PANOC+."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa
from pantr_amd import workloads

for problem in ("quadcopter", "realistic_quadcopter"):
    task = [t for t in workloads.dodo_pantr_tasks(problem) if t["horizon"] == 60 and not t["warm"]][0]
    opts = tuple(o for o in task["argv"][3:] if o.startswith(("solver.", "alm.", "dir.", "accel.")))
    b = pa.PantrBatch(problem, 60, 50, opts=opts)
    states, inputs, st = b.mpc(60, False, np.zeros((1, 4)), num_dist=0)  # one cold closed loop (dodo.py:97-100)
    x0 = np.ascontiguousarray(states[10:60, 0])  # its steps 10..59, replayed as one cold batch
    u, y, st = b.solve_al(x0, np.tile(b.initial_guess(), (len(x0), 1)))
    same = np.array_equal(u, inputs[10:60, 0])
    print(f"{problem} N=60: {len(x0)} cold steps of a closed loop in one launch, kernel {b.last_kernel_ms():.1f} ms, converged "
          f"{int((st['status']==1).sum())}, mean {st['elapsed_s'].mean()*1e3:.1f} ms per instance, {st['iterations'].mean():.0f} iterations, "
          f"{st['n_hess_psi_prod'].mean():.0f} Hv, identical to the closed loop: {same}")
    b.close()
d = np.load(os.path.join(ROOT, "tests/golden/mpc_hanging_chain_N60_warm300.npz"))
x0 = np.vstack([d["x0"][:60], d["x0"][:60]]); u0 = np.vstack([d["guess"][:60], np.zeros_like(d["guess"][:60])])
for m in ("panoc", "strucpanoc"):
    b = pa.PantrBatch("hanging_chain", 60, len(x0), opts=pa.DODO_OPTS[m])
    u, st = b.solve(x0, u0)
    print(f"hanging_chain N=60 {m}: {len(x0)} instances, kernel {b.last_kernel_ms():.1f} ms, converged {int((st['status']==1).sum())}, "
          f"mean {st['elapsed_s'].mean()*1e3:.1f} ms per instance, {st['iterations'].mean():.0f} iterations")
    b.close()
