#!/usr/bin/env python3
"""Quadcopter-family solve time by phase: one cold closed loop (60 steps) per problem, per-step
elapsed time regressed on the evaluation counters (development aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa

problem = sys.argv[1] if len(sys.argv) > 1 else "quadcopter"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
from pantr_amd import workloads
task = [t for t in workloads.dodo_pantr_tasks(problem) if t["horizon"] == N and not t["warm"]][0]
batch = pa.PantrBatch(problem, N, 1, opts=tuple(o for o in task["argv"][3:] if o.startswith(("solver.", "alm.", "dir.", "accel."))))
u_dist = np.zeros((1, 4))  # dodo.py:97-100: num_dist=0
states, inputs, st = batch.mpc(60, False, u_dist, num_dist=0)
st = st[:, 0]
names = ["n_hess_psi_prod", "n_grad_L", "n_psi_grad_psi", "n_psi"]
A = np.column_stack([st[k].astype(float) for k in names])
el = st["elapsed_s"] * 1e6
coef, *_ = np.linalg.lstsq(A, el, rcond=None)
print(f"{problem} N={N}: mean step {el.mean()/1e3:.2f} ms, {np.sum(st['status']==1)}/60 converged, outer {st['outer_iterations'].mean():.1f}")
for k, c, col in zip(names, coef, A.T):
    print(f"  {k:18s} {c:9.2f} us each  x mean {col.mean():8.1f} = {c*col.mean()/1e3:7.2f} ms ({100*c*col.mean()/el.mean():.0f} %)")
print("  residual rms %.1f us" % np.sqrt(np.mean((A @ coef - el) ** 2)))
