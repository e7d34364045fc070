#!/usr/bin/env python3
"""The bench workload (600 instances in one launch) against the oracle, instance by instance, and
twice in a row (determinism under full occupancy).  Test infrastructure: uses the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pantr_amd as pa
import oracle_binding as ob

d = np.load(os.path.join(ROOT, "tests/golden/mpc_hanging_chain_N60_warm300.npz"))
x0 = np.vstack([d["x0"], d["x0"]]); u0 = np.vstack([d["guess"], np.zeros_like(d["guess"])])
b = pa.PantrBatch("hanging_chain", 60, 600, opts=pa.DODO_PANTR_OPTS)
runs = [b.solve(x0, u0) for _ in range(3)]
b.close()
for r in range(1, 3):
    same = np.array_equal(runs[r][0], runs[0][0]) and np.array_equal(runs[r][1]["n_hess_psi_prod"], runs[0][1]["n_hess_psi_prod"])
    print(f"run {r} identical to run 0: {same}")
u_o, st_o = ob.solve_batch(60, x0, u0, ob.dodo_params(), nthreads=16)
hv_o = np.array([s["n_hess_psi_prod"] for s in st_o]); it_o = np.array([s["iterations"] for s in st_o])
u, st = runs[0]
bad = np.flatnonzero(np.any(u != u_o, axis=1) | (st["n_hess_psi_prod"] != hv_o) | (st["iterations"] != it_o))
print(f"instances differing from the oracle: {len(bad)} of 600", bad[:10])
for i in bad[:5]:
    print("  ", i, "gpu iters/hv", st["iterations"][i], st["n_hess_psi_prod"][i], "oracle", it_o[i], hv_o[i], "max|du|", np.abs(u[i] - u_o[i]).max())
sys.exit(1 if len(bad) else 0)
