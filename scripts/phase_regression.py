#!/usr/bin/env python3
"""Attributes per-instance solve time to phases by regressing elapsed time on the evaluation
counters of the 600-instance replay workload (development aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa

d = np.load(os.path.join(ROOT, "tests/golden/mpc_hanging_chain_N60_warm300.npz"))
x0 = np.vstack([d["x0"], d["x0"]]); u0 = np.vstack([d["guess"], np.zeros_like(d["guess"])])
B = int(sys.argv[1]) if len(sys.argv) > 1 else len(x0)
x0, u0 = x0[:B], u0[:B]
batch = pa.PantrBatch("hanging_chain", 60, B, opts=pa.DODO_PANTR_OPTS + ("alm.max_time=3s",))
for _ in range(2):
    u, st = batch.solve(x0, u0)
names = ["n_hess_psi_prod", "n_psi", "iterations"]  # the other counters are collinear with these
A = np.column_stack([st[k].astype(float) for k in names] + [np.ones(B)])
el = st["elapsed_s"] * 1e6
coef, *_ = np.linalg.lstsq(A, el, rcond=None)
print(f"B={B} kernel {batch.last_kernel_ms():.1f} ms; mean elapsed {el.mean()/1e3:.2f} ms")
for k, c, col in zip(names + ["const"], coef, A.T):
    print(f"  {k:18s} {c:9.2f} us each  x mean {col.mean():8.1f} = {c*col.mean()/1e3:7.2f} ms ({100*c*col.mean()/el.mean():.0f} %)")
print("  residual rms %.1f us" % np.sqrt(np.mean((A @ coef - el) ** 2)))
