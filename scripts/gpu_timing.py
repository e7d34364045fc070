#!/usr/bin/env python3
"""Quick GPU timing of the replay workload (development aid, not the bench)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa

d = np.load(os.path.join(ROOT, "tests/golden/mpc_hanging_chain_N60_warm300.npz"))
x0, guess, sol = d["x0"], d["guess"], d["sol"]
B = len(x0)
N = 60
for name, xs, us in (("warm300", x0, guess), ("cold300", x0, np.zeros_like(guess)),
                     ("both600", np.vstack([x0, x0]), np.vstack([guess, np.zeros_like(guess)]))):
    batch = pa.PantrBatch("hanging_chain", N, len(xs), opts=pa.DODO_PANTR_OPTS + ("alm.max_time=3s",))
    for rep in range(2):
        t = time.time(); u, st = batch.solve(xs, us); wall = time.time() - t
        ms = batch.last_kernel_ms()
    el = st["elapsed_s"]
    print(f"{name}: kernel {ms:.1f} ms wall {wall*1e3:.1f} ms -> {len(xs)/(ms*1e-3):.1f} steps/s; "
          f"per-instance elapsed mean {el.mean()*1e3:.1f} med {np.median(el)*1e3:.1f} max {el.max()*1e3:.1f} ms; "
          f"conv {np.sum(st['status']==1)}/{len(xs)} iters mean {st['iterations'].mean():.1f} hv mean {st['n_hess_psi_prod'].mean():.1f}")
    if name == "warm300":
        print("  max |u - oracle sol| =", np.abs(u - sol).max(), " hv equal:", np.array_equal(st["n_hess_psi_prod"], d["hv"]))
        print("  per-Hv time (us):", (el.sum() / st["n_hess_psi_prod"].sum()) * 1e6)
        print("  alg bytes/launch: %.3e" % pa.algorithmic_bytes(st, N, 57, 3))
    batch.close()
