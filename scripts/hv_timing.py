#!/usr/bin/env python3
"""Times the tape-replay Hessian-vector product in isolation (development aid)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa
d = np.load(os.path.join(ROOT, "tests/golden/mpc_hanging_chain_N60_warm300.npz"))
x0 = np.vstack([d["x0"], d["x0"]]); u = np.vstack([d["sol"], d["guess"]])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 600
x0, u = x0[:B], u[:B]
v = np.random.default_rng(0).standard_normal(u.shape)
L = pa.lib()
b = pa.PantrBatch("hanging_chain", 60, B)
depth = int(os.environ.get("PREFETCH_DEPTH", "1"))  # 1 | 2: which form of the product the evaluation kernel runs
L.pantr_debug_set_eval_prefetch_depth(b._h, depth)
print(f"tape rows requested {depth} stage(s) ahead")
res = {}
for reps in (1, 101):
    L.pantr_debug_set_eval_reps(b._h, reps)
    b.eval_hess_psi_prod(x0, u, v)
    t = time.perf_counter(); Hv = b.eval_hess_psi_prod(x0, u, v); res[reps] = time.perf_counter() - t
import ctypes as C
vp = C.c_void_p
sc = np.zeros((B, 4)); Hv = np.zeros_like(u)
x0c, uc, vc = map(np.ascontiguousarray, (x0, u, v))
L.pantr_debug_hess_prod_timed(b._h, B, x0c.ctypes.data_as(vp), uc.ctypes.data_as(vp), vc.ctypes.data_as(vp),
                              Hv.ctypes.data_as(vp), sc.ctypes.data_as(vp))
cyc, ticks = sc[:, 1], sc[:, 2]
print(f"in-kernel: {np.median(cyc)/101/60:.0f} shader cycles per stage (fwd+adj), clock {np.median(cyc/ticks)*100:.0f} MHz, "
      f"{np.median(ticks)/101*10/1e3:.1f} us per product")
if "+phase-timing" in pa.version():
    f = np.median(sc[:, 3]) / 101 / 60
    print(f"   forward sweep {f:.0f} cycles per stage, adjoint sweep {np.median(cyc)/101/60 - f:.0f}")
print(f"B={B}: Hv (tape replay) = {(res[101]-res[1])/100*1e6:.1f} us per product per wave "
      f"({60*(res[101]-res[1])/100/60*1e9/60:.0f} ns/stage)")
