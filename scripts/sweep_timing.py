#!/usr/bin/env python3
"""Shader cycles per stage of the three first-order sweeps of the hanging chain on a lone wavefront
(profiling build: PANTR_LIB_PATH=.../libpantr_mi355x_phase.so).  Development aid."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa
if "+phase-timing" not in pa.version():
    raise SystemExit("needs the profiling build (make phase; PANTR_LIB_PATH)")
d = np.load(os.path.join(ROOT, "tests/golden/mpc_hanging_chain_N60_warm300.npz"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x0, u = np.ascontiguousarray(d["x0"][:B]), np.ascontiguousarray(d["sol"][:B])
L = pa.lib()
b = pa.PantrBatch("hanging_chain", 60, B)
reps = 50
L.pantr_debug_set_eval_reps(b._h, reps)
vp = C.c_void_p
sc, g = np.zeros((B, 4)), np.zeros_like(u)
for _ in range(2):
    L.pantr_debug_psi_grad_timed(b._h, B, x0.ctypes.data_as(vp), u.ctypes.data_as(vp), g.ctypes.data_as(vp),
                                 sc.ctypes.data_as(vp))
f, a, r = (np.median(sc[:, i]) / reps / 60 for i in (1, 2, 3))
print(f"B={B}: cycles per stage: forward (stored) {f:.0f}, adjoint {a:.0f}, adjoint + recording {r:.0f}")
