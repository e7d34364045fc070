#!/usr/bin/env python3
"""Times the quadcopter tape product in isolation (development aid): eval entry point with a repeat count."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa
L = pa.lib()
rng = np.random.default_rng(0)
PROBLEM = sys.argv[1] if len(sys.argv) > 1 else "quadcopter"
for N in (12, 16, 20, 60):
    for B in (1, 60):
        b = pa.PantrBatch(PROBLEM, N, B)
        n, m = 4 * N, 4 * N + 4
        x0 = np.tile(b.initial_state(), (B, 1)); u = np.tile(b.initial_guess(), (B, 1)) + 0.1 * rng.standard_normal((B, n))
        y, Sig, v = np.zeros((B, m)), np.full((B, m), 1e4), rng.standard_normal((B, n))
        res = {}
        for reps in (1, 201):
            L.pantr_debug_set_eval_reps(b._h, reps)
            b.eval_hess_psi_prod_al(x0, u, y, Sig, v)
            t = time.perf_counter(); b.eval_hess_psi_prod_al(x0, u, y, Sig, v); res[reps] = time.perf_counter() - t
        L.pantr_debug_set_eval_reps(b._h, 1)
        per = (res[201] - res[1]) / 200
        print(f"N={N} B={B}: {per*1e6:.2f} us per product = {per/N*1e9:.0f} ns per stage")
        b.close()
