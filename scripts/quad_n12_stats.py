#!/usr/bin/env python3
"""Counters of the slowest instances of BASELINE.json configs[0] (quadcopter N=12, 60 warm steps as one batch). Development aid."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa
from pantr_amd import workloads
w = workloads.closed_loop_batch("quadcopter", 12, starts=(True,))
for spec in ("auto", "off"):
    b = pa.PantrBatch("quadcopter", 12, len(w["x0"]), opts=tuple(w["opts"]) + (f"speculation={spec}",))
    for rep in range(3):
        st = b.solve_al(w["x0"], w["u0"], w["y0"])[-1].ravel()
    print(spec, b.kernel_name(), "pass %.3f ms" % b.last_kernel_ms(), "mean solve %.3f ms" % (st["elapsed_s"].mean() * 1e3))
    for i in np.argsort(-st["elapsed_s"])[:6]:
        s = st[i]
        print(f"  inst {i:3d}: {s['elapsed_s']*1e3:6.3f} ms  outer {s['outer_iterations']:2d} inner {s['iterations']:3d} cg {s['cg_iterations']:4d} Hv {s['n_hess_psi_prod']:4d} "
              f"psi {s['n_psi']:3d} psi_grad {s['n_psi_grad_psi']:3d} grad_psi {s['n_grad_psi']:3d} grad_L {s['n_grad_L']:3d} prox {s['n_prox_grad_step']:3d} qub {s['qub_backtracks']} acc {s['accepted']} rej {s['rejected']}")
    print("  totals: outer", st["outer_iterations"].sum(), "inner", st["iterations"].sum(), "Hv", st["n_hess_psi_prod"].sum())
    b.close()
