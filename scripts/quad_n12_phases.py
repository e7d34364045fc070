#!/usr/bin/env python3
"""Phases of the slowest instances of BASELINE.json configs[0] (quadcopter N=12, 60 warm steps as one batch); profiling build. Development aid."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa
from pantr_amd import workloads
w = workloads.closed_loop_batch("quadcopter", 12, starts=(True,))
b = pa.PantrBatch("quadcopter", 12, len(w["x0"]), opts=tuple(w["opts"]))
for rep in range(3):
    st = b.solve_al(w["x0"], w["u0"], w["y0"])[-1].ravel()
print(pa.version(), b.kernel_name(), "pass %.3f ms" % b.last_kernel_ms())
for i in np.argsort(-st["elapsed_s"])[:3]:
    s = st[i]; ph = s["phase_s"] * 1e3
    print(f"  inst {i}: {s['elapsed_s']*1e3:.3f} ms outer {s['outer_iterations']} inner {s['iterations']} Hv {s['n_hess_psi_prod']} ({ph[0]:.3f} ms, {ph[0]/s['n_hess_psi_prod']*1e3:.2f} us each) "
          f"psi_grad {s['n_psi_grad_psi']} ({ph[1]:.3f} ms) psi {s['n_psi']} ({ph[2]:.3f} ms) grad_L {s['n_grad_L']} ({ph[3]:.3f} ms) rest {s['elapsed_s']*1e3 - ph.sum():.3f} ms")
b.close()
