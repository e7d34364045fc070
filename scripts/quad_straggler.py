#!/usr/bin/env python3
"""The instances that decide the realistic_quadcopter N=60 replay batch (BASELINE.json configs[2]):
counters and, with the profiling build, phases.  Development aid."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa
from pantr_amd import workloads

problem = sys.argv[1] if len(sys.argv) > 1 else "realistic_quadcopter"
spec = sys.argv[2] if len(sys.argv) > 2 else "auto"
w = workloads.closed_loop_batch(problem, 60)
B = len(w["x0"])
b = pa.PantrBatch(problem, 60, B, opts=tuple(w["opts"]) + (f"speculation={spec}",))
print(b.kernel_name(), "LDS, workspace per instance:", b.footprint())
for _ in range(2):
    out = b.solve_al(w["x0"], w["u0"], w["y0"])
st = out[-1].ravel()
print(f"{problem} N=60, {B} instances: pass {b.last_kernel_ms():.1f} ms, sum of solves {st['elapsed_s'].sum()*1e3:.0f} ms")
for i in np.argsort(-st["elapsed_s"])[:5]:
    s = st[i]
    line = (f"  inst {i} ({'warm' if w['warm'][i] else 'cold'}): {s['elapsed_s']*1e3:.1f} ms, outer {s['outer_iterations']}, it {s['iterations']}, "
            f"accepted {s['accepted']}, rejected {s['rejected']}, qub {s['qub_backtracks']}, Hv {s['n_hess_psi_prod']}, cg {s['cg_iterations']}, "
            f"psi {s['n_psi']}, psi_grad {s['n_psi_grad_psi']}, grad_L {s['n_grad_L']}")
    if "+phase-timing" in pa.version():
        ph = s["phase_s"] * 1e3
        line += f" | ms: Hv {ph[0]:.1f}, psi_grad(cand) {ph[1]:.1f}, psi(xhat) {ph[2]:.1f}, grad_L+tape {ph[3]:.1f}, rest {s['elapsed_s']*1e3 - ph.sum():.1f}"
    print(line)
b.close()
