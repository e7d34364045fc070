#!/usr/bin/env python3
"""Mean in-kernel solve time of the cold closed loops of the quadcopter family at N = 60 (60 steps each):
the quick A/B number for kernel variants (PANTR_LIB_PATH selects the library)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa
from pantr_amd import workloads
for problem in sys.argv[1:] or ["quadcopter", "realistic_quadcopter"]:
    b = pa.PantrBatch(problem, 60, 1, opts=workloads._quad_solver_opts(problem))
    _, _, st = b.mpc(60, False, np.zeros((1, 4)), num_dist=0)
    st = st.ravel()
    print(f"{os.path.basename(pa.LIB_PATH)} {problem} N=60 cold: mean {st['elapsed_s'].mean()*1e3:.2f} ms, max {st['elapsed_s'].max()*1e3:.0f} ms, "
          f"{st['iterations'].mean():.1f} iterations, {st['n_hess_psi_prod'].mean():.0f} Hv, converged {int((st['status']==1).sum())}")
    b.close()
