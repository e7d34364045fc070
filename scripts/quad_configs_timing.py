#!/usr/bin/env python3
"""Pass time of the quadcopter-family configurations of BASELINE.json (configs[0]: quadcopter N=12, 60 warm steps;
configs[2]: realistic_quadcopter N=60, 120 instances) and of the quadcopter N=60 replay: the quick A/B numbers for
kernel variants (PANTR_LIB_PATH selects the library).  Development aid."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa
from pantr_amd import workloads
for problem, N, starts in (("quadcopter", 12, (True,)), ("quadcopter", 60, (True, False)), ("realistic_quadcopter", 60, (True, False))):
    w = workloads.closed_loop_batch(problem, N, starts=starts)
    b = pa.PantrBatch(problem, N, len(w["x0"]), opts=w["opts"])
    ms = []
    for rep in range(4):
        st = b.solve_al(w["x0"], w["u0"], w["y0"])[-1].ravel()
        ms.append(b.last_kernel_ms())
    print(f"{os.path.basename(pa.LIB_PATH)} {problem} N={N} {len(w['x0'])} instances ({b.kernel_name()}): pass {min(ms):.3f} ms "
          f"-> {len(w['x0'])/min(ms)*1e3:.0f} steps/s; mean solve {st['elapsed_s'].mean()*1e3:.3f} ms, slowest {st['elapsed_s'].max()*1e3:.2f} ms "
          f"({int(st['iterations'][np.argmax(st['elapsed_s'])])} iterations)")
    b.close()
