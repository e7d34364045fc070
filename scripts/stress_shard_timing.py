#!/usr/bin/env python3
"""One GPU's shard of BASELINE.json configs[4] on 8 / 4 / 2 GPUs: 128 / 256 / 512 of the 1024 N=120 instances.
Pass time; PANTR_LIB_PATH selects the library.  Development aid."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa
from pantr_amd import workloads
w = workloads.stress_batch(1024)
for B in (128, 256, 512):
    idx = np.arange(0, 1024, 1024 // B)
    b = pa.PantrBatch("hanging_chain", 120, B, opts=w["opts"])
    ms = []
    for rep in range(3):
        u, st = b.solve(w["x0"][idx], w["u0"][idx]); ms.append(b.last_kernel_ms())
    print(f"{os.path.basename(pa.LIB_PATH)} N=120 {B} instances ({b.kernel_name()}): pass {min(ms):.1f} ms -> {B/min(ms)*1e3:.0f} steps/s, mean solve {st['elapsed_s'].mean()*1e3:.1f} ms, converged {int((st['status']==1).sum())}")
    b.close()
