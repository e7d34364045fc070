#!/usr/bin/env python3
"""Shards of the N=120 stress batch with one and with two wavefronts per instance (speculation=off | on). Development aid."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa
from pantr_amd import workloads
w = workloads.stress_batch(1024)
for B in (128, 256, 512):
    idx = np.arange(0, 1024, 1024 // B)
    for spec in ("on", "off"):
        b = pa.PantrBatch("hanging_chain", 120, B, opts=tuple(w["opts"]) + (f"speculation={spec}",))
        ms = []
        for rep in range(3):
            u, st = b.solve(w["x0"][idx], w["u0"][idx]); ms.append(b.last_kernel_ms())
        print(f"N=120 {B} instances speculation={spec} ({b.kernel_name()}): pass {min(ms):.1f} ms, mean solve {st['elapsed_s'].mean()*1e3:.1f} ms, slowest {st['elapsed_s'].max()*1e3:.1f}")
        b.close()
