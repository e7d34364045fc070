#!/usr/bin/env python3
"""configs[3] (six horizons, six concurrent launches) with and without the helper wavefronts: which does
`speculation=auto` do better by?  Development aid (uses bench.py's resident batches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import torch
import bench
import pantr_amd as pa
from pantr_amd import workloads

dev = torch.device("cuda:0")
base = [workloads.closed_loop_batch("hanging_chain", N) for N in workloads.HORIZONS]
for spec in ("auto", "off", "auto", "off"):
    ws = [dict(w, opts=tuple(w["opts"]) + (f"speculation={spec}",)) for w in base]
    rs = [bench.Resident(w, torch, pa, dev, 0) for w in ws]
    wall, kms = bench.time_group(rs, torch)
    print(f"speculation={spec}: {sum(r.B for r in rs) / wall:.0f} steps/s, pass {wall*1e3:.1f} ms, kernels "
          + " ".join(f"{k:.1f}" for k in kms) + " ms")
    for r in rs:
        r.close()
