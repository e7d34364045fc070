#!/usr/bin/env python3
"""Throughput of the hanging_chain N=60 replay workload tiled R times into one launch
(development aid: where the latency-bound 600-instance pass ends and the memory-bound regime starts)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa

d = np.load(os.path.join(ROOT, "tests/golden/mpc_hanging_chain_N60_warm300.npz"))
x0 = np.vstack([d["x0"], d["x0"]]); u0 = np.vstack([d["guess"], np.zeros_like(d["guess"])])
for R in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    X0, U0 = np.tile(x0, (R, 1)), np.tile(u0, (R, 1))
    B = len(X0)
    batch = pa.PantrBatch("hanging_chain", 60, B, opts=pa.DODO_PANTR_OPTS + ("alm.max_time=5s", "launch_order=auto"))
    lds, ws = batch.footprint()
    for _ in range(2):
        u, st = batch.solve(X0, U0)
        ms = batch.last_kernel_ms()
    alg = pa.algorithmic_bytes(st, 60, 57, 3)
    ok = np.array_equal(u[:600], u[-600:]) and int((st["status"] == 1).sum()) == B
    print(f"R={R}: {B} instances, LDS {lds} B/workgroup, kernel {ms:.1f} ms -> {B/(ms*1e-3):.0f} steps/s, "
          f"algorithmic {alg/(ms*1e-3)/1e9:.0f} GB/s ({100*alg/(ms*1e-3)/8e12:.1f} % of 8 TB/s), all converged+identical: {ok}")
    batch.close()
