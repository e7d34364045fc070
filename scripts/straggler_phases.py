#!/usr/bin/env python3
"""The instance that decides the pass: its phases inside the batch of 600 and solved on its own.

Profiling build (scripts/phase_timing.sh builds it; run with PANTR_LIB_PATH pointing at it).
Development aid: says which phases of the straggler a full chip slows down.
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa

if "+phase-timing" not in pa.version():
    raise SystemExit(f"{pa.LIB_PATH} is not the profiling build ({pa.version()})")

d = np.load(os.path.join(ROOT, "tests/golden/mpc_hanging_chain_N60_warm300.npz"))
x0 = np.vstack([d["x0"], d["x0"]]); u0 = np.vstack([d["guess"], np.zeros_like(d["guess"])])
spec = sys.argv[1] if len(sys.argv) > 1 else "auto"
opts = pa.DODO_PANTR_OPTS + ("alm.max_time=3s", f"speculation={spec}")

PHASES = (("hess_prod", 0, "n_hess_psi_prod"), ("psi_grad(cand)", 1, "n_psi_grad_psi"),
          ("psi(xhat)", 2, "n_psi"), ("grad_L(xhat)+tape", 3, "n_grad_L"))


def line(tag, s):
    el = float(s["elapsed_s"])
    parts, rest = [], el
    for label, f, cnt in PHASES:
        t = float(s["phase_s"][f]); rest -= t
        parts.append(f"{label} {t*1e3:6.2f} ms ({t/max(int(s[cnt]),1)*1e6:5.1f} us x {int(s[cnt])})")
    print(f"{tag:28s} {el*1e3:6.2f} ms, {int(s['iterations'])} it | " + " | ".join(parts) +
          f" | rest {rest*1e3:5.2f} ms ({rest/int(s['iterations'])*1e6:.1f} us/it)")


def solve(idx):
    b = pa.PantrBatch("hanging_chain", 60, len(idx), opts=opts)
    for _ in range(2):
        _, st = b.solve(x0[idx], u0[idx])
    k = b.kernel_name(len(idx)); ms = b.last_kernel_ms()
    b.close()
    return st.ravel(), k, ms


full, k, ms = solve(np.arange(600))
print(f"batch of 600: {k}, pass {ms:.2f} ms")
order = np.argsort(-full["elapsed_s"])[:3]
for i in order:
    line(f"instance {i} in the batch", full[i])
    for size in (1, 64, 256):
        # the same instance first in a batch of the `size` slowest instances
        idx = np.concatenate([[i], np.argsort(-full["elapsed_s"])[:size][np.argsort(-full["elapsed_s"])[:size] != i]])[:size]
        st, k, ms = solve(idx)
        line(f"  … among the {size} slowest", st[0])
