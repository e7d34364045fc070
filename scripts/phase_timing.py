#!/usr/bin/env python3
"""Per-phase solve time from the profiling build of the library (run through scripts/phase_timing.sh,
which builds it to its own file and points PANTR_LIB_PATH at it)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa
from pantr_amd import workloads

if "+phase-timing" not in pa.version():
    raise SystemExit(f"{pa.LIB_PATH} is not the profiling build ({pa.version()}): run scripts/phase_timing.sh")


def report(name, st):
    st = st.ravel()
    el = st["elapsed_s"].sum()
    it = st["iterations"].sum()
    print(f"{name}: {len(st)} solves, mean {el/len(st)*1e3:.2f} ms, {it/len(st):.1f} iterations, "
          f"{st['n_hess_psi_prod'].sum()/len(st):.1f} Hv")
    rest = el
    for label, field, cnt in (("hess_prod", 0, "n_hess_psi_prod"), ("psi_grad(cand)", 1, "n_psi_grad_psi"),
                              ("psi(xhat)", 2, "n_psi"), ("grad_L(xhat)+tape", 3, "n_grad_L")):
        t, c = st["phase_s"][:, field].sum(), st[cnt].sum()
        rest -= t
        print(f"   {label:18s} {100*t/el:5.1f} %   {t/max(c,1)*1e6:8.1f} us per call ({c/len(st):.1f} calls)")
    print(f"   {'everything else':18s} {100*rest/el:5.1f} %   {rest/max(it,1)*1e6:8.1f} us per iteration")


def report_panoc(name, st):
    st = st.ravel()
    el, it = st["elapsed_s"].sum(), st["iterations"].sum()
    print(f"{name}: {len(st)} solves, mean {el/len(st)*1e3:.2f} ms, {it/len(st):.1f} iterations")
    rest = el
    for label, field, cnt in (("L-BFGS apply", 0, "iterations"), ("psi_grad", 1, "n_psi_grad_psi"),
                              ("psi(xhat)", 2, "n_psi"), ("L-BFGS update", 3, "iterations")):
        t, c = st["phase_s"][:, field].sum(), st[cnt].sum()
        rest -= t
        print(f"   {label:18s} {100*t/el:5.1f} %   {t/max(c,1)*1e6:8.1f} us per call")
    print(f"   {'everything else':18s} {100*rest/el:5.1f} %   {rest/max(it,1)*1e6:8.1f} us per iteration")


which = sys.argv[1:] or ["hanging_chain", "quadcopter", "realistic_quadcopter"]
if "panoc" in which or "strucpanoc" in which:
    d = np.load(os.path.join(ROOT, "tests/golden/mpc_hanging_chain_N60_warm300.npz"))
    x0 = np.vstack([d["x0"][:60], d["x0"][:60]]); u0 = np.vstack([d["guess"][:60], np.zeros_like(d["guess"][:60])])
    for m in [w for w in which if w in ("panoc", "strucpanoc")]:
        b = pa.PantrBatch("hanging_chain", 60, len(x0), opts=pa.DODO_OPTS[m])
        report_panoc(f"hanging_chain N=60 {m} (60 warm + 60 cold)", b.solve(x0, u0)[1])
        b.close()
    which = [w for w in which if w not in ("panoc", "strucpanoc")]
    if not which:
        sys.exit(0)
if "hanging_chain" in which:
    d = np.load(os.path.join(ROOT, "tests/golden/mpc_hanging_chain_N60_warm300.npz"))
    x0 = np.vstack([d["x0"], d["x0"]]); u0 = np.vstack([d["guess"], np.zeros_like(d["guess"])])
    b = pa.PantrBatch("hanging_chain", 60, len(x0), opts=pa.DODO_PANTR_OPTS + ("alm.max_time=3s",))
    report("hanging_chain N=60 replay (600)", b.solve(x0, u0)[1])
    b.close()
for problem in which:
    if problem == "hanging_chain":
        continue
    for N in (20, 60):
        task = [t for t in workloads.dodo_pantr_tasks(problem) if t["horizon"] == N and not t["warm"]][0]
        b = pa.PantrBatch(problem, N, 1, opts=tuple(o for o in task["argv"][3:] if o.startswith(("solver.", "alm.", "dir.", "accel."))))
        _, _, st = b.mpc(60, False, np.zeros((1, 4)), num_dist=0)
        report(f"{problem} N={N} cold closed loop", st)
        b.close()
