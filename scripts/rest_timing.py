#!/usr/bin/env python3
"""Where a PANTR iteration's time outside the function evaluations goes (library built with
`make phase EXTRA_HIPFLAGS=-DPANTR_REST_TIMING`; with `-DPANTR_REST_SPLIT=<base>` on top of it the four fields
hold pieces <base>..<base>+3 of the sixteen named in pantr_kernel.hip.h).  Development aid."""
import os, re, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa
from pantr_amd import workloads

if "+rest-timing" not in pa.version():
    raise SystemExit(f"{pa.LIB_PATH} is not the rest-timing build ({pa.version()})")
split = re.search(r"\+rest-split=(\d+)", pa.version())
PIECES = ["swap + wait for the helper", "bound test", "progress check", "stop criterion",
          "prox copy / adoption", "prox step", "request", "inactive set + rhs",
          "CG start", "CG pass 1 + d'Bd", "CG pass 2", "CG rest + model value",
          "full step", "candidate", "candidate's prox step", "ratio test + radius"]

N = int(os.environ.get("HORIZON", "60"))
for problem in sys.argv[1:] or ["quadcopter", "realistic_quadcopter", "hanging_chain"]:
    if problem == "hanging_chain":
        d = np.load(os.path.join(ROOT, "tests/golden/mpc_hanging_chain_N60_warm300.npz"))
        x0 = np.vstack([d["x0"], d["x0"]]); u0 = np.vstack([d["guess"], np.zeros_like(d["guess"])])
        b = pa.PantrBatch(problem, 60, 600, opts=pa.DODO_PANTR_OPTS + ("alm.max_time=3s",))
        st = b.solve(x0, u0)[-1].ravel()
    else:
        w = workloads.closed_loop_batch(problem, N)
        b = pa.PantrBatch(problem, N, len(w["x0"]), opts=w["opts"])
        st = b.solve_al(w["x0"], w["u0"], w["y0"])[-1].ravel()
    i = int(np.argmax(st["elapsed_s"])); s = st[i]; it = int(s["iterations"])
    ph = s["phase_s"]
    print(f"{problem} N={60 if problem == 'hanging_chain' else N} ({b.kernel_name()}), pass {b.last_kernel_ms():.2f} ms: slowest instance {i}: {s['elapsed_s']*1e3:.1f} ms, {it} iterations, {s['cg_iterations']} CG iterations")
    if split:
        base = int(split.group(1))
        print("   per iteration, us: " + " | ".join(f"[{base+k}] {PIECES[base+k]} {ph[k]/it*1e6:.2f}" for k in range(4)))
    else:
        print(f"   per iteration, us: evaluations {ph[0]/it*1e6:.1f} | end of the CG loop .. stop criterion {ph[1]/it*1e6:.1f} | prox iterate .. CG start {ph[2]/it*1e6:.1f} "
              f"| CG algebra {ph[3]/it*1e6:.1f} ({ph[3]/max(int(s['cg_iterations']),1)*1e6:.2f} per CG iteration, two clock reads included)")
    b.close()
