#!/usr/bin/env python3
"""Microseconds per Hessian-vector product INSIDE the solves of differently loaded launches (profiling build:
stats.phase_s[0] / n_hess_psi_prod), next to the lone-wavefront figure: what the memory system's load costs a
product (development aid for the prefetch depth of WaveProblem::hess_prod_tape).

    PANTR_LIB_PATH=.../libpantr_mi355x_phase.so python scripts/product_time_under_load.py [opt=value ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
import pantr_amd as pa
from pantr_amd import workloads

if "+phase-timing" not in pa.version():
    raise SystemExit(f"{pa.LIB_PATH} is not the profiling build ({pa.version()})")
extra = tuple(a for a in sys.argv[1:] if "=" in a)
d = np.load(os.path.join(ROOT, "tests/golden/mpc_hanging_chain_N60_warm300.npz"))
x0 = np.vstack([d["x0"], d["x0"]]); u0 = np.vstack([d["guess"], np.zeros_like(d["guess"])])
stress = workloads.stress_batch(1024)


def run(name, N, X0, U0, opts):
    b = pa.PantrBatch("hanging_chain", N, len(X0), opts=tuple(opts) + extra)
    b.solve(X0, U0)
    u, st = b.solve(X0, U0)
    ms = b.last_kernel_ms()
    t, c = st["phase_s"][:, 0].sum(), st["n_hess_psi_prod"].sum()
    other = st["elapsed_s"].sum() - st["phase_s"].sum()
    print(f"{name:44s} {b.kernel_name(len(X0)):38s} pass {ms:7.1f} ms  mean solve {st['elapsed_s'].mean()*1e3:7.1f} ms  "
          f"{t/c*1e6:6.1f} us per product  (psi_grad {st['phase_s'][:,1].sum()/max(st['n_psi_grad_psi'].sum(),1)*1e6:5.0f}, "
          f"grad_L+tape {st['phase_s'][:,3].sum()/max(st['n_grad_L'].sum(),1)*1e6:5.0f} us; forward halves "
          f"{100*st['sum_tau'].sum()/2.4e9/max(t,1e-30):3.0f} % of the products' time at 2.4 GHz)  converged {int((st['status']==1).sum())}")
    b.close()


base = tuple(pa.DODO_PANTR_OPTS) + ("alm.max_time=10s",)
run("N=60, 1 instance (the slowest, #419)", 60, x0[419:420], u0[419:420], base + ("speculation=off",))
run("N=60, 600 instances, one wavefront each", 60, x0, u0, base + ("speculation=off",))
run("N=60, 600 instances, two wavefronts each", 60, x0, u0, base + ("speculation=on",))
for R in (2, 4, 8):
    run(f"N=60, tiled x{R} ({600*R} instances)", 60, np.tile(x0, (R, 1)), np.tile(u0, (R, 1)), base)
run("N=120 stress, 1 instance", 120, stress["x0"][:1], stress["u0"][:1], tuple(stress["opts"]) + ("speculation=off",))
run("N=120 stress, 256 instances", 120, stress["x0"][:256], stress["u0"][:256], tuple(stress["opts"]) + ("speculation=off",))
run("N=120 stress, 1024 instances", 120, stress["x0"], stress["u0"], tuple(stress["opts"]))
