#!/usr/bin/env python3
"""Regenerates the fixtures under tests/golden/ with the CPU oracle (oracle/liboracle.so).

The reference holds no golden vectors for the PANTR path, and neither
alpaqa nor CasADi exist in this pipeline, so these fixtures are produced by this project's
own CPU restatement ("parity unpinned").  They pin the oracle against accidental change and
provide the replay workload of BASELINE.json configs[1]:

  mpc_hanging_chain_N60_warm300.npz
      closed loop of drivers/mpc-driver.cpp:123-210 with the options of
      new-benchmarks-paper/dodo.py:12-33,94-95 and num_sim=300 warm=true:
      x0[300,57]  state each timed solve was solved for
      guess[300,180] warm start it began from (previous solution shifted by nu)
      sol[300,180]   its solution, iters/cg/hv: its counters
  mpc_hanging_chain_N{10,20}_cold8.npz   small closed loops (cold) for CPU-side regression
  problem_functions_N7.npz               psi, grad psi, Hv at seeded points
  solves_rows_f1_f4.npz                  rows f1/f4 of SURVEY.md section 8: one cold solve per
      (problem, method) with the option sets of dodo.py:12-40,97-105 -- hanging_chain N=20 and
      quadcopter / realistic_quadcopter N=12 x {pantr, panoc, strucpanoc}: initial state, solution,
      multipliers and the solver's counters; plus psi, grad psi, yhat and Hv of the two quadcopter
      models at a seeded point

Usage: python tests/golden/make_golden.py
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob  # noqa: E402


def closed_loop(N, num_sim, warm):
    t = time.time()
    states, inputs, guesses, stats = ob.mpc(N, num_sim, warm, num_dist=5, u_dist=(-1.0, 1.0, 1.0),
                                            params=ob.dodo_params())
    print(f"N={N} num_sim={num_sim} warm={warm}: {time.time() - t:.1f} s, "
          f"converged {sum(s['status'] == 1 for s in stats)}/{num_sim}")
    return dict(
        x0=states[:-1], x_final=states[-1], guess=guesses, sol=inputs,
        status=np.array([s["status"] for s in stats], np.int32),
        iters=np.array([s["iterations"] for s in stats], np.int32),
        cg=np.array([s["cg_iterations"] for s in stats], np.int32),
        hv=np.array([s["n_hess_psi_prod"] for s in stats], np.int32),
        eps=np.array([s["eps"] for s in stats]),
        cpu_time=np.array([s["elapsed_s"] for s in stats]),
    )


def main():
    np.savez_compressed(os.path.join(HERE, "mpc_hanging_chain_N60_warm300.npz"),
                        **closed_loop(60, 300, True))
    for N in (10, 20):
        np.savez_compressed(os.path.join(HERE, f"mpc_hanging_chain_N{N}_cold8.npz"),
                            **closed_loop(N, 8, False))
    rng = np.random.default_rng(7)
    N = 7
    x0 = ob.initial_state()
    for _ in range(5):
        x0 = ob.simulate(x0, (-1.0, 1.0, 1.0))
    u = rng.uniform(-1, 1, (3, 3 * N))
    v = rng.standard_normal((3, 3 * N))
    psi, grad, Hv = [], [], []
    for i in range(3):
        f, g = ob.psi_grad(N, x0, u[i])
        psi.append(f)
        grad.append(g)
        Hv.append(ob.hess_prod(N, x0, u[i], v[i]))
    np.savez_compressed(os.path.join(HERE, "problem_functions_N7.npz"), N=N, x0=x0, u=u, v=v,
                        psi=np.array(psi), grad=np.array(grad), Hv=np.array(Hv))
    rows_f1_f4()


ROWS = [(p, m) for p in ("hanging_chain", "quadcopter", "realistic_quadcopter")
        for m in ("pantr", "panoc", "strucpanoc")]
COUNTERS = ("status", "iterations", "outer_iterations", "n_psi", "n_psi_grad_psi", "n_grad_L",
            "n_hess_psi_prod", "n_prox_grad_step", "lbfgs_rejected", "linesearch_backtracks")


def solve_row(problem, method):
    """One cold solve from the benchmark's initial state; returns x0, u, y, counters."""
    if problem == "hanging_chain":
        N = 20
        x0 = ob.initial_state()
        for _ in range(5):
            x0 = ob.simulate(x0, (-1.0, 1.0, 1.0))
        u, st, _ = ob.solve(N, x0, np.zeros(3 * N), ob.dodo_params(method))
        y = np.zeros(0)
    else:
        N = 12
        x0 = ob.quad_initial_state()
        u, y, st = ob.quad_solve(problem, N, x0, ob.quad_initial_guess(N), params=ob.quad_dodo_params(problem, method))
    return N, x0, u, y, np.array([st[k] for k in COUNTERS], np.int64), np.array([st["eps"], st["final_psi"]])


def rows_f1_f4():
    out = {}
    for problem, method in ROWS:
        N, x0, u, y, cnt, fl = solve_row(problem, method)
        key = f"{problem}.{method}"
        out.update({f"{key}.N": N, f"{key}.x0": x0, f"{key}.u": u, f"{key}.y": y, f"{key}.counters": cnt,
                    f"{key}.eps_psi": fl})
        print(f"{key}: N={N} status {cnt[0]} iterations {cnt[1]} outer {cnt[2]}")
    rng = np.random.default_rng(11)
    for problem in ("quadcopter", "realistic_quadcopter"):
        N, n, m = 9, 36, 40
        x0 = ob.quad_initial_state() + 0.05 * rng.standard_normal(9)
        u = ob.quad_initial_guess(N) + 0.3 * rng.standard_normal(n)
        y, Sig, v = rng.standard_normal(m), 10 ** rng.uniform(0, 3, m), rng.standard_normal(n)
        psi, grad, yhat = ob.quad_psi_grad(problem, N, x0, y, Sig, u)
        out.update({f"{problem}.fun.x0": x0, f"{problem}.fun.u": u, f"{problem}.fun.y": y, f"{problem}.fun.Sig": Sig,
                    f"{problem}.fun.v": v, f"{problem}.fun.psi": psi, f"{problem}.fun.grad": grad,
                    f"{problem}.fun.yhat": yhat, f"{problem}.fun.Hv": ob.quad_hess_prod(problem, N, x0, y, Sig, u, v)})
    np.savez_compressed(os.path.join(HERE, "solves_rows_f1_f4.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rows":
        rows_f1_f4()
        sys.exit(0)
    main()
