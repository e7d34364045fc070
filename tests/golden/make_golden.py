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


if __name__ == "__main__":
    main()
