"""GPU test of the C++ host side: the benchmark-mpc-driver CLI clone and its results file.

The results module is loaded the way new-benchmarks-paper/benchmark_util.py:4-9 loads it and
compared with the oracle's closed loop (drivers/mpc-driver.cpp:123-210).
"""
import importlib.util
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
import pantr_amd as pa
from pantr_amd import workloads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "pantr-cdc2023-experiments_amd", "bin", "benchmark-mpc-driver")
SELFTEST = os.path.join(ROOT, "pantr-cdc2023-experiments_amd", "bin", "facade-selftest")


def load_results(path, name):
    spec = importlib.util.spec_from_file_location("", os.path.join(path, f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.results


@pytest.mark.parametrize("warm", [True, False])
def test_driver_closed_loop_matches_oracle(tmp_path, warm):
    N, num_sim = 20, 12
    name = f"mpc-hanging_chain-{N},{num_sim}-{'warm' if warm else 'cold'}-pantr"
    opts = list(pa.DODO_PANTR_OPTS) + list(pa.DODO_HANGING_CHAIN_OPTS)
    cmd = [DRIVER, "hanging_chain", str(N), "ss2p", "method=pantr", f"num_sim={num_sim}",
           f"results_name={name}", f"warm={'true' if warm else 'false'}", "replay=true"] + opts
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stderr.count("Converged") == num_sim + 1          # status line per solve (:218)
    assert "identical to the closed loop" in r.stdout
    res = load_results(str(tmp_path), name)
    states, inputs, guesses, stats = ob.mpc(N, num_sim, warm)
    assert res["states"].shape == (57, num_sim + 1) and res["inputs"].shape == (3 * N, num_sim)
    assert np.array_equal(res["states"].T, states) and np.array_equal(res["inputs"].T, inputs)
    assert np.all(res["runtimes"] > 0) and len(res["runtimes"]) == num_sim
    ev = res["evaluations"]
    # the reference's convention (drivers/results.hpp:63-87, summed by benchmark_util.py:12-27): alpaqa counts the
    # call of hess_ψ_prod also where the product is known to be zero; the products evaluated ride along
    assert ev["hess_ψ_prod"] == sum(s["n_hess_psi_prod_calls"] for s in stats)
    assert ev["hess_ψ_prod_evaluated"] == sum(s["n_hess_psi_prod"] for s in stats)
    assert ev["hess_ψ_prod"] > ev["hess_ψ_prod_evaluated"] > 0
    assert ev["ψ"] == sum(s["n_psi"] for s in stats)
    assert ev["ψ_grad_ψ"] == sum(s["n_psi_grad_psi"] for s in stats)
    assert ev["grad_L"] == sum(s["n_grad_L"] for s in stats)
    assert res["problem"] == "hanging_chain" and res["nvar"] == 3 * N and res["ncon"] == 0
    assert res["horizon"] == N and res["formulation"] == "ss2p" and res["second_order"] is False
    assert res["solver"].startswith("ALMSolver<PANTRSolver<") and "method=pantr" in res["opts"]


@pytest.mark.parametrize("method,solver", [("panoc", "ALMSolver<PANOCSolver<LBFGSDirection>>"),
                                           ("strucpanoc", "ALMSolver<PANOCSolver<StructuredLBFGSDirection>>")])
def test_driver_comparators_closed_loop_matches_oracle(tmp_path, method, solver):
    """Row f4 through the CLI: method=panoc / strucpanoc with the options of dodo.py:34-40."""
    N, num_sim = 10, 8
    task = [t for t in workloads.dodo_tasks("hanging_chain", method, [N], num_sim) if t["warm"]][0]
    r = subprocess.run([DRIVER] + task["argv"] + ["replay=true"], cwd=tmp_path, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stderr.count("Converged") == num_sim + 1 and "identical to the closed loop" in r.stdout
    res = load_results(str(tmp_path), task["name"])
    states, inputs, guesses, stats = ob.mpc(N, num_sim, True, params=ob.dodo_params(method))
    assert np.array_equal(res["states"].T, states) and np.array_equal(res["inputs"].T, inputs)
    ev = res["evaluations"]
    assert ev["hess_ψ_prod"] == 0 and ev["ψ"] == sum(s["n_psi"] for s in stats)
    assert ev["ψ_grad_ψ"] == sum(s["n_psi_grad_psi"] for s in stats)
    assert ev["grad_L"] == sum(s["n_grad_L"] for s in stats)
    assert res["solver"] == solver and f"method={method}" in res["opts"]


def test_driver_errors(tmp_path):
    r = subprocess.run([DRIVER, "hanging_chain", "10", "ss2p", "method=nope"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode != 0 and "Unknown solver 'nope'" in r.stderr
    r = subprocess.run([DRIVER, "no_such_problem", "10", "ss2p", "method=pantr"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode != 0 and "unknown problem" in r.stderr
    r = subprocess.run([DRIVER, "hanging_chain", "10", "ss2p", "method=pantr", "solver.bogus=1"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode != 0 and "Invalid key" in r.stderr
    r = subprocess.run([DRIVER, "hanging_chain", "10", "ss2p", "method=pantr", "u_dist=1,2"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode != 0 and "Invalid u_dist size" in r.stderr


@pytest.mark.parametrize("problem", ["quadcopter", "realistic_quadcopter"])
def test_driver_quadcopter_closed_loop_matches_oracle(tmp_path, problem):
    """The CLI clone on a problem with general constraints (ALM, multiplier shift)."""
    N, num_sim = 12, 6
    name = f"mpc-{problem}-{N},{num_sim}-warm-pantr"
    opts = list(pa.DODO_PANTR_OPTS) + ["num_dist=0", "solver.max_iter=250"]
    if problem == "realistic_quadcopter":
        opts += ["accel.tol_scale=1", "alm.initial_tolerance=1e-1"]   # dodo.py:101-105
    cmd = [DRIVER, problem, str(N), "ss2p", "method=pantr", f"num_sim={num_sim}",
           f"results_name={name}", "warm=true", "replay=true"] + opts
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    res = load_results(str(tmp_path), name)
    states, inputs, stats = ob.quad_mpc(problem, N, num_sim, True)
    assert np.array_equal(res["states"].T, states) and np.array_equal(res["inputs"].T, inputs)
    assert res["nvar"] == 4 * N and res["ncon"] == 4 * N + 4 and np.all(res["runtimes"] > 0)
    assert res["evaluations"]["hess_ψ_prod"] == sum(s["n_hess_psi_prod_calls"] for s in stats)
    assert res["evaluations"]["hess_ψ_prod_evaluated"] == sum(s["n_hess_psi_prod"] for s in stats)


@pytest.mark.parametrize("problem", ["quadcopter", "realistic_quadcopter", "hanging_chain"])
def test_cpp_facade_eval_functions_match_ctypes_path(tmp_path, problem):
    """Every TypeErasedProblem::eval_* of host/alpaqa_compat.hpp, called once by bin/facade-selftest,
    against the ctypes path on the same inputs (and eval_grad_L against its definition: for m > 0 it
    must use y: grad_L(x, yhat(x; y, Sigma)) = grad psi(x; y, Sigma), formulation.hpp:79-99)."""
    N = 7
    rng = np.random.default_rng(11)
    b = pa.PantrBatch(problem, N, 1)
    n, m, nx, nu = b.n, b.m, b.nx, b.nu
    param = b.initial_state() + 0.03 * rng.standard_normal(nx)
    x = b.initial_guess() + 0.05 * rng.standard_normal(n)
    y, Sigma, v = rng.standard_normal(m), 10 ** rng.uniform(0, 3, m), rng.standard_normal(n)
    with open(tmp_path / "in.csv", "w") as f:
        for row in (param, x, y, Sigma, v):
            f.write(",".join(repr(float(t)) for t in row) + "\n")
    r = subprocess.run([SELFTEST, problem, str(N), str(tmp_path / "in.csv")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = {ln.split()[0]: np.array([float(t) for t in ln.split()[1:]]) for ln in r.stdout.splitlines() if ln}
    P, X, Y, S, V = param[None], x[None], y[None], Sigma[None], v[None]
    if m:
        psi, grad, yhat = b.eval_psi_grad_psi_al(P, X, Y, S)
        Hv = b.eval_hess_psi_prod_al(P, X, Y, S, V)
        g = b.eval_g(P, X)
        lbD, ubD = b.box_D()
        assert np.array_equal(out["yhat"], yhat[0]) and np.array_equal(out["g"], g[0])
        assert np.array_equal(out["proj_diff_g"], g[0] - np.clip(g[0], lbD, ubD))
        # eval_g against the oracle's constraint function
        f_o, g_o = ob.quad_f_g(problem, N, param, x)
        assert np.array_equal(g[0], g_o)
        # grad_L uses y: with y := yhat it reproduces grad psi, with another y it does not
        assert np.array_equal(b.eval_grad_L(P, X, yhat)[0], grad[0])
        assert not np.array_equal(out["grad_L"], grad[0])
    else:
        psi, grad = b.eval_psi_grad_psi(P, X)
        Hv = b.eval_hess_psi_prod(P, X, V)
        assert out["yhat"].size == 0 and out["g"].size == 0
    grad_L = b.eval_grad_L(P, X, Y)
    xh, p, sc = b.eval_prox_grad_step(0.05, X, grad)
    assert out["psi"][0] == psi[0] == out["psi2"][0]
    assert np.array_equal(out["grad_psi"], grad[0]) and np.array_equal(out["hess_psi_prod"], Hv[0])
    assert np.array_equal(out["grad_L"], grad_L[0])
    assert np.array_equal(out["xhat"], xh[0]) and np.array_equal(out["p"], p[0])
    assert np.array_equal(out["simulate"], b.simulate_dynamics(P, X[:, :nu])[0])
    lb, ub = b.box_C()
    assert np.array_equal(out["C_lb"], lb) and np.array_equal(out["C_ub"], ub)
    # KKT residuals with the semantics of drivers/results.hpp:258-286
    z = np.clip(x - grad_L[0], lb, ub) - x
    kkt = [np.max(np.abs(z)), 0.0, 0.0]
    if m:
        vv = g[0] - np.clip(g[0], lbD, ubD)
        kkt[1:] = [np.max(np.abs(vv)), np.max(np.abs(y * vv))]
    assert np.allclose(out["kkt"], kkt, rtol=1e-13, atol=0)
    b.close()


def test_driver_uses_sidecar_files_next_to_the_problem_path(tmp_path):
    """benchmark-mpc-driver <dir>/<name> ...: when the generated problem's sidecar files lie in <dir>
    (generate_problem.py:62-73), their bounds / initial state / initial guess drive the run
    (casadi-dll-wrapper.cpp:25-67); the closed loop equals the one the Python binding runs from the
    same files, and the tighter box binds."""
    N, num_sim = 15, 6
    d = tmp_path / "problems"
    d.mkdir()
    bound = np.tile([0.5, 0.3, 0.6], N)
    pa.write_problem_files(str(d), "hanging_chain", "ss2p", N, dict(C_lb=-bound, C_ub=bound))
    name = "run-sidecar"
    opts = list(pa.DODO_PANTR_OPTS) + list(pa.DODO_HANGING_CHAIN_OPTS)
    cmd = [DRIVER, str(d / "hanging_chain"), str(N), "ss2p", "method=pantr", f"num_sim={num_sim}",
           f"results_name={name}", "warm=true"] + opts
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    res = load_results(str(tmp_path), name)
    b = pa.PantrBatch(horizon=N, opts=pa.DODO_PANTR_OPTS, files=(str(d / "hanging_chain"), "ss2p"))
    states, inputs, stats = b.mpc(num_sim, True, np.array([[-1.0, 1.0, 1.0]]), num_dist=5)
    b.close()
    assert np.array_equal(res["inputs"].T, inputs[:, 0]) and np.array_equal(res["states"].T, states[:, 0])
    assert np.all(np.abs(inputs[:, 0]) <= bound) and np.any(np.abs(inputs[:, 0]) == bound)
    assert np.all(res["runtimes"] > 0)


def test_driver_print_interval_prints_the_device_iterates(tmp_path):
    """solver.print_interval=k (mpc-driver.cpp:52): the rows of the device trace in a fixed table, one line per
    k-th iteration (and the last) of every solve; at full precision they are the oracle's per-iteration trace."""
    import common
    N = 10
    opts = list(pa.DODO_PANTR_OPTS) + list(pa.DODO_HANGING_CHAIN_OPTS)
    cmd = [DRIVER, "hanging_chain", str(N), "ss2p", "method=pantr", "num_sim=1", "results_name=r",
           "solver.print_precision=16"] + opts

    def tables(stdout):
        out, cur = [], None
        for ln in stdout.splitlines():
            if not ln.startswith("[PANTR]"):
                continue
            f = ln.split()
            if f[1] == "iter":
                cur = []
                out.append(cur)
            elif f[1] == "exit":
                cur.append(("exit", f[2], int(f[4])))
            else:
                cur.append((int(f[1]), [float(x) for x in f[2:10]], int(f[10]), int(f[11]), f[12]))
        return out

    r = subprocess.run(cmd + ["solver.print_interval=1"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    t = tables(r.stdout)
    assert len(t) == 2                                          # the untimed first solve (:158) and the one MPC step
    x0 = common.disturbed_state()
    _, st, tr = ob.solve(N, x0, np.zeros(3 * N), trace_cap=500)
    rows, last = t[0][:-1], t[0][-1]
    assert last == ("exit", "Converged", st["iterations"]) and len(rows) == st["iterations"] == len(tr)
    for (k, v, nJ, cg, acc), o in zip(rows, tr):
        # table: phi_gamma psi ||p|| gamma eps Delta rho q_model | trace: eps psi pnorm gamma radius rho q_model fbe nJ cg accepted
        want = [o[7], o[1], o[2], o[3], o[0], o[4], o[5], o[6]]
        assert np.array_equal(np.array(v), np.array(want), equal_nan=True), (k, v, want)
        assert (nJ, cg, acc) == (int(o[8]), int(o[9]), "accepted" if o[10] else "rejected")
    assert [row[0] for row in rows] == list(range(len(rows)))
    r = subprocess.run(cmd + ["solver.print_interval=4"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    its = [row[0] for row in tables(r.stdout)[0][:-1]]
    assert its == sorted(set(list(range(0, len(rows), 4)) + [len(rows) - 1]))
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=300)   # default 0: silent
    assert r.returncode == 0 and "[PANTR]" not in r.stdout
