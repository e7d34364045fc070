"""GPU parity tests of the comparator solvers: PANOC+ with L-BFGS and
with the structured L-BFGS direction (drivers/mpc-driver.cpp:371-384, dodo.py:34-40), the HIP path
through the C ABI against the CPU oracle.  Same bar as test_gpu_parity.py: bit-exact.
"""
import os

import numpy as np
import pytest

import oracle_binding as ob
import pantr_amd as pa
from common import NU, disturbed_state, random_inputs, random_states

pytestmark = pytest.mark.gpu
RTOL = 1e-10  # north star: fp64 iterates within 1e-10 relative of the CPU solver
COUNTERS = ("iterations", "qub_backtracks", "n_psi", "n_grad_psi", "n_psi_grad_psi", "n_grad_L",
            "n_hess_psi_prod", "n_prox_grad_step", "linesearch_failures", "linesearch_backtracks",
            "lbfgs_failures", "lbfgs_rejected", "tau_1_accepted", "count_tau")


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(20231004)


@pytest.mark.parametrize("method", ["panoc", "strucpanoc"])
@pytest.mark.parametrize("N,memory", [(5, 50), (22, 50), (60, 50), (22, 3)])
def test_chain_solve_iterates_match_oracle(rng, method, N, memory):
    """Whole solves: same iterates (per-iteration trace), same counters, same solution; memory = 3
    exercises the ring wrap-around, memory = 50 > n the dodo.py setting at short horizons."""
    B, cap = 4, 1500
    x0s = np.vstack([disturbed_state()[None, :], random_states(rng, B - 1, 0.02)])
    u0s = np.vstack([np.zeros((1, NU * N)), random_inputs(rng, B - 1, N, -0.3, 0.3)])
    batch = pa.PantrBatch("hanging_chain", N, B, opts=pa.DODO_OPTS[method] + (f"accel.memory={memory}",))
    batch.set_trace(B, cap)
    u, st = batch.solve(x0s, u0s)
    tr = batch.get_trace()
    batch.close()
    prm = ob.dodo_params(method, lbfgs_memory=memory)
    for b in range(B):
        u_o, st_o, tr_o = ob.solve(N, x0s[b], u0s[b], prm, trace_cap=cap)
        assert st["status"][b] == st_o["status"] == 1
        for key in COUNTERS:
            assert st[key][b] == st_o[key], key
        assert st["sum_tau"][b] == st_o["sum_tau"] and st["n_hess_psi_prod"][b] == 0
        k = min(st_o["iterations"] + 1, cap)
        assert len(tr_o) == k
        a, o = tr[b, :k, :7], tr_o[:, :7]
        fin = np.isfinite(o)
        assert np.array_equal(np.isfinite(a), fin)
        assert rel(a[fin], o[fin]) <= RTOL and np.array_equal(a[fin], o[fin])
        assert rel(u[b], u_o) <= RTOL and np.array_equal(u[b], u_o)
        assert st["eps"][b] == st_o["eps"] and st["eps"][b] <= 1e-8
        assert st["final_psi"][b] == st_o["final_psi"]


def test_three_methods_reach_the_same_optimum():
    N = 30
    x0, u0 = disturbed_state()[None, :], np.zeros((1, NU * N))
    sols = {}
    for method in ("pantr", "panoc", "strucpanoc"):
        batch = pa.PantrBatch("hanging_chain", N, 1, opts=pa.DODO_OPTS[method])
        u, st = batch.solve(x0, u0)
        batch.close()
        assert st["status"][0] == 1
        sols[method] = (u[0], st["final_psi"][0], st["iterations"][0])
    for method in ("panoc", "strucpanoc"):
        assert abs(sols[method][1] - sols["pantr"][1]) <= 1e-9 * abs(sols["pantr"][1])
        assert np.abs(sols[method][0] - sols["pantr"][0]).max() <= 1e-5
        assert sols[method][2] > sols["pantr"][2]  # first-order methods need more iterations


def quad_opts(problem, method):
    """dodo.py:12-40 + :97-105"""
    o = list(pa.DODO_OPTS[method]) + ["num_dist=0", "solver.max_iter=250"]
    if problem == "realistic_quadcopter":
        o += ["alm.initial_tolerance=1e-1"]
    return o


@pytest.mark.parametrize("method", ["panoc", "strucpanoc"])
@pytest.mark.parametrize("problem,N", [("quadcopter", 12), ("realistic_quadcopter", 20)])
def test_quadcopter_alm_solve_matches_oracle(rng, method, problem, N):
    B = 3
    b = pa.PantrBatch(problem, N, B, opts=quad_opts(problem, method))
    x0 = np.vstack([ob.quad_initial_state(), ob.quad_initial_state() + 0.02 * rng.standard_normal((B - 1, 9))])
    u0 = np.tile(ob.quad_initial_guess(N), (B, 1))
    u, y, st = b.solve_al(x0, u0)
    b.close()
    prm = ob.quad_dodo_params(problem, method)
    for i in range(B):
        u_o, y_o, st_o = ob.quad_solve(problem, N, x0[i], u0[i], params=prm)
        assert st["status"][i] == st_o["status"] == 1
        for key in COUNTERS + ("outer_iterations", "inner_convergence_failures"):
            assert st[key][i] == st_o[key], key
        assert rel(u[i], u_o) <= RTOL and np.array_equal(u[i], u_o)
        assert rel(y[i], y_o) <= RTOL and np.array_equal(y[i], y_o)
        assert st["eps"][i] == st_o["eps"] and st["delta"][i] == st_o["delta"]


def test_unsupported_comparator_options_fail_loudly():
    x0, u0 = disturbed_state()[None, :], np.zeros((1, NU * 5))
    for extra in (("method=strucpanoc", "accel.memory=50"),               # dir.hessian_vec defaults to true
                  ("method=panoc", "accel.cbfgs.ε=1e-3"),
                  ("method=panoc", "accel.memory=100000")):
        batch = pa.PantrBatch("hanging_chain", 5, 1, opts=pa.DODO_ALM_OPTS + extra)
        with pytest.raises(pa.PantrError):
            batch.solve(x0, u0)
        batch.close()
    with pytest.raises(pa.PantrError):
        pa.PantrBatch("hanging_chain", 5, 1, opts=("method=fbetrust",))


# ---- the committed fixtures of rows f1/f4, without consulting the oracle -----------------------
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "solves_rows_f1_f4.npz")
ROW_COUNTERS = ("status", "iterations", "outer_iterations", "n_psi", "n_psi_grad_psi", "n_grad_L",
                "n_hess_psi_prod", "n_prox_grad_step", "lbfgs_rejected", "linesearch_backtracks")


@pytest.mark.parametrize("problem", ["hanging_chain", "quadcopter", "realistic_quadcopter"])
@pytest.mark.parametrize("method", ["pantr", "panoc", "strucpanoc"])
def test_golden_rows_f1_f4_on_the_device(problem, method):
    d = np.load(GOLDEN)
    key = f"{problem}.{method}"
    N, x0 = int(d[f"{key}.N"]), d[f"{key}.x0"][None, :]
    if problem == "hanging_chain":
        b = pa.PantrBatch(problem, N, 1, opts=pa.DODO_OPTS[method])
        u, st = b.solve(x0, np.zeros((1, NU * N)))
        y = np.zeros((1, 0))
    else:
        opts = list(pa.DODO_OPTS[method]) + ["solver.max_iter=250"]
        if problem == "realistic_quadcopter":
            opts += ["alm.initial_tolerance=1e-1"] + (["accel.tol_scale=1"] if method == "pantr" else [])
        b = pa.PantrBatch(problem, N, 1, opts=opts)
        u, y, st = b.solve_al(x0, b.initial_guess()[None, :])
    b.close()
    assert np.array_equal(u[0], d[f"{key}.u"]) and np.array_equal(y[0], d[f"{key}.y"])
    assert [int(st[k][0]) for k in ROW_COUNTERS] == list(d[f"{key}.counters"])
    assert st["eps"][0] == d[f"{key}.eps_psi"][0] and st["final_psi"][0] == d[f"{key}.eps_psi"][1]


@pytest.mark.parametrize("problem", ["quadcopter", "realistic_quadcopter"])
def test_golden_quadcopter_functions_on_the_device(problem):
    d = np.load(GOLDEN)
    g = lambda k: d[f"{problem}.fun.{k}"]  # noqa: E731
    b = pa.PantrBatch(problem, 9, 1)
    psi, grad, yhat = b.eval_psi_grad_psi_al(g("x0")[None], g("u")[None], g("y")[None], g("Sig")[None])
    Hv = b.eval_hess_psi_prod_al(g("x0")[None], g("u")[None], g("y")[None], g("Sig")[None], g("v")[None])
    b.close()
    assert psi[0] == g("psi") and np.array_equal(grad[0], g("grad")) and np.array_equal(yhat[0], g("yhat"))
    assert np.array_equal(Hv[0], g("Hv"))

