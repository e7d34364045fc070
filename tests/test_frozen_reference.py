"""A reference that does not move with the arithmetic specification.

tests/golden/frozen_r01.npz holds the values the oracle produced at the end of round 1 (extracted from
commit be709d6: tests/golden/{solves_rows_f1_f4, problem_functions_N7, mpc_hanging_chain_N20_cold8}.npz)
and is NEVER regenerated.  The bit-exact fixtures next to it are regenerated whenever oracle and
kernels change an operation order together, so they cannot show drift across
rounds; these tests can: problem functions within 1e-12 relative, solutions within 1e-5, on the CPU
oracle and - marked gpu - on the HIP path through the C ABI.  (Iteration counts are not compared:
a changed rounding legitimately moves accept/reject decisions.)
"""
import os

import numpy as np
import pytest

import oracle_binding as ob

FROZEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frozen_r01.npz")
FUN_RTOL, SOL_RTOL, PSI_RTOL = 1e-12, 1e-5, 1e-6
NU = 3
ROWS = [(p, m) for p in ("hanging_chain", "quadcopter", "realistic_quadcopter")
        for m in ("pantr", "panoc", "strucpanoc")]


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)) if a.size else 0.0


@pytest.fixture(scope="module")
def frozen():
    return np.load(FROZEN)


# ---- CPU oracle ------------------------------------------------------------------------------------
def test_oracle_chain_functions_against_round_1(frozen):
    g = lambda k: frozen[f"chain_fun.{k}"]  # noqa: E731
    N = int(g("N"))
    for i in range(len(g("u"))):
        f, grad = ob.psi_grad(N, g("x0"), g("u")[i])
        assert rel(f, g("psi")[i]) <= FUN_RTOL and rel(grad, g("grad")[i]) <= FUN_RTOL
        assert rel(ob.hess_prod(N, g("x0"), g("u")[i], g("v")[i]), g("Hv")[i]) <= FUN_RTOL


@pytest.mark.parametrize("problem", ["quadcopter", "realistic_quadcopter"])
def test_oracle_quadcopter_functions_against_round_1(frozen, problem):
    g = lambda k: frozen[f"rows.{problem}.fun.{k}"]  # noqa: E731
    psi, grad, yhat = ob.quad_psi_grad(problem, 9, g("x0"), g("y"), g("Sig"), g("u"))
    assert rel(psi, g("psi")) <= FUN_RTOL and rel(grad, g("grad")) <= FUN_RTOL and rel(yhat, g("yhat")) <= FUN_RTOL
    assert rel(ob.quad_hess_prod(problem, 9, g("x0"), g("y"), g("Sig"), g("u"), g("v")), g("Hv")) <= FUN_RTOL


@pytest.mark.parametrize("problem,method", ROWS)
def test_oracle_solves_against_round_1(frozen, problem, method):
    key = f"rows.{problem}.{method}"
    N, x0 = int(frozen[f"{key}.N"]), frozen[f"{key}.x0"]
    if problem == "hanging_chain":
        u, st, _ = ob.solve(N, x0, np.zeros(NU * N), ob.dodo_params(method))
        y = np.zeros(0)
    else:
        u, y, st = ob.quad_solve(problem, N, x0, ob.quad_initial_guess(N), params=ob.quad_dodo_params(problem, method))
    assert st["status"] == 1 == int(frozen[f"{key}.counters"][0]) and st["eps"] <= 1e-8
    assert rel(u, frozen[f"{key}.u"]) <= SOL_RTOL and rel(y, frozen[f"{key}.y"]) <= SOL_RTOL
    assert rel(st["final_psi"], frozen[f"{key}.eps_psi"][1]) <= PSI_RTOL


def test_oracle_chain_closed_loop_against_round_1(frozen):
    g = lambda k: frozen[f"chain_mpc_N20.{k}"]  # noqa: E731
    states, inputs, _, stats = ob.mpc(20, 8, warm=False)
    assert all(s["status"] == 1 for s in stats)
    assert rel(states[:-1], g("x0")) <= SOL_RTOL and rel(inputs, g("sol")) <= SOL_RTOL


# ---- HIP path --------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_device_chain_functions_against_round_1(frozen):
    import pantr_amd as pa
    g = lambda k: frozen[f"chain_fun.{k}"]  # noqa: E731
    N, B = int(g("N")), len(g("u"))
    b = pa.PantrBatch("hanging_chain", N, B)
    x0 = np.tile(g("x0"), (B, 1))
    psi, grad = b.eval_psi_grad_psi(x0, g("u"))
    Hv = b.eval_hess_psi_prod(x0, g("u"), g("v"))
    b.close()
    assert rel(psi, g("psi")) <= FUN_RTOL and rel(grad, g("grad")) <= FUN_RTOL and rel(Hv, g("Hv")) <= FUN_RTOL


@pytest.mark.gpu
@pytest.mark.parametrize("problem", ["quadcopter", "realistic_quadcopter"])
def test_device_quadcopter_functions_against_round_1(frozen, problem):
    import pantr_amd as pa
    g = lambda k: frozen[f"rows.{problem}.fun.{k}"]  # noqa: E731
    b = pa.PantrBatch(problem, 9, 1)
    psi, grad, yhat = b.eval_psi_grad_psi_al(g("x0")[None], g("u")[None], g("y")[None], g("Sig")[None])
    Hv = b.eval_hess_psi_prod_al(g("x0")[None], g("u")[None], g("y")[None], g("Sig")[None], g("v")[None])
    b.close()
    assert rel(psi[0], g("psi")) <= FUN_RTOL and rel(grad[0], g("grad")) <= FUN_RTOL
    assert rel(yhat[0], g("yhat")) <= FUN_RTOL and rel(Hv[0], g("Hv")) <= FUN_RTOL


@pytest.mark.gpu
@pytest.mark.parametrize("problem,method", ROWS)
def test_device_solves_against_round_1(frozen, problem, method):
    import pantr_amd as pa
    key = f"rows.{problem}.{method}"
    N, x0 = int(frozen[f"{key}.N"]), frozen[f"{key}.x0"][None, :]
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
    assert int(st["status"][0]) == 1 and st["eps"][0] <= 1e-8
    assert rel(u[0], frozen[f"{key}.u"]) <= SOL_RTOL and rel(y[0], frozen[f"{key}.y"]) <= SOL_RTOL
    assert rel(st["final_psi"][0], frozen[f"{key}.eps_psi"][1]) <= PSI_RTOL
