"""CPU tests of the oracle: derivative checks, independent cross-checks, golden fixtures.

The reference holds no golden vectors for the PANTR path (parity
unpinned), so the oracle is pinned by (i) mathematical certificates that do not depend on
the algorithm under test, (ii) SciPy's independent Steihaug-Toint implementation and an
independently written numpy chain model, (iii) this project's own committed fixtures.
"""
import os

import numpy as np
import pytest
from scipy.optimize._trustregion_ncg import CGSteihaugSubproblem

import oracle_binding as ob
from common import NX, NU, disturbed_state, random_inputs, random_states


# ---- independent numpy transcription of the model (hanging_chain.py:24-80 read as text) ------
def numpy_fc(s, u, D=0.1, L=0.033, m=0.03, g=9.81):
    x = s[:30].reshape(10, 3)
    v = s[30:].reshape(9, 3)
    dist = np.vstack([x[0:1], x[1:] - x[:-1]])
    norm = np.sqrt((dist * dist).sum(axis=1))
    F = dist * (D * (1 - L / norm))[:, None]
    acc = (F[1:] - F[:-1]) / m + np.array([0, 0, -g])
    return np.concatenate([v.ravel(), u, acc.ravel()])


def numpy_rk4(s, u, Ts=0.1):
    k1 = numpy_fc(s, u)
    k2 = numpy_fc(s + Ts * k1 / 2, u)
    k3 = numpy_fc(s + Ts * k2 / 2, u)
    k4 = numpy_fc(s + Ts * k3, u)
    return s + (Ts / 6) * (k1 + 2 * k2 + 2 * k3 + k4)


def numpy_cost(N, x0, u, alpha=25, beta=1, gamma=0.01):
    s, f = x0.copy(), 0.0
    e = np.array([1.0, 0, 0])
    for k in range(N):
        uk = u[3 * k:3 * k + 3]
        f += alpha * np.sum((s[27:30] - e) ** 2) + beta * np.sum(s[30:] ** 2) + gamma * np.sum(uk ** 2)
        s = numpy_rk4(s, uk)
    return f + alpha * np.sum((s[27:30] - e) ** 2) + beta * np.sum(s[30:] ** 2)


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(42)


def test_initial_state_and_box():
    x0 = ob.initial_state()
    assert x0.shape == (NX,)
    assert np.allclose(x0[0:30:3], np.arange(1, 11) / 10) and np.count_nonzero(x0) == 10
    lb, ub = np.zeros(3), np.zeros(3)
    ob.lib().oracle_chain_box(lb, ub)
    assert np.all(lb == -1) and np.all(ub == 1)


def test_dynamics_match_independent_numpy_model(rng):
    for _ in range(5):
        s = random_states(rng, 1)[0]
        u = rng.uniform(-1, 1, 3)
        assert np.allclose(ob.fc(s, u), numpy_fc(s, u), rtol=1e-13, atol=1e-13)
        assert np.allclose(ob.simulate(s, u), numpy_rk4(s, u), rtol=1e-13, atol=1e-13)


def test_rk4_has_order_four():
    """Halving the step of the oracle's integrator cuts the error ~16x against a fine reference."""
    s, u = disturbed_state(), np.array([0.3, -0.2, 0.1])

    def integrate(n_sub):
        x, h = s.copy(), 0.1 / n_sub
        for _ in range(n_sub):
            k1 = numpy_fc(x, u); k2 = numpy_fc(x + h * k1 / 2, u)
            k3 = numpy_fc(x + h * k2 / 2, u); k4 = numpy_fc(x + h * k3, u)
            x = x + (h / 6) * (k1 + 2 * k2 + 2 * k3 + k4)
        return x
    ref = integrate(64)
    e1 = np.abs(ob.simulate(s, u) - ref).max()
    e2 = np.abs(integrate(2) - ref).max()
    assert 10 < e1 / e2 < 24


@pytest.mark.parametrize("N", [1, 6, 25])
def test_cost_and_gradient(rng, N):
    x0, u = random_states(rng, 1)[0], random_inputs(rng, 1, N)[0]
    f, g = ob.psi_grad(N, x0, u)
    assert ob.psi(N, x0, u) == f
    assert abs(f - numpy_cost(N, x0, u)) <= 1e-12 * abs(f)
    h = 1e-6
    gfd = np.array([(ob.psi(N, x0, u + h * e) - ob.psi(N, x0, u - h * e)) / (2 * h)
                    for e in np.eye(NU * N)])
    assert np.abs(g - gfd).max() <= 1e-6 * np.abs(g).max()


@pytest.mark.parametrize("N", [3, 20])
def test_hessian_vector_product(rng, N):
    x0, u = random_states(rng, 1)[0], random_inputs(rng, 1, N)[0]
    v, w = rng.standard_normal(NU * N), rng.standard_normal(NU * N)
    Hv, Hw = ob.hess_prod(N, x0, u, v), ob.hess_prod(N, x0, u, w)
    h = 1e-6
    fd = (ob.psi_grad(N, x0, u + h * v)[1] - ob.psi_grad(N, x0, u - h * v)[1]) / (2 * h)
    assert np.abs(Hv - fd).max() <= 1e-6 * np.abs(Hv).max()
    assert abs(w @ Hv - v @ Hw) <= 1e-12 * abs(w @ Hv)          # symmetry
    assert np.array_equal(ob.hess_prod(N, x0, u, np.zeros(NU * N)), np.zeros(NU * N))
    assert np.allclose(ob.hess_prod(N, x0, u, 2.5 * v), 2.5 * Hv, rtol=1e-13)  # linearity
    # the K/M tape product against the independent dual-number (forward-over-reverse) product
    Hd = ob.hess_prod(N, x0, u, v, dual=True)
    assert np.abs(Hv - Hd).max() <= 1e-12 * np.abs(Hd).max()


def test_prox_grad_step_is_box_projection(rng):
    N = 9
    x, g = random_inputs(rng, 1, N, -1.5, 1.5)[0], 20 * rng.standard_normal(NU * N)
    for gamma in (1.0, 0.01):
        xh, p, pTp, gTp = ob.prox_grad_step(N, gamma, x, g)
        assert np.allclose(xh, x + np.clip(-gamma * g, -1 - x, 1 - x), rtol=0, atol=0)
        assert np.array_equal(xh, x + p)
        assert np.isclose(pTp, p @ p, rtol=1e-14) and np.isclose(gTp, p @ g, rtol=1e-13)


def test_wave_dot_order(rng):
    """oracle_dot = per-lane strided partial sums + 64-lane butterfly (the wavefront order)."""
    for n in (1, 63, 64, 65, 180, 360):
        a, b = rng.standard_normal(n), rng.standard_normal(n)
        part = np.zeros(64)
        for l in range(64):
            acc = 0.0
            for i in range(l, n, 64):
                acc = acc + a[i] * b[i]
            part[l] = acc
        for m in (32, 16, 8, 4, 2, 1):
            part = part + part[np.arange(64) ^ m]
        assert ob.lib().oracle_dot(a, b, n) == part[0]


# ---- Steihaug-Toint CG against SciPy's independent implementation ----------------------------
def scipy_steihaug(H, g, radius, tol_scale):
    sub = CGSteihaugSubproblem(np.zeros(len(g)), lambda x: 0.0, lambda x: g, lambda x: H,
                               lambda x, p: H @ p)
    # SciPy hard-codes tolerance = min(0.5, sqrt(|g|)) |g|; reproduce the oracle's scale by
    # scaling the problem is not possible, so compare in the regimes where the tolerance does
    # not decide the exit (boundary exits), and where it does only for tol_scale = SciPy's.
    return sub.solve(radius)


@pytest.mark.parametrize("case", ["interior", "boundary", "negative_curvature"])
def test_steihaug_matches_scipy(rng, case):
    n = 24
    A = rng.standard_normal((n, n))
    if case == "negative_curvature":
        H = A + A.T
        radius = 1.5
    else:
        H = A @ A.T + 0.5 * np.eye(n)
        radius = 100.0 if case == "interior" else 0.05
    g = rng.standard_normal(n)
    p_sp, hits = scipy_steihaug(H, g, radius, None)
    # SciPy: tolerance = min(0.5, sqrt(gnorm)) * gnorm -> tol_scale 0.5 when sqrt(gnorm) >= 0.5...
    gn = np.linalg.norm(g)
    scale = min(0.5, np.sqrt(gn)) / min(1.0, np.sqrt(gn))
    p_or, m_or, its = ob.steihaug_dense(H, g, radius, tol_scale=scale)
    assert np.allclose(p_or, p_sp, rtol=1e-9, atol=1e-12)
    assert np.isclose(m_or, g @ p_or + 0.5 * p_or @ H @ p_or, rtol=1e-12)
    assert hits == (case != "interior")
    if case != "interior":
        assert np.isclose(np.linalg.norm(p_or), radius, rtol=1e-12)


# ---- whole solves: optimality certificates (results.hpp:258-286 semantics) --------------------
@pytest.mark.parametrize("N,warm", [(10, False), (30, False), (30, True)])
def test_solve_reaches_kkt_point(N, warm):
    x0 = disturbed_state()
    u0 = np.zeros(NU * N)
    u, st, tr = ob.solve(N, x0, u0, trace_cap=500)
    assert ob.STATUS[st["status"]] == "Converged" and st["eps"] <= 1e-8
    if warm:  # re-solve from the shifted solution: fewer iterations
        x1 = ob.simulate(x0, u[:3])
        u1, st1, _ = ob.solve(N, x1, np.concatenate([u[3:], u[-3:]]))
        assert st1["status"] == 1 and st1["iterations"] < st["iterations"]
        u, x0 = u1, x1
    assert np.all(np.abs(u) <= 1.0)                                   # box feasibility
    _, g = ob.psi_grad(N, x0, u)
    stationarity = np.abs(np.clip(u - g, -1, 1) - u).max()            # || Pi_C(x - grad L) - x ||_inf
    assert stationarity <= 1e-8
    assert len(tr) == st["iterations"] and st["accepted"] + st["rejected"] == st["iterations"]
    assert np.all(np.diff(tr[:, 7]) <= 1e-9 * np.abs(tr[:-1, 7]))     # FBE decreases monotonically


def test_mpc_loop_semantics():
    """mpc-driver.cpp:123-210: disturbance, untimed first solve, shift-by-nu warm start, simulate."""
    N, num_sim = 8, 5
    states, inputs, guesses, stats = ob.mpc(N, num_sim, warm=True)
    x = ob.initial_state()
    for _ in range(5):
        x = ob.simulate(x, (-1.0, 1.0, 1.0))
    u, _, _ = ob.solve(N, x, np.zeros(NU * N))
    x = ob.simulate(x, u[:NU])
    for i in range(num_sim):
        assert np.array_equal(states[i], x)
        u[:-NU] = u[NU:].copy()
        assert np.array_equal(guesses[i], u)
        u, st, _ = ob.solve(N, x, u)
        assert np.array_equal(inputs[i], u) and st["status"] == stats[i]["status"] == 1
        x = ob.simulate(x, u[:NU])
    assert np.array_equal(states[num_sim], x)
    _, _, g_cold, _ = ob.mpc(N, 2, warm=False)
    assert np.all(g_cold == 0)


# ---- committed fixtures ----------------------------------------------------------------------
def test_golden_problem_functions(golden_dir):
    d = np.load(os.path.join(golden_dir, "problem_functions_N7.npz"))
    N = int(d["N"])
    for i in range(len(d["u"])):
        f, g = ob.psi_grad(N, d["x0"], d["u"][i])
        assert f == d["psi"][i] and np.array_equal(g, d["grad"][i])
        assert np.array_equal(ob.hess_prod(N, d["x0"], d["u"][i], d["v"][i]), d["Hv"][i])


@pytest.mark.parametrize("N", [10, 20])
def test_golden_closed_loop_cold(golden_dir, N):
    d = np.load(os.path.join(golden_dir, f"mpc_hanging_chain_N{N}_cold8.npz"))
    states, inputs, guesses, stats = ob.mpc(N, 8, warm=False)
    assert np.array_equal(states[:-1], d["x0"]) and np.array_equal(inputs, d["sol"])
    assert [s["iterations"] for s in stats] == list(d["iters"])
    assert [s["n_hess_psi_prod"] for s in stats] == list(d["hv"])


def test_golden_replay_fixture_is_a_pure_function_of_its_inputs(golden_dir):
    """Replay batching: each recorded step re-solved in isolation reproduces the closed loop."""
    d = np.load(os.path.join(golden_dir, "mpc_hanging_chain_N60_warm300.npz"))
    assert d["x0"].shape == (300, NX) and d["guess"].shape == (300, 180)
    assert np.all(d["status"] == 1) and np.all(d["eps"] <= 1e-8)
    idx = [0, 137, 299]
    us, stats = ob.solve_batch(60, d["x0"][idx], d["guess"][idx], nthreads=3)
    for j, i in enumerate(idx):
        assert np.array_equal(us[j], d["sol"][i]) and stats[j]["iterations"] == d["iters"][i]
    # the warm start of step i+1 is the solution of step i shifted by nu
    assert np.array_equal(d["guess"][1:, :-NU], d["sol"][:-1, NU:])
    assert np.array_equal(d["guess"][1:, -NU:], d["sol"][:-1, -NU:])
    # and the next state is the plant step of the first input block
    assert np.array_equal(d["x0"][1], ob.simulate(d["x0"][0], d["sol"][0, :NU]))


# ---- quadcopter problems: general constraints, ALM -------------------
def test_deterministic_sincos():
    xs = np.concatenate([np.linspace(-10, 10, 401), np.random.default_rng(3).uniform(-1e3, 1e3, 400)])
    for x in xs:
        s, c = ob.sincos(float(x))
        assert abs(s - np.sin(x)) <= 2.5e-16 and abs(c - np.cos(x)) <= 2.5e-16
    assert ob.sincos(0.0) == (0.0, 1.0)


def numpy_quad_fc(problem, s, u):
    """Independent transcription of quadcopter.py:30-46 / realistic_quadcopter.py:30-50."""
    th, at, om = s[6:9], u[0], u[1:4]
    if problem == "quadcopter":
        cr, sr, cp, sp, cy, sy = np.cos(th[0]), np.sin(th[0]), np.cos(th[1]), np.sin(th[1]), np.cos(th[2]), np.sin(th[2])
        R = np.array([[cy*cp, cy*sp*sr - sy*cr, cy*sp*cr + sy*sr],
                      [sy*cp, sy*sp*sr + cy*cr, sy*sp*cr - cy*sr],
                      [-sp, cp*sr, cp*cr]])
        dth = om
    else:
        phi, tht, psi = th
        cps, cph, cth, sps, sph, sth = np.cos(psi), np.cos(phi), np.cos(tht), np.sin(psi), np.sin(phi), np.sin(tht)
        R = np.array([[cps*cth - sph*sps*sth, -cph*sps, cps*sth + cth*sph*sps],
                      [cth*sps + cps*sph*sth, cph*cps, sps*sth - cps*cth*sph],
                      [-cph*sth, sph, cph*cth]])
        Om = np.array([[cth, 0, -cph*sth], [0, 1, sph], [sth, 0, cph*cth]])
        dth = np.linalg.solve(Om, om)
    a = R @ np.array([0, 0, at]) + np.array([0, 0, -9.81])
    return np.concatenate([s[3:6], a, dth])


@pytest.mark.parametrize("problem", ["quadcopter", "realistic_quadcopter"])
def test_quadcopter_dynamics_and_derivatives(rng, problem):
    N = 5
    n, m = 4 * N, 4 * N + 4
    for _ in range(3):
        s = ob.quad_initial_state() + 0.3 * rng.standard_normal(9)
        u = np.array([9.81, 0, 0, 0]) + 0.3 * rng.standard_normal(4)
        assert np.allclose(ob.quad_fc(problem, s, u), numpy_quad_fc(problem, s, u), rtol=1e-13, atol=1e-13)
    x0 = ob.quad_initial_state() + 0.05 * rng.standard_normal(9)
    u = ob.quad_initial_guess(N) + 0.3 * rng.standard_normal(n)
    y, Sig = rng.standard_normal(m), 10 ** rng.uniform(0, 3, m)
    psi, g, yh = ob.quad_psi_grad(problem, N, x0, y, Sig, u)
    # psi = f + 1/2 yhat'd with zeta = g + y/Sigma (formulation.hpp:79-86)
    f, gc = ob.quad_f_g(problem, N, x0, u)
    lb = np.tile([-np.pi / 2, -np.pi / 2, np.cos(0.55), -np.inf], N + 1)
    ub = np.tile([np.pi / 2, np.pi / 2, np.inf, 0.0], N + 1)
    zeta = gc + y / Sig
    d = zeta - np.clip(zeta, lb, ub)
    assert np.allclose(yh, Sig * d, rtol=1e-13, atol=1e-13)
    assert abs(psi - (f + 0.5 * (Sig * d) @ d)) <= 1e-12 * abs(psi)
    h = 1e-6
    gfd = np.array([(ob.quad_psi_grad(problem, N, x0, y, Sig, u + h * e)[0] -
                     ob.quad_psi_grad(problem, N, x0, y, Sig, u - h * e)[0]) / (2 * h) for e in np.eye(n)])
    assert np.abs(g - gfd).max() <= 1e-6 * np.abs(g).max()
    v, w = rng.standard_normal(n), rng.standard_normal(n)
    Hv, Hw = ob.quad_hess_prod(problem, N, x0, y, Sig, u, v), ob.quad_hess_prod(problem, N, x0, y, Sig, u, w)
    fd = (ob.quad_psi_grad(problem, N, x0, y, Sig, u + h * v)[1] -
          ob.quad_psi_grad(problem, N, x0, y, Sig, u - h * v)[1]) / (2 * h)
    assert np.abs(Hv - fd).max() <= 1e-6 * np.abs(Hv).max()
    assert abs(w @ Hv - v @ Hw) <= 1e-11 * abs(w @ Hv)
    # the adjoint sweep the solvers use (realistic_quadcopter: affine in the attitude adjoints) against the
    # stage-by-stage sweep through the full stage adjoint
    gg = ob.quad_grad_generic(problem, N, x0, y, Sig, u)
    assert np.abs(g - gg).max() <= 1e-13 * np.abs(gg).max()
    # the stage-matrix tape against dual numbers through the whole horizon
    Hd = ob.quad_hess_prod(problem, N, x0, y, Sig, u, v, dual=True)
    assert np.abs(Hv - Hd).max() <= 1e-12 * np.abs(Hd).max()
    assert np.array_equal(ob.quad_hess_prod(problem, N, x0, y, Sig, u, np.zeros(n)), np.zeros(n))


@pytest.mark.parametrize("problem,N", [("quadcopter", 12), ("realistic_quadcopter", 20)])
def test_alm_reaches_kkt_point(problem, N):
    """BASELINE.json configs[0] (quadcopter horizon 12) and a realistic_quadcopter case: the KKT
    certificate of drivers/results.hpp:258-286 on the ALM solution."""
    x0, u0 = ob.quad_initial_state(), ob.quad_initial_guess(N)
    u, y, st = ob.quad_solve(problem, N, x0, u0)
    assert ob.STATUS[st["status"]] == "Converged" and st["eps"] <= 1e-8 and st["delta"] <= 1e-8
    assert 1 < st["outer_iterations"] <= 300
    ulb, uub = np.tile([0, -0.1, -0.1, -0.1], N), np.tile([9.81 * 5, 0.1, 0.1, 0.1], N)
    assert np.all(u >= ulb) and np.all(u <= uub)
    # stationarity: || Pi_C(x - grad L(x,y)) - x ||_inf with grad L = grad psi at (y, Sigma) s.t. yhat = y
    f, g = ob.quad_f_g(problem, N, x0, u)
    lb = np.tile([-np.pi / 2, -np.pi / 2, np.cos(0.55), -np.inf], N + 1)
    ub = np.tile([np.pi / 2, np.pi / 2, np.inf, 0.0], N + 1)
    viol = g - np.clip(g, lb, ub)
    assert np.abs(viol).max() <= 1e-7                                    # constraint violation
    assert np.abs(y * viol).max() <= 1e-8 * max(1.0, np.abs(y).max())      # complementarity: |y| x the ALM tolerance on the violation
    # gradient of the Lagrangian via psi with huge-penalty-free trick: Sigma=1, y' = y - g + clip(g + y)
    Sig = np.ones_like(y)
    zeta_target = y  # want yhat = Sigma (zeta - clip(zeta)) = y  with zeta = g + y'/Sigma
    yprime = np.where(y != 0, y + np.clip(g + 0 * y, lb, ub) - g, 0.0)
    _, gradL, yhat = ob.quad_psi_grad(problem, N, x0, yprime, Sig, u)
    assert np.allclose(yhat, y, atol=1e-6)
    stationarity = np.abs(np.clip(u - gradL, ulb, uub) - u).max()
    assert stationarity <= 1e-6
    # multipliers have the sign their bound allows
    assert np.all(y[2::4] <= 1e-12) and np.all(y[3::4] >= -1e-12)


def test_quadcopter_closed_loop():
    """mpc-driver.cpp:123-210 with multipliers: every step converges; the loop drives the
    quadcopter towards its target (0.25, 0.25, 0.5) around the cylinder at the origin."""
    N, num_sim = 10, 6
    _, _, cold = ob.quad_mpc("quadcopter", N, num_sim, warm=False)
    st_w, in_w, warm = ob.quad_mpc("quadcopter", N, num_sim, warm=True)
    assert all(s["status"] == 1 for s in cold + warm)
    assert np.all(np.isfinite(st_w)) and st_w.shape == (num_sim + 1, 9)
    target = np.array([0.25, 0.25, 0.5])
    assert np.linalg.norm(st_w[-1, :3] - target) < np.linalg.norm(st_w[0, :3] - target)
    assert np.all(st_w[:, 0] ** 2 + st_w[:, 1] ** 2 >= 0.01 - 1e-6)   # stays outside the cylinder


# ---- comparators: PANOC+ with L-BFGS / structured L-BFGS ----------
def _dense_lbfgs(S, Y, q, gamma):
    """H q with H from the BFGS recursion H+ = (I - rho s y')H(I - rho y s') + rho s s', H0 = gamma I."""
    n = len(q)
    H = gamma * np.eye(n)
    for s_, y_ in zip(S, Y):
        rho = 1.0 / (y_ @ s_)
        V = np.eye(n) - rho * np.outer(y_, s_)
        H = V.T @ H @ V + rho * np.outer(s_, s_)
    return H @ q


@pytest.mark.parametrize("memory,count", [(5, 3), (4, 9)])
def test_lbfgs_two_loop_matches_dense_bfgs(rng, memory, count):
    """The two-loop recursion (alpaqa::LBFGS::apply) against the dense recursion it abbreviates;
    count > memory exercises the ring (only the newest `memory` pairs count)."""
    n = 12
    A = rng.standard_normal((n, n))
    A = A @ A.T + n * np.eye(n)                       # SPD: y = A s has positive curvature
    S = rng.standard_normal((count, n))
    Y = S @ A
    q = rng.standard_normal(n)
    ok, Hq = ob.lbfgs_apply(S, Y, q, memory)
    keep = slice(max(0, count - memory), count)
    gamma = (Y[-1] @ S[-1]) / (Y[-1] @ Y[-1])         # step size from the newest pair
    assert ok and np.allclose(Hq, _dense_lbfgs(S[keep], Y[keep], q, gamma), rtol=1e-10, atol=1e-12)
    # on an index set J: the same recursion on the sub-vectors, nothing touched outside J
    mask = np.arange(n) % 3 != 0
    ok, HqJ = ob.lbfgs_apply(S, Y, q, memory, mask)
    SJ, YJ = S[keep][:, mask], Y[keep][:, mask]
    gammaJ = (YJ[-1] @ SJ[-1]) / (YJ[-1] @ YJ[-1])
    assert ok and np.array_equal(HqJ[~mask], q[~mask])
    assert np.allclose(HqJ[mask], _dense_lbfgs(SJ, YJ, q[mask], gammaJ), rtol=1e-10, atol=1e-12)
    assert ob.lbfgs_apply(S[:0], Y[:0], q, memory)[0] is False  # no pairs: no direction


@pytest.mark.parametrize("method", ["panoc", "strucpanoc"])
def test_comparators_reach_the_pantr_optimum(method):
    N = 20
    x0 = disturbed_state()
    u_ref, st_ref, _ = ob.solve(N, x0, np.zeros(NU * N), ob.dodo_params())
    u, st, tr = ob.solve(N, x0, np.zeros(NU * N), ob.dodo_params(method), trace_cap=2000)
    assert ob.STATUS[st["status"]] == "Converged" and st["eps"] <= 1e-8
    assert abs(st["final_psi"] - st_ref["final_psi"]) <= 1e-10 * abs(st_ref["final_psi"])
    assert np.abs(u - u_ref).max() <= 1e-6 and np.all(np.abs(u) <= 1.0)
    assert st["n_hess_psi_prod"] == 0 and st["iterations"] > st_ref["iterations"]
    assert st["tau_1_accepted"] + st["linesearch_failures"] <= st["count_tau"] <= st["iterations"]
    # the line search makes the forward-backward envelope decrease monotonically (col 5 = fbe)
    fbe = tr[:, 5]
    assert np.all(np.diff(fbe) <= 1e-9 * (1 + np.abs(fbe[:-1])))
    assert len(tr) == st["iterations"] + 1 and tr[-1, 0] == st["eps"]


@pytest.mark.parametrize("method", ["panoc", "strucpanoc"])
def test_comparators_alm_reaches_kkt_point(method):
    problem, N = "quadcopter", 12
    x0, u0 = ob.quad_initial_state(), ob.quad_initial_guess(N)
    u, y, st = ob.quad_solve(problem, N, x0, u0, params=ob.quad_dodo_params(problem, method))
    assert ob.STATUS[st["status"]] == "Converged" and st["eps"] <= 1e-8 and st["delta"] <= 1e-8
    f, g = ob.quad_f_g(problem, N, x0, u)
    lb = np.tile([-np.pi / 2, -np.pi / 2, np.cos(0.55), -np.inf], N + 1)
    ub = np.tile([np.pi / 2, np.pi / 2, np.inf, 0.0], N + 1)
    assert np.abs(g - np.clip(g, lb, ub)).max() <= 1e-7
    u_ref, y_ref, st_ref = ob.quad_solve(problem, N, x0, u0)
    f_ref, _ = ob.quad_f_g(problem, N, x0, u_ref)
    assert abs(f - f_ref) <= 1e-6 * abs(f_ref)  # same local solution as PANTR


# ---- committed fixtures of the rows built around the path (f1, f4) --------
ROWS_F = [(p, m) for p in ("hanging_chain", "quadcopter", "realistic_quadcopter")
          for m in ("pantr", "panoc", "strucpanoc")]
ROW_COUNTERS = ("status", "iterations", "outer_iterations", "n_psi", "n_psi_grad_psi", "n_grad_L",
                "n_hess_psi_prod", "n_prox_grad_step", "lbfgs_rejected", "linesearch_backtracks")


@pytest.mark.parametrize("problem,method", ROWS_F)
def test_golden_rows_f1_f4(golden_dir, problem, method):
    """The oracle reproduces tests/golden/solves_rows_f1_f4.npz (make_golden.py rows) bit for bit."""
    d = np.load(os.path.join(golden_dir, "solves_rows_f1_f4.npz"))
    key = f"{problem}.{method}"
    N, x0 = int(d[f"{key}.N"]), d[f"{key}.x0"]
    if problem == "hanging_chain":
        u, st, _ = ob.solve(N, x0, np.zeros(NU * N), ob.dodo_params(method))
        y = np.zeros(0)
    else:
        u, y, st = ob.quad_solve(problem, N, x0, ob.quad_initial_guess(N), params=ob.quad_dodo_params(problem, method))
    assert np.array_equal(u, d[f"{key}.u"]) and np.array_equal(y, d[f"{key}.y"])
    assert [st[k] for k in ROW_COUNTERS] == list(d[f"{key}.counters"])
    assert st["eps"] == d[f"{key}.eps_psi"][0] and st["final_psi"] == d[f"{key}.eps_psi"][1]
    assert st["status"] == 1 and st["eps"] <= 1e-8


@pytest.mark.parametrize("problem", ["quadcopter", "realistic_quadcopter"])
def test_golden_quadcopter_functions(golden_dir, problem):
    d = np.load(os.path.join(golden_dir, "solves_rows_f1_f4.npz"))
    g = lambda k: d[f"{problem}.fun.{k}"]  # noqa: E731
    psi, grad, yhat = ob.quad_psi_grad(problem, 9, g("x0"), g("y"), g("Sig"), g("u"))
    assert psi == g("psi") and np.array_equal(grad, g("grad")) and np.array_equal(yhat, g("yhat"))
    assert np.array_equal(ob.quad_hess_prod(problem, 9, g("x0"), g("y"), g("Sig"), g("u"), g("v")), g("Hv"))



@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 63, 64, 65, 130])
def test_scan_orders_of_the_specification(rng, n):
    """prefix_scan / affine_scan (the summation and composition orders the quadcopter-family kernels'
    wavefront scans follow): equal to the sequential recursion up to rounding, exactly on inputs whose
    sums are exact, block boundaries at 64 included."""
    r = rng.standard_normal(n)
    e = ob.prefix_scan(r, 0.75)
    assert e.shape == (n + 1,) and e[0] == 0.75
    assert np.allclose(e, 0.75 + np.concatenate([[0.0], np.cumsum(r)]), rtol=0, atol=1e-13 * max(1, n))
    ri = rng.integers(-1000, 1000, n).astype(float)  # integer sums are exact in any order
    assert np.array_equal(ob.prefix_scan(ri, 3.0), 3.0 + np.concatenate([[0.0], np.cumsum(ri)]))
    maps = np.column_stack([1 + 0.1 * rng.standard_normal(n), 0.1 * rng.standard_normal(n), 0.1 * rng.standard_normal(n),
                            1 + 0.1 * rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)]).reshape(n, 6)
    y0 = np.array([0.3, -0.2])
    out = ob.affine_scan(maps, y0)
    y, seq = y0.copy(), [y0.copy()]
    for m in maps:
        y = np.array([m[0] * y[0] + m[1] * y[1] + m[4], m[2] * y[0] + m[3] * y[1] + m[5]])
        seq.append(y)
    seq = np.array(seq)
    assert out.shape == (n + 1, 2) and np.array_equal(out[0], y0)
    assert np.abs(out - seq).max() <= 1e-12 * max(1.0, np.abs(seq).max())
    shifts = np.column_stack([np.ones(n), np.zeros(n), np.zeros(n), np.ones(n), rng.integers(-9, 9, n), rng.integers(-9, 9, n)]).astype(float)
    acc = np.array([1.0, 2.0]) + np.concatenate([[[0.0, 0.0]], np.cumsum(shifts[:, 4:], axis=0)]) if n else np.array([[1.0, 2.0]])
    assert np.array_equal(ob.affine_scan(shifts, [1.0, 2.0]), acc)  # translations compose exactly
