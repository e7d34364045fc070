"""GPU parity tests of SURVEY.md section 8 row f1: problems with general constraints (quadcopter,
realistic_quadcopter) and the ALM outer loop, against the CPU oracle, through the C ABI."""
import numpy as np
import pytest

import oracle_binding as ob
import pantr_amd as pa

pytestmark = pytest.mark.gpu
RTOL = 1e-10  # north star tolerance; the asserts below are exact where noted


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(2023)


def quad_opts(problem):
    """dodo.py:12-33 + :97-105"""
    o = list(pa.DODO_PANTR_OPTS) + ["num_dist=0", "solver.max_iter=250"]
    if problem == "realistic_quadcopter":
        o += ["accel.tol_scale=1", "alm.initial_tolerance=1e-1"]
    return o


@pytest.mark.parametrize("problem", ["quadcopter", "realistic_quadcopter"])
def test_problem_data(problem):
    b = pa.PantrBatch(problem, 12, 1)
    assert (b.n, b.m, b.nx, b.nu, b.nc, b.nc_N) == (48, 52, 9, 4, 4, 4)
    lb, ub = b.box_C()
    assert np.array_equal(lb[:4], [0, -0.1, -0.1, -0.1]) and np.array_equal(ub[:4], [9.81 * 5, 0.1, 0.1, 0.1])
    dl, du = b.box_D()
    assert np.allclose(dl[:4], [-np.pi / 2, -np.pi / 2, np.cos(0.55), -np.inf]) and du[3] == 0 and du[2] == np.inf
    assert np.array_equal(b.initial_state(), ob.quad_initial_state())
    assert np.array_equal(b.initial_guess(), ob.quad_initial_guess(12))
    b.close()


@pytest.mark.parametrize("problem", ["quadcopter", "realistic_quadcopter"])
@pytest.mark.parametrize("N", [1, 7, 30, 64, 70, 110])  # 64 | 70: one full block of stages per wavefront scan | a carry into a second; 110: 155 KiB of the 160 KiB LDS (realistic_quadcopter)
def test_psi_grad_hess_prod_match_oracle(rng, problem, N):
    B, n, m = 3, 4 * N, 4 * N + 4
    b = pa.PantrBatch(problem, N, B)
    x0 = ob.quad_initial_state()[None, :] + 0.05 * rng.standard_normal((B, 9))
    u = ob.quad_initial_guess(N)[None, :] + 0.3 * rng.standard_normal((B, n))
    y, Sig = rng.standard_normal((B, m)), 10 ** rng.uniform(0, 3, (B, m))
    v = rng.standard_normal((B, n))
    psi, grad, yhat = b.eval_psi_grad_psi_al(x0, u, y, Sig)
    Hv = b.eval_hess_psi_prod_al(x0, u, y, Sig, v)
    xn = b.simulate_dynamics(x0, u[:, :4])
    b.close()
    for i in range(B):
        p_o, g_o, yh_o = ob.quad_psi_grad(problem, N, x0[i], y[i], Sig[i], u[i])
        assert psi[i] == p_o and np.array_equal(grad[i], g_o) and np.array_equal(yhat[i], yh_o)
        Hv_o = ob.quad_hess_prod(problem, N, x0[i], y[i], Sig[i], u[i], v[i])
        assert rel(Hv[i], Hv_o) <= RTOL and np.array_equal(Hv[i], Hv_o)
        assert np.array_equal(xn[i], ob.quad_simulate(problem, x0[i], u[i, :4]))


@pytest.mark.parametrize("problem,N", [("quadcopter", 12), ("realistic_quadcopter", 20)])
def test_alm_solve_matches_oracle(rng, problem, N):
    """BASELINE.json configs[0] (quadcopter horizon 12) and a realistic_quadcopter case."""
    B, n = 3, 4 * N
    b = pa.PantrBatch(problem, N, B, opts=quad_opts(problem))
    x0 = np.vstack([ob.quad_initial_state(), ob.quad_initial_state() + 0.02 * rng.standard_normal((B - 1, 9))])
    u0 = np.tile(ob.quad_initial_guess(N), (B, 1))
    u, y, st = b.solve_al(x0, u0)
    b.close()
    for i in range(B):
        u_o, y_o, st_o = ob.quad_solve(problem, N, x0[i], u0[i])
        assert st["status"][i] == st_o["status"] == 1
        for key in ("outer_iterations", "iterations", "cg_iterations", "n_hess_psi_prod", "n_hess_psi_prod_calls", "n_psi",
                    "n_psi_grad_psi", "n_grad_L", "inner_convergence_failures"):
            assert st[key][i] == st_o[key], key
        assert rel(u[i], u_o) <= RTOL and np.array_equal(u[i], u_o) and np.array_equal(y[i], y_o)
        assert st["eps"][i] == st_o["eps"] and st["delta"][i] == st_o["delta"]
        assert st["eps"][i] <= 1e-8 and st["delta"][i] <= 1e-8


def test_longest_realistic_quadcopter_horizon_solves_like_the_oracle():
    """N = 110: 155 KiB of the CU's 160 KiB LDS (the two-wavefront kernel does not fit: one wavefront), two
    blocks of 64 stages in every scan; one cold ALM solve - thousands of PANTR iterations - bit for bit."""
    problem, N = "realistic_quadcopter", 110
    b = pa.PantrBatch(problem, N, 1, opts=quad_opts(problem))
    assert "spec" not in b.kernel_name()
    x0, u0 = ob.quad_initial_state()[None, :], ob.quad_initial_guess(N)[None, :]
    u, y, st = b.solve_al(x0, u0)
    b.close()
    u_o, y_o, st_o = ob.quad_solve(problem, N, x0[0], u0[0])
    assert st["status"][0] == st_o["status"] == 1
    for key in ("outer_iterations", "iterations", "cg_iterations", "n_hess_psi_prod", "n_hess_psi_prod_calls", "n_psi", "n_grad_L"):
        assert st[key][0] == st_o[key], key
    assert np.array_equal(u[0], u_o) and np.array_equal(y[0], y_o)


@pytest.mark.parametrize("spec", ["on", "off"])
def test_generic_passes_beyond_n_256_solve_like_the_oracle(rng, spec):
    """n = 280 (quadcopter, N = 70): more than four elements per lane, so both ALM kernels of method=pantr run their
    generic element-wise passes (last template argument 0; up to n = 256 the passes are straight-line code on four
    elements per lane, which every other solve test of this file exercises)."""
    problem, N, B = "quadcopter", 70, 2
    b = pa.PantrBatch(problem, N, B, opts=quad_opts(problem) + [f"speculation={spec}"])
    assert b.kernel_name() == ("pantr_alm_spec_kernel<QuadModelT<false>, 0, 0>" if spec == "on" else "pantr_alm_kernel<QuadModelT<false>, 0, 0, 0>")
    x0 = np.vstack([ob.quad_initial_state(), ob.quad_initial_state() + 0.02 * rng.standard_normal((B - 1, 9))])
    u0 = np.tile(ob.quad_initial_guess(N), (B, 1))
    u, y, st = b.solve_al(x0, u0)
    b.close()
    for i in range(B):
        u_o, y_o, st_o = ob.quad_solve(problem, N, x0[i], u0[i])
        assert st["status"][i] == st_o["status"] == 1
        for key in ("outer_iterations", "iterations", "cg_iterations", "n_hess_psi_prod", "n_hess_psi_prod_calls", "n_psi", "n_grad_L"):
            assert st[key][i] == st_o[key], key
        assert np.array_equal(u[i], u_o) and np.array_equal(y[i], y_o)


@pytest.mark.parametrize("problem", ["quadcopter", "realistic_quadcopter"])
@pytest.mark.parametrize("N", [1, 2])
def test_smallest_horizons_solve_like_the_oracle(rng, problem, N):
    """One- and two-stage horizons through the ALM kernel `speculation=auto` selects (two wavefronts)."""
    B = 2
    b = pa.PantrBatch(problem, N, B, opts=quad_opts(problem))
    x0 = np.vstack([ob.quad_initial_state(), ob.quad_initial_state() + 0.02 * rng.standard_normal((B - 1, 9))])
    u0 = np.tile(ob.quad_initial_guess(N), (B, 1))
    u, y, st = b.solve_al(x0, u0)
    b.close()
    for i in range(B):
        u_o, y_o, st_o = ob.quad_solve(problem, N, x0[i], u0[i])
        assert st["status"][i] == st_o["status"]
        for key in ("outer_iterations", "iterations", "cg_iterations", "n_hess_psi_prod", "n_hess_psi_prod_calls", "n_psi", "n_grad_L"):
            assert st[key][i] == st_o[key], key
        assert np.array_equal(u[i], u_o) and np.array_equal(y[i], y_o)


@pytest.mark.parametrize("problem,N", [("quadcopter", 12), ("realistic_quadcopter", 20), ("realistic_quadcopter", 60)])
def test_speculation_changes_nothing_but_time(rng, problem, N):
    """Two wavefronts per instance (pantr_alm_spec_kernel: the helper evaluates the successor of a rejected
    candidate during the CG loop and the candidate's evaluation) against one: solutions, multipliers and
    every counter equal.  The other tests of this file run whatever `speculation=auto` selects - the
    two-wavefront kernel at these batch sizes - against the oracle."""
    B = 4
    x0 = np.vstack([ob.quad_initial_state(), ob.quad_initial_state() + 0.02 * rng.standard_normal((B - 1, 9))])
    u0 = np.tile(ob.quad_initial_guess(N), (B, 1))
    res = {}
    for mode in ("on", "off", "late", "midway"):
        # "midway": a patience that runs out in the middle of a solve, while the helper is inside a sweep that writes
        # yhat to the prox iterate's slot - the solver goes on only once the helper has said that it ended
        extra = {"late": ["speculation=on", "spec_patience=0"], "midway": ["speculation=on", "spec_patience=2us"]}.get(
            mode, [f"speculation={mode}"])
        b = pa.PantrBatch(problem, N, B, opts=quad_opts(problem) + extra)
        assert ("spec" in b.kernel_name()) == (mode != "off")
        res[mode] = b.solve_al(x0, u0)
        b.close()
    (u1, y1, s1), (u0_, y0_, s0) = res["on"], res["off"]
    assert np.array_equal(u1, u0_) and np.array_equal(y1, y0_)
    for key in s1.dtype.names:
        if key not in ("elapsed_s", "phase_s"):
            assert np.array_equal(s1[key], s0[key]), key
    # a helper declared late (spec_patience=0: the first result asked for): the solver stops waiting, tells it to
    # stop and goes on alone - same results, the timeout reported, the launch returns
    u2, y2, s2 = res["late"]
    assert np.array_equal(u2, u0_) and np.array_equal(y2, y0_)
    assert np.all(s2["spec_timeouts"] >= 1) and np.all(s2["spec_timeouts"] <= 2) and np.all(s1["spec_timeouts"] == 0)
    for key in ("status", "iterations", "outer_iterations", "cg_iterations", "n_hess_psi_prod", "n_psi", "n_grad_L", "eps"):
        assert np.array_equal(s2[key], s0[key]), key
    u3, y3, s3 = res["midway"]
    assert np.array_equal(u3, u0_) and np.array_equal(y3, y0_)
    assert N < 60 or s3["spec_timeouts"].sum() >= 1    # (at N = 60 a helper's sweeps take tens of microseconds)
    for key in ("status", "iterations", "outer_iterations", "cg_iterations", "n_hess_psi_prod", "n_psi", "n_grad_L", "eps", "delta"):
        assert np.array_equal(s3[key], s0[key]), key
    assert np.all(s1["status"] == 1) and s1["rejected"].sum() > 0   # (rejections happened: results were taken)
    b = pa.PantrBatch(problem, N, 4096, opts=quad_opts(problem))
    assert "spec" not in b.kernel_name()                             # auto: large batches keep one wavefront
    assert "spec" in b.kernel_name(B)
    b.close()


@pytest.mark.parametrize("warm", [True, False])
def test_closed_loop_with_multipliers(warm):
    """mpc-driver.cpp:123-210 incl. the multiplier shift (:173-175), on the device."""
    problem, N, num_sim = "quadcopter", 10, 5
    b = pa.PantrBatch(problem, N, 1, opts=quad_opts(problem))
    states, inputs, stats = b.mpc(num_sim, warm, np.zeros((1, 4)), num_dist=0)
    b.close()
    st_o, in_o, stats_o = ob.quad_mpc(problem, N, num_sim, warm)
    assert np.array_equal(states[:, 0], st_o) and np.array_equal(inputs[:, 0], in_o)
    assert [s["iterations"] for s in stats_o] == stats["iterations"][:, 0].tolist()


def test_config3_realistic_quadcopter_horizon_60():
    """BASELINE.json configs[2] at its stated size: realistic_quadcopter N=60 (n=240, m=244, 64 KB of
    LDS per instance), all 60 warm + 60 cold time steps (dodo.py:10,70,96-105).  The device closed
    loops equal the oracle's step for step, and the 120 recorded steps replayed as ONE batch (the
    form bench.py measures) reproduce u, y and every counter of the oracle."""
    from concurrent.futures import ThreadPoolExecutor
    problem, N, num_sim = "realistic_quadcopter", 60, 60
    keys = ("iterations", "outer_iterations", "cg_iterations", "n_hess_psi_prod", "n_hess_psi_prod_calls", "n_psi", "n_psi_grad_psi",
            "n_grad_L", "inner_convergence_failures")
    with ThreadPoolExecutor(2) as ex:   # the oracle's two closed loops run beside the device's
        fut = {w: ex.submit(ob.quad_mpc_full, problem, N, num_sim, w) for w in (True, False)}
        b = pa.PantrBatch(problem, N, 2 * num_sim, opts=quad_opts(problem))
        assert (b.n, b.m) == (240, 244) and b.footprint()[0] > 64 * 1024
        dev = {w: b.mpc(num_sim, w, np.zeros((1, 4)), num_dist=0) for w in (True, False)}
        orc = {w: fut[w].result() for w in (True, False)}
    for warm in (True, False):
        states, inputs, stats = dev[warm]
        o = orc[warm]
        assert np.array_equal(states[:, 0], o["states"]) and np.array_equal(inputs[:, 0], o["inputs"])
        assert np.all(stats["status"][:, 0] == 1) and all(s["status"] == 1 for s in o["stats"])
        for k in keys:
            assert stats[k][:, 0].tolist() == [s[k] for s in o["stats"]], (warm, k)
        assert np.all(stats["eps"] <= 1e-8) and np.all(stats["delta"] <= 1e-8)
    cat = lambda k: np.vstack([orc[True][k], orc[False][k]])
    u, y, st = b.solve_al(cat("states")[np.r_[0:num_sim, num_sim + 1:2 * num_sim + 1]], cat("guess_u"), cat("guess_y"))
    b.close()
    assert np.array_equal(u, cat("inputs")) and np.array_equal(y, cat("multipliers"))
    stats_o = orc[True]["stats"] + orc[False]["stats"]
    for k in keys + ("status",):
        assert st[k].tolist() == [s[k] for s in stats_o], k
    assert st["eps"].tolist() == [s["eps"] for s in stats_o] and st["delta"].tolist() == [s["delta"] for s in stats_o]


def test_config1_quadcopter_horizon_12_at_its_bench_size():
    """BASELINE.json configs[0] at the size bench.py times it: quadcopter N = 12 (n = 48, m = 52), the 60 warm
    closed-loop steps of `benchmark-mpc-driver quadcopter 12 ss2p method=pantr num_sim=60 warm=true` with the options
    of dodo.py:12-33,97-100.  The device closed loop equals the oracle's step for step - states, solutions,
    multipliers, every counter - and the 60 recorded steps replayed as ONE batch (what the bench measures) reproduce
    the oracle's u, y, eps, delta and counters."""
    problem, N, num_sim = "quadcopter", 12, 60
    keys = ("status", "iterations", "outer_iterations", "cg_iterations", "n_hess_psi_prod", "n_hess_psi_prod_calls", "n_psi",
            "n_psi_grad_psi", "n_grad_L", "n_prox_grad_step", "accepted", "rejected", "qub_backtracks",
            "inner_convergence_failures")
    o = ob.quad_mpc_full(problem, N, num_sim, True)
    b = pa.PantrBatch(problem, N, num_sim, opts=quad_opts(problem))
    assert (b.n, b.m) == (48, 52)
    states, inputs, stats, rec = b.mpc(num_sim, True, np.zeros((1, 4)), num_dist=0, record=True)
    assert np.array_equal(states[:, 0], o["states"]) and np.array_equal(inputs[:, 0], o["inputs"])
    assert np.array_equal(rec["guess_u"][:, 0], o["guess_u"]) and np.array_equal(rec["guess_y"][:, 0], o["guess_y"])
    assert np.array_equal(rec["multipliers"][:, 0], o["multipliers"])
    assert np.all(stats["status"][:, 0] == 1) and np.all(stats["eps"] <= 1e-8) and np.all(stats["delta"] <= 1e-8)
    for k in keys:
        assert stats[k][:, 0].tolist() == [s[k] for s in o["stats"]], k
    # the replay: all 60 steps in one launch (the short-horizon kernel, two wavefronts per instance)
    assert b.kernel_name(num_sim) == "pantr_alm_spec_kernel<QuadModelT<false>, 1, 1>"
    u, y, st = b.solve_al(o["states"][:num_sim], o["guess_u"], o["guess_y"])
    b.close()
    assert np.array_equal(u, o["inputs"]) and np.array_equal(y, o["multipliers"])
    for k in keys:
        assert st[k].tolist() == [s[k] for s in o["stats"]], k
    assert st["eps"].tolist() == [s["eps"] for s in o["stats"]] and st["delta"].tolist() == [s["delta"] for s in o["stats"]]


@pytest.mark.parametrize("warm", [True, False])
def test_closed_loop_records_what_a_replay_needs(warm):
    """pantr_batch_mpc_record: the guesses every step was solved from and the multipliers it returned equal
    the oracle's closed loop (mpc-driver.cpp:168-175,192-197), and solving the recorded steps again as
    one batch reproduces inputs and multipliers (the replay batching bench.py measures)."""
    problem, N, num_sim = "quadcopter", 12, 8
    b = pa.PantrBatch(problem, N, num_sim, opts=quad_opts(problem))
    states, inputs, stats, rec = b.mpc(num_sim, warm, np.zeros((1, 4)), num_dist=0, record=True)
    o = ob.quad_mpc_full(problem, N, num_sim, warm)
    assert np.array_equal(rec["guess_u"][:, 0], o["guess_u"]) and np.array_equal(rec["guess_y"][:, 0], o["guess_y"])
    assert np.array_equal(rec["multipliers"][:, 0], o["multipliers"]) and np.array_equal(inputs[:, 0], o["inputs"])
    u, y, st = b.solve_al(states[:num_sim, 0], rec["guess_u"][:, 0], rec["guess_y"][:, 0])
    b.close()
    assert np.array_equal(u, inputs[:, 0]) and np.array_equal(y, rec["multipliers"][:, 0])
    assert st["iterations"].tolist() == stats["iterations"][:, 0].tolist()


@pytest.mark.parametrize("problem", ["quadcopter", "realistic_quadcopter"])
def test_non_finite_state_is_handled_like_the_oracle(problem):
    """A NaN in the initial state of one instance: that instance ends as the oracle's does (status, counters,
    whatever the solution slot holds), its neighbour in the batch is unaffected."""
    N = 12
    b = pa.PantrBatch(problem, N, 2, opts=quad_opts(problem))
    x0 = np.vstack([ob.quad_initial_state(), ob.quad_initial_state()])
    x0[1, 2] = np.nan
    u0 = np.tile(ob.quad_initial_guess(N), (2, 1)) + 0.01
    u, y, st = b.solve_al(x0, u0)
    b.close()
    for i in range(2):
        u_o, y_o, st_o = ob.quad_solve(problem, N, x0[i], u0[i])
        assert st["status"][i] == st_o["status"] and np.array_equal(u[i], u_o, equal_nan=True)
        assert st["outer_iterations"][i] == st_o["outer_iterations"] and st["iterations"][i] == st_o["iterations"]
    assert st["status"][0] == 1 and np.all(np.isfinite(u[0]))
