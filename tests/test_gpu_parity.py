"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle.

Bar: exact equality of fp64 results (the kernels and the oracle follow one arithmetic
specification), which is stricter than the 1e-10 relative tolerance the
north star states; RTOL below is that stated tolerance and is what the asserts enforce
where a comparison is not required to be exact.
"""
import os

import numpy as np
import pytest

import oracle_binding as ob
import pantr_amd as pa
from common import NX, NU, disturbed_state, random_inputs, random_states

pytestmark = pytest.mark.gpu
RTOL = 1e-10  # north star: fp64 iterates within 1e-10 relative of the CPU solver


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(20231003)


# 1..5, 7: every way the stage loops of the products end (landing buffers in rotation, first and last stages by
# themselves); 682: the longest horizon a handle accepts (n = 2046)
@pytest.mark.parametrize("N", [1, 2, 3, 4, 5, 7, 21, 22, 60, 682])
def test_psi_grad_hess_prod_match_oracle(rng, N):
    B = 5 if N < 600 else 2
    with_batch = pa.PantrBatch("hanging_chain", N, B)
    x0, u, v = random_states(rng, B), random_inputs(rng, B, N), rng.standard_normal((B, NU * N))
    psi = with_batch.eval_psi(x0, u)
    psi2, grad = with_batch.eval_psi_grad_psi(x0, u)
    Hv = with_batch.eval_hess_psi_prod(x0, u, v, prefetch_depth=1)  # tape record + replay, the solver's path
    # ... and with the tape rows requested two stages ahead (explicit vmcnt waits, three landing buffers, the
    # tangent-state row by LDS-DMA: what the one-wavefront solve kernel runs) - the same bits
    Hv2 = with_batch.eval_hess_psi_prod(x0, u, v, prefetch_depth=2)
    assert np.array_equal(Hv2, Hv)
    # ... and in the form the one-wavefront kernel of large batches runs: the tangent state of every stage kept instead
    # of its tangent spring vectors, the stage's forward tangent run again inside the adjoint sweep (3 = 2 + that)
    Hv3 = with_batch.eval_hess_psi_prod(x0, u, v, prefetch_depth=3)
    assert np.array_equal(Hv3, Hv)
    for b in range(B):
        f_o, g_o = ob.psi_grad(N, x0[b], u[b])
        Hv_o = ob.hess_prod(N, x0[b], u[b], v[b])
        assert psi[b] == f_o and psi2[b] == f_o
        assert rel(grad[b], g_o) <= RTOL and np.array_equal(grad[b], g_o)
        assert rel(Hv[b], Hv_o) <= RTOL and np.array_equal(Hv[b], Hv_o)
        # and the independent dual-number product of the oracle agrees to rounding
        assert rel(Hv[b], ob.hess_prod(N, x0[b], u[b], v[b], dual=True)) <= (1e-12 if N < 600 else 1e-9)
    with_batch.close()


@pytest.mark.parametrize("N", [1, 22, 60])
def test_prox_grad_step_matches_oracle(rng, N):
    B = 4
    batch = pa.PantrBatch("hanging_chain", N, B)
    x = random_inputs(rng, B, N, -1.5, 1.5)          # some components outside the box
    g = 30 * rng.standard_normal((B, NU * N))
    for gamma in (1.0, 1e-3, 0.05):
        xh, p, s = batch.eval_prox_grad_step(gamma, x, g)
        for b in range(B):
            xh_o, p_o, pTp, gTp = ob.prox_grad_step(N, gamma, x[b], g[b])
            assert np.array_equal(xh[b], xh_o) and np.array_equal(p[b], p_o)
            assert s[b, 0] == pTp and s[b, 1] == gTp
            lb, ub = -1.0, 1.0
            assert s[b, 2] == np.max(np.abs(np.clip(-g[b], lb - x[b], ub - x[b])))
            xfw = x[b] - gamma * g[b]
            assert s[b, 3] == np.sum((lb < xfw) & (xfw < ub))
    batch.close()


def test_simulate_dynamics_matches_oracle(rng):
    B = 9
    batch = pa.PantrBatch("hanging_chain", 3, B)
    x, u = random_states(rng, B), rng.uniform(-1, 1, (B, NU))
    xn = batch.simulate_dynamics(x, u)
    for b in range(B):
        assert np.array_equal(xn[b], ob.simulate(x[b], u[b]))
    batch.close()


def _solve_both(N, x0s, u0s, opts, trace_cap=400):
    B = len(x0s)
    batch = pa.PantrBatch("hanging_chain", N, B, opts=opts)
    batch.set_trace(B, trace_cap)
    u, st = batch.solve(x0s, u0s)
    tr = batch.get_trace()
    batch.close()
    return u, st, tr


@pytest.mark.parametrize("N", [5, 22, 60])
def test_solve_iterates_match_oracle(rng, N):
    """Whole solves: same iterates (per-iteration trace), same counters, same solution."""
    B = 4
    x0s = np.vstack([disturbed_state()[None, :], random_states(rng, B - 1, 0.02)])
    u0s = np.vstack([np.zeros((1, NU * N)), random_inputs(rng, B - 1, N, -0.3, 0.3)])
    u, st, tr = _solve_both(N, x0s, u0s, pa.DODO_PANTR_OPTS)
    prm = ob.dodo_params()
    for b in range(B):
        u_o, st_o, tr_o = ob.solve(N, x0s[b], u0s[b], prm, trace_cap=400)
        assert st["status"][b] == st_o["status"] == 1
        for key in ("iterations", "cg_iterations", "accepted", "rejected", "qub_backtracks",
                    "n_psi", "n_grad_psi", "n_psi_grad_psi", "n_grad_L", "n_hess_psi_prod", "n_hess_psi_prod_calls",
                    "n_prox_grad_step"):
            assert st[key][b] == st_o[key], key
        k = st_o["iterations"]
        assert len(tr_o) == k
        a, o = tr[b, :k], tr_o
        fin = np.isfinite(o)
        assert np.array_equal(np.isfinite(a), fin)
        assert rel(a[fin], o[fin]) <= RTOL and np.array_equal(a[fin], o[fin])
        assert rel(u[b], u_o) <= RTOL and np.array_equal(u[b], u_o)
        assert st["eps"][b] == st_o["eps"] and st["eps"][b] <= 1e-8
        assert st["final_psi"][b] == st_o["final_psi"]


@pytest.mark.parametrize("N", [1, 2, 3])
@pytest.mark.parametrize("spec", ["on", "off"])
def test_smallest_horizons_solve_like_the_oracle(rng, N, spec):
    """The stage loops are unrolled by two and prefetch a stage ahead: horizons of one, two and three stages
    through both solve kernels (one and two wavefronts per instance)."""
    B = 3
    x0s = np.vstack([disturbed_state()[None, :], random_states(rng, B - 1, 0.02)])
    u0s = np.vstack([np.zeros((1, NU * N)), random_inputs(rng, B - 1, N, -0.3, 0.3)])
    b = pa.PantrBatch("hanging_chain", N, B, opts=tuple(pa.DODO_PANTR_OPTS) + (f"speculation={spec}",))
    assert ("spec" in b.kernel_name()) == (spec == "on")
    u, st = b.solve(x0s, u0s)
    b.close()
    prm = ob.dodo_params()
    for i in range(B):
        u_o, st_o, _ = ob.solve(N, x0s[i], u0s[i], prm)
        assert st["status"][i] == st_o["status"]
        for key in ("iterations", "cg_iterations", "accepted", "rejected", "n_psi", "n_psi_grad_psi", "n_grad_L",
                    "n_hess_psi_prod", "n_hess_psi_prod_calls"):
            assert st[key][i] == st_o[key], key
        assert np.array_equal(u[i], u_o) and st["eps"][i] == st_o["eps"]


def test_longest_horizon_through_the_solve_kernel(rng):
    """n = 2046 (N = 682: 32 elements per lane, 112 KiB of CG vectors in LDS): two PANTR iterations, like the oracle's."""
    N = 682
    x0s, u0s = disturbed_state()[None, :], np.zeros((1, NU * N))
    b = pa.PantrBatch("hanging_chain", N, 1, opts=tuple(pa.DODO_PANTR_OPTS) + ("solver.max_iter=2",))
    assert b.kernel_name() == "pantr_solve_kernel<ChainModel, 0, 0>"   # (the two-wavefront kernel's LDS does not fit; generic passes)
    u, st = b.solve(x0s, u0s)
    b.close()
    u_o, st_o, _ = ob.solve(N, x0s[0], u0s[0], ob.dodo_params(max_iter=2))
    assert st["status"][0] == st_o["status"] == 3 and st["iterations"][0] == st_o["iterations"] == 2
    for key in ("cg_iterations", "n_hess_psi_prod", "n_hess_psi_prod_calls", "n_psi", "n_psi_grad_psi", "n_grad_L"):
        assert st[key][0] == st_o[key], key
    assert np.array_equal(u[0], u_o) and st["eps"][0] == st_o["eps"]


def test_sizes_outside_the_limits_are_refused():
    """n = nu N <= 2048 (32 elements per lane); 1 <= B <= max_instances."""
    with pytest.raises(pa.PantrError, match="horizon too long"):
        pa.PantrBatch("hanging_chain", 683, 1)                  # n = 2049
    b = pa.PantrBatch("hanging_chain", 5, 2)
    for B in (0, 3):
        with pytest.raises(pa.PantrError, match="batch size"):
            b.solve(np.zeros((B, NX)), np.zeros((B, 15)))
    b.close()


def test_solve_status_codes(rng):
    """max_iter exhaustion gives MaxIter (negative runtime in the reference, mpc-driver.cpp:219-221)."""
    N = 22
    x0s, u0s = disturbed_state()[None, :], np.zeros((1, NU * N))
    u, st, _ = _solve_both(N, x0s, u0s, pa.DODO_PANTR_OPTS + ("solver.max_iter=3",))
    prm = ob.dodo_params(max_iter=3)
    u_o, st_o, _ = ob.solve(N, x0s[0], u0s[0], prm)
    assert st["status"][0] == st_o["status"] == 3 and st["iterations"][0] == 3
    assert np.array_equal(u[0], u_o)  # always_overwrite_results in the m = 0 ALM shell


def test_unsupported_option_fails_loudly():
    batch = pa.PantrBatch("hanging_chain", 5, 1, opts=("dir.finite_diff=true",))
    with pytest.raises(pa.PantrError):
        batch.solve(np.zeros((1, NX)), np.zeros((1, 15)))
    batch.close()


@pytest.mark.parametrize("warm", [True, False])
def test_closed_loops_in_lock_step_match_oracle(rng, warm):
    """B trajectories advanced on the device (solve, plant step, shift) == B oracle closed loops."""
    N, B, num_sim = 15, 6, 7
    u_dist = rng.uniform(-1, 1, (B, NU))
    u_dist[0] = (-1.0, 1.0, 1.0)                       # dodo.py:94-95
    batch = pa.PantrBatch("hanging_chain", N, B, opts=pa.DODO_PANTR_OPTS)
    states, inputs, stats = batch.mpc(num_sim, warm, u_dist, num_dist=5)
    batch.close()
    for b in range(B):
        st_o, in_o, _, stats_o = ob.mpc(N, num_sim, warm, num_dist=5, u_dist=u_dist[b])
        assert np.array_equal(states[:, b, :], st_o) and np.array_equal(inputs[:, b, :], in_o)
        assert [s["iterations"] for s in stats_o] == stats["iterations"][:, b].tolist()
        assert np.all(stats["status"][:, b] == 1)


def test_config5_stress_horizon_120():
    """BASELINE.json configs[4], one GPU's shard: N=120 (n=360), 128 of the 1024 synthetic initial
    states x0_j = f_d^5(x_rest, d_j), d_j from mt19937_64(12345).  All 128 are
    certified (status, eps, box), 16 are compared with the oracle bit for bit."""
    from concurrent.futures import ThreadPoolExecutor
    from pantr_amd import workloads
    N, B = 120, 128
    d = workloads.config5_disturbances(1024)[:B]
    batch = pa.PantrBatch("hanging_chain", N, B, opts=pa.DODO_PANTR_OPTS)
    x0 = np.tile(batch.initial_state(), (B, 1))
    for _ in range(5):
        x0 = batch.simulate_dynamics(x0, d)
    u, st = batch.solve(x0, np.zeros((B, NU * N)))
    lds, ws = batch.footprint()
    assert lds <= 160 * 1024 and np.all(st["status"] == 1) and np.all(st["eps"] <= 1e-8)
    assert np.all(np.abs(u) <= 1.0)
    _, grad = batch.eval_psi_grad_psi(x0, u)           # ProjGradUnitNorm of the returned points
    assert np.max(np.abs(np.clip(u - grad, -1, 1) - u)) <= 1e-8
    batch.close()
    pick = list(range(0, B, 8))

    def oracle(b):
        xo = ob.initial_state()
        for _ in range(5):
            xo = ob.simulate(xo, d[b])
        return (xo,) + ob.solve(N, xo, np.zeros(NU * N))[:2]
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        res = list(ex.map(oracle, pick))
    for b, (xo, u_o, st_o) in zip(pick, res):
        assert np.array_equal(xo, x0[b]) and np.array_equal(u[b], u_o)
        for key in ("iterations", "cg_iterations", "n_hess_psi_prod", "n_hess_psi_prod_calls", "n_psi", "n_psi_grad_psi", "n_grad_L"):
            assert st[key][b] == st_o[key], key


def test_config5_full_launch_through_the_one_wavefront_kernel():
    """BASELINE.json configs[4] as bench.py launches it: all 1024 instances of N=120 in ONE launch, which
    `speculation=auto` sends to pantr_solve_kernel<ChainModel, 0, 6> (one wavefront per instance, tape rows requested
    two stages ahead, four instances per compute unit - the memory system under load).  All 1024 are certified;
    32 of them, spread over the launch, equal the oracle's solve bit for bit, counters included; a second launch
    reproduces the first."""
    from concurrent.futures import ThreadPoolExecutor
    from pantr_amd import workloads
    w = workloads.stress_batch(1024)
    N, B = 120, 1024
    batch = pa.PantrBatch("hanging_chain", N, B, opts=w["opts"])
    assert batch.kernel_name(B) == "pantr_solve_kernel<ChainModel, 0, 6>"   # (n = 360: six elements per lane at compile time)
    u, st = batch.solve(w["x0"], w["u0"])
    u2, st2 = batch.solve(w["x0"], w["u0"])
    batch.close()
    assert np.all(st["status"] == 1) and np.all(st["eps"] <= 1e-8) and np.all(np.abs(u) <= 1.0)
    assert np.array_equal(u, u2) and np.array_equal(st["n_hess_psi_prod"], st2["n_hess_psi_prod"])
    pick = list(range(5, B, 32))
    assert len(pick) == 32
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        res = list(ex.map(lambda b: ob.solve(N, w["x0"][b], w["u0"][b])[:2], pick))
    for b, (u_o, st_o) in zip(pick, res):
        assert np.array_equal(u[b], u_o), b
        for key in ("status", "iterations", "cg_iterations", "accepted", "rejected", "n_hess_psi_prod", "n_hess_psi_prod_calls",
                    "n_psi", "n_psi_grad_psi", "n_grad_L"):
            assert st[key][b] == st_o[key], (b, key)
        assert st["eps"][b] == st_o["eps"] and st["final_psi"][b] == st_o["final_psi"]


def test_config4_horizon_sweep():
    """BASELINE.json configs[3] at its stated size: hanging_chain N in {10,...,60} x {cold, warm} x 60
    timed MPC steps = 720 instances (dodo.py:10,69-95).  Every device closed loop equals the
    oracle's, and each horizon's 120 steps replayed as one batch reproduce the solutions and
    counters (one launch per horizon: the form bench.py measures)."""
    from concurrent.futures import ThreadPoolExecutor
    from pantr_amd import workloads
    num_sim = workloads.NUM_SIM
    combos = [(N, warm) for N in workloads.HORIZONS for warm in (True, False)]
    with ThreadPoolExecutor(min(12, os.cpu_count() or 1)) as ex:
        fut = {c: ex.submit(ob.mpc, c[0], num_sim, c[1]) for c in combos}
        dev = {}
        for N in workloads.HORIZONS:
            batch = pa.PantrBatch("hanging_chain", N, 1, opts=pa.DODO_PANTR_OPTS)
            for warm in (True, False):
                dev[(N, warm)] = batch.mpc(num_sim, warm, np.array([[-1.0, 1.0, 1.0]]), num_dist=5)
            batch.close()
        orc = {c: fut[c].result() for c in combos}
    total = 0
    for N in workloads.HORIZONS:
        x0s, guesses, sols, stats_o = [], [], [], []
        for warm in (True, False):
            states, inputs, stats = dev[(N, warm)]
            st_o, in_o, gs_o, so = orc[(N, warm)]
            assert np.array_equal(states[:, 0], st_o) and np.array_equal(inputs[:, 0], in_o), (N, warm)
            assert np.all(stats["status"][:, 0] == 1) and np.all(stats["eps"] <= 1e-8)
            assert stats["iterations"][:, 0].tolist() == [s["iterations"] for s in so]
            x0s.append(st_o[:num_sim]); guesses.append(gs_o); sols.append(in_o); stats_o += so
        batch = pa.PantrBatch("hanging_chain", N, 2 * num_sim, opts=pa.DODO_PANTR_OPTS)
        u, st = batch.solve(np.vstack(x0s), np.vstack(guesses))
        batch.close()
        assert np.array_equal(u, np.vstack(sols)), N
        for key in ("iterations", "cg_iterations", "n_hess_psi_prod", "n_hess_psi_prod_calls", "n_psi", "n_psi_grad_psi", "n_grad_L", "status"):
            assert st[key].tolist() == [s[key] for s in stats_o], (N, key)
        total += len(u)
    assert total == 720


def test_bench_workload_matches_oracle_at_full_occupancy():
    """The 600-instance replay batch bench.py times (BASELINE.json configs[1], warm + cold), solved
    in one launch with every SIMD of the chip busy: each instance equals the oracle's solve, and a
    second launch reproduces the first (no dependence on scheduling)."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpc_hanging_chain_N60_warm300.npz"))
    x0 = np.vstack([d["x0"], d["x0"]])
    u0 = np.vstack([d["guess"], np.zeros_like(d["guess"])])
    batch = pa.PantrBatch("hanging_chain", 60, len(x0), opts=pa.DODO_PANTR_OPTS)
    u, st = batch.solve(x0, u0)
    u2, st2 = batch.solve(x0, u0)
    batch.close()
    assert np.array_equal(u, u2) and np.array_equal(st["n_hess_psi_prod"], st2["n_hess_psi_prod"])
    assert np.array_equal(u[:300], d["sol"])  # the recorded closed loop (committed fixture)
    u_o, st_o = ob.solve_batch(60, x0, u0, ob.dodo_params(), nthreads=min(16, os.cpu_count() or 1))
    assert np.array_equal(u, u_o)
    for key in ("iterations", "n_hess_psi_prod", "n_hess_psi_prod_calls", "cg_iterations", "status"):
        assert np.array_equal(st[key], np.array([s[key] for s in st_o])), key



def test_problem_created_from_sidecar_files(tmp_path, rng):
    """load_benchmark_casadi_problem's data half (casadi-dll-wrapper.cpp:25-67): files in the layout of
    generate_problem.py:62-73.  Unmodified files give the compiled-in problem bit for bit; a tighter
    box C, another initial state and another initial guess are honoured by the solver."""
    N, B = 20, 3
    d = str(tmp_path)
    pa.write_problem_files(d, "hanging_chain", "ss2p", N)
    x0, u0 = random_states(rng, B, 0.02), random_inputs(rng, B, N, -0.3, 0.3)
    ref = pa.PantrBatch("hanging_chain", N, B, opts=pa.DODO_PANTR_OPTS)
    same = pa.PantrBatch(horizon=N, max_instances=B, opts=pa.DODO_PANTR_OPTS,
                         files=(os.path.join(d, "hanging_chain"), "ss2p"))
    u_ref, st_ref = ref.solve(x0, u0)
    u_same, st_same = same.solve(x0, u0)
    assert np.array_equal(u_ref, u_same) and np.array_equal(st_ref["iterations"], st_same["iterations"])
    assert same.penalty_alm_split() == 0 and ref.penalty_alm_split() == 0   # m = 0 (generate_problem.py:32)
    same.close()
    bound = np.tile([0.4, 0.25, 0.6], N)
    guess = np.tile([0.1, -0.1, 0.05], N)
    param = disturbed_state(3)
    pa.write_problem_files(d, "hanging_chain", "ss2p", N, dict(C_lb=-bound, C_ub=bound, guess=guess, param=param))
    tight = pa.PantrBatch(horizon=N, max_instances=B, opts=pa.DODO_PANTR_OPTS,
                          files=(os.path.join(d, "hanging_chain"), "ss2p"))
    lb, ub = tight.box_C()
    assert np.array_equal(ub, bound) and np.array_equal(lb, -bound)
    assert np.array_equal(tight.initial_guess(), guess) and np.array_equal(tight.initial_state(), param)
    u_t, st_t = tight.solve(x0, u0)
    assert np.all(st_t["status"] == 1) and np.all(st_t["eps"] <= 1e-8)
    assert np.all(np.abs(u_t) <= bound[None, :]) and np.any(np.abs(u_t) == bound[None, :])   # the new box binds
    assert np.any(np.abs(u_ref) > bound[None, :])                                           # and the old one did not
    # optimality in the new box: ||Pi_C(u - grad) - u||_inf <= tolerance
    _, grad = tight.eval_psi_grad_psi(x0, u_t)
    assert np.max(np.abs(np.clip(u_t - grad, -bound, bound) - u_t)) <= 1e-8
    # the closed loop starts from the file's state and resets to the file's guess (cold)
    states, inputs, stats = tight.mpc(2, False, np.zeros((1, NU)), num_dist=0)
    assert np.array_equal(states[0, 0], ref.simulate_dynamics(param[None], inputs_first(tight, param, guess))[0])
    tight.close()
    ref.close()


def inputs_first(batch, x0, guess):
    """First input block of the solve from (x0, guess): what the closed loop applies before step 0."""
    u, st = batch.solve(x0[None], guess[None])
    assert st["status"][0] == 1
    return u[:, :NU]


def test_device_copy_bandwidth_is_plausible():
    """The measured HBM yardstick bench.py reports next to the 8 TB/s spec peak (16 B/lane device copy)."""
    gbps = pa.copy_bandwidth(0, nbytes=1 << 30, reps=3)
    assert 1000.0 < gbps < 8000.0, gbps
    with pytest.raises(pa.PantrError):
        pa.copy_bandwidth(0, nbytes=1024, reps=1)


def test_solve_device_refuses_pointers_it_cannot_see():
    """pantr_batch_solve_device checks every buffer with the HIP runtime the library is bound to: host
    memory (or memory of another runtime, the load-order trap of INTEGRATION.md section 3) is an error,
    not a fault."""
    b = pa.PantrBatch("hanging_chain", 5, 1, opts=pa.DODO_PANTR_OPTS)
    host = np.zeros(64)
    with pytest.raises(pa.PantrError, match="not device .or managed. memory"):
        b.solve_device(1, host.ctypes.data, host.ctypes.data, host.ctypes.data)
    # a refused set is not remembered as checked: the same pointers are refused again
    with pytest.raises(pa.PantrError, match="not device .or managed. memory"):
        b.solve_device(1, host.ctypes.data, host.ctypes.data, host.ctypes.data)
    b.close()


def test_lane_layout_of_the_stored_trajectory(rng):
    """chain_model.hip.h: lane = 32 chain + 10 component + point; both halves of the wavefront carry the full state, lanes
    30, 31 of a half are idle.  The stored trajectory rows ({x, v} per lane, 1 KiB per stage) show what the lanes carried
    through a forward sweep: the two halves hold the same bits in every stage, the idle lanes stay at zero (nothing is
    ever summed over them, but a non-finite value there could reach a neighbour through a masked fma), the active lanes
    are finite, and the stage-0 row is the initial state itself."""
    import ctypes as C
    N, B = 22, 3
    b = pa.PantrBatch("hanging_chain", N, B)
    x0, u = random_states(rng, B), random_inputs(rng, B, N)
    b.eval_psi_grad_psi(x0, u)
    rows = np.zeros((N + 1) * 128)
    for inst in range(B):
        rc = pa.lib().pantr_debug_read_workspace(b._h, inst, C.c_int64(0), C.c_int64(rows.size),
                                                 rows.ctypes.data_as(C.c_void_p))
        assert rc == 0
        lanes = rows.reshape(N + 1, 64, 2)
        assert np.array_equal(lanes[:, :32, :], lanes[:, 32:, :])
        assert np.all(lanes[:, [30, 31, 62, 63], :] == 0.0)
        active = np.array([10 * c + p for c in range(3) for p in range(10)])
        assert np.all(np.isfinite(lanes[:, active, :]))
        # lane 10c+p holds component c of point p
        assert np.array_equal(lanes[0, active, 0], x0[inst][[3 * p + c for c in range(3) for p in range(10)]])
    b.close()


def test_speculation_changes_nothing_but_time():
    """`speculation=on|off` (two wavefronts per instance: the successor of a rejected candidate is
    evaluated by the second one during the CG loop) - same solutions, same counters, bit for bit; the
    batch has accepted and rejected candidates, and quadratic-upper-bound backtracks the helper must
    decline."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpc_hanging_chain_N60_warm300.npz"))
    idx = np.arange(0, 300, 25)
    x0 = np.vstack([d["x0"][idx], d["x0"][idx]])
    u0 = np.vstack([d["guess"][idx], np.zeros_like(d["guess"][idx])])
    res = {}
    for mode in ("off", "on", "auto"):
        b = pa.PantrBatch("hanging_chain", 60, len(x0), opts=tuple(pa.DODO_PANTR_OPTS) + (f"speculation={mode}",))
        res[mode] = b.solve(x0, u0)
        b.close()
    u_off, st_off = res["off"]
    assert np.all(st_off["status"] == 1) and st_off["rejected"].sum() > 0 and st_off["accepted"].sum() > 0
    for mode in ("on", "auto"):
        u, st = res[mode]
        assert np.array_equal(u, u_off)
        for k in ("status", "iterations", "cg_iterations", "accepted", "rejected", "qub_backtracks", "n_psi",
                  "n_psi_grad_psi", "n_grad_L", "n_hess_psi_prod", "n_hess_psi_prod_calls", "n_prox_grad_step", "eps", "final_psi", "final_fbe",
                  "final_gamma", "final_radius"):
            assert np.array_equal(st[k], st_off[k]), (mode, k)
    with pytest.raises(pa.PantrError, match="speculation"):
        pa.PantrBatch("hanging_chain", 5, 1, opts=("speculation=maybe",))


def test_a_helper_that_misses_its_deadline_costs_time_only():
    """`spec_patience=0`: the first result the solver wavefront asks its helper for counts as late (normally the
    helper is done long before): the solver tells the helper to stop (SpecBox::abort), evaluates the step itself
    and posts no further requests; stats.spec_timeouts reports it.  Same solutions and counters as without a
    helper; the workgroup ends (the helper sees the flag), so the launch returns."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpc_hanging_chain_N60_warm300.npz"))
    idx = np.arange(0, 300, 30)
    x0 = np.vstack([d["x0"][idx], d["x0"][idx]])
    u0 = np.vstack([d["guess"][idx], np.zeros_like(d["guess"][idx])])
    res = {}
    for opts in (("speculation=off",), ("speculation=on",), ("speculation=on", "spec_patience=0")):
        b = pa.PantrBatch("hanging_chain", 60, len(x0), opts=tuple(pa.DODO_PANTR_OPTS) + opts)
        res[opts] = b.solve(x0, u0)
        b.close()
    u_off, st_off = res[("speculation=off",)]
    assert np.all(st_off["spec_timeouts"] == 0) and np.all(res[("speculation=on",)][1]["spec_timeouts"] == 0)
    u, st = res[("speculation=on", "spec_patience=0")]
    # the first wait of every instance that asks at all (and a second one if a request was still out); none after
    assert np.all(st["spec_timeouts"] <= 2) and np.sum(st["spec_timeouts"] >= 1) >= len(x0) - 2
    assert np.array_equal(u, u_off)
    for k in ("status", "iterations", "cg_iterations", "accepted", "rejected", "n_psi", "n_psi_grad_psi", "n_grad_L",
              "n_hess_psi_prod", "n_hess_psi_prod_calls", "eps", "final_psi"):
        assert np.array_equal(st[k], st_off[k]), k


def test_residency_caps_workgroups_per_cu_and_changes_nothing_else():
    """`residency=k` (at most k workgroups per compute unit, through the LDS the launch asks for): same
    solutions and counters, for the one- and the two-wavefront kernel."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpc_hanging_chain_N60_warm300.npz"))
    idx = np.arange(0, 300, 20)
    x0 = np.vstack([d["x0"][idx], d["x0"][idx]])
    u0 = np.vstack([d["guess"][idx], np.zeros_like(d["guess"][idx])])
    ref = None
    for opts in ((), ("residency=1",), ("residency=2", "speculation=off"), ("residency=5", "speculation=off")):
        b = pa.PantrBatch("hanging_chain", 60, len(x0), opts=tuple(pa.DODO_PANTR_OPTS) + opts)
        u, st = b.solve(x0, u0)
        b.close()
        if ref is None:
            ref = (u, st)
        assert np.array_equal(u, ref[0]) and np.array_equal(st["n_hess_psi_prod"], ref[1]["n_hess_psi_prod"]), opts
        assert np.all(st["status"] == 1)


def test_launch_order_from_the_solve_before_changes_nothing_but_time():
    """`launch_order=previous|auto` (the default is natural): behind every solve kernel the instances are ranked on the
    device, longest first by the time each took, and the next launch of that size starts them in that order.  Same
    solutions and counters whatever the order; auto does it only for launches with more workgroups than the chip holds
    at once; an order set with pantr_batch_set_launch_order wins; quadcopter family alike."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpc_hanging_chain_N60_warm300.npz"))
    idx = np.arange(0, 300, 4)
    x0 = np.vstack([d["x0"][idx], d["x0"][idx]])
    u0 = np.vstack([d["guess"][idx], np.zeros_like(d["guess"][idx])])
    B = len(x0)
    base = tuple(pa.DODO_PANTR_OPTS)
    for opts in (base, base + ("launch_order=natural",)):    # the default is natural
        b = pa.PantrBatch("hanging_chain", 60, B, opts=opts)
        b.solve(x0, u0)
        u_ref, st_ref = b.solve(x0, u0)
        assert b.last_launch_order(B) == ("index", None)
        b.close()
    b = pa.PantrBatch("hanging_chain", 60, B, opts=base + ("launch_order=previous",))
    u, st = b.solve(x0, u0)
    src, nxt = b.last_launch_order(B)
    assert src == "index"                                   # nothing to go by in the first launch
    assert sorted(nxt.tolist()) == list(range(B))           # a permutation ...
    t = st["elapsed_s"][nxt]
    assert np.all(t[:-1] >= t[1:])                          # ... longest first by the time each instance took
    assert np.array_equal(u, u_ref) and np.array_equal(st["n_hess_psi_prod"], st_ref["n_hess_psi_prod"])
    for _ in range(2):
        u, st = b.solve(x0, u0)
        assert b.last_launch_order()[0] == "previous solve"
        assert np.array_equal(u, u_ref) and np.array_equal(st["n_hess_psi_prod"], st_ref["n_hess_psi_prod"])
        assert np.array_equal(st["iterations"], st_ref["iterations"]) and np.all(st["status"] == 1)
    u, st = b.solve(x0[:40], u0[:40])                       # another batch size: the ranking does not apply
    assert b.last_launch_order()[0] == "index" and np.array_equal(u, u_ref[:40])
    b.set_launch_order(np.arange(B - 1, -1, -1))            # an order of the caller's wins
    u, st = b.solve(x0, u0)
    assert b.last_launch_order(B) == ("user", None) and np.array_equal(u, u_ref)
    b.set_launch_order(None)
    b.close()
    # auto: only where the launch has more workgroups than the chip holds at once
    b = pa.PantrBatch("hanging_chain", 60, B, opts=base + ("launch_order=auto",))
    b.solve(x0, u0); b.solve(x0, u0)
    assert b.last_launch_order(B) == ("index", None)
    b.close()
    R = 14                                                  # 2100 workgroups of one wavefront: six fit a compute unit
    X0, U0 = np.tile(x0, (R, 1)), np.tile(u0, (R, 1))
    b = pa.PantrBatch("hanging_chain", 60, R * B, opts=base + ("launch_order=auto",))
    u1, st1 = b.solve(X0, U0)
    assert b.last_launch_order()[0] == "index"
    u2, st2 = b.solve(X0, U0)
    assert b.last_launch_order()[0] == "previous solve"
    b.close()
    assert np.array_equal(u1, u2) and np.array_equal(u1, np.tile(u_ref, (R, 1)))
    assert np.array_equal(st1["n_hess_psi_prod"], st2["n_hess_psi_prod"])
    # quadcopter family (ALM kernels)
    from pantr_amd import workloads
    w = workloads.closed_loop_batch("quadcopter", 12, starts=(True,))
    opts = tuple(w["opts"])
    ref = None
    for extra in (("launch_order=natural",), ("launch_order=previous",)):
        b = pa.PantrBatch("quadcopter", 12, len(w["x0"]), opts=opts + extra)
        for k in range(2):
            u, y, st = b.solve_al(w["x0"], w["u0"], w["y0"])
            assert b.last_launch_order()[0] == ("previous solve" if extra[0].endswith("previous") and k else "index")
            if ref is None:
                ref = (u, y, st["iterations"].copy())
            assert np.array_equal(u, ref[0]) and np.array_equal(y, ref[1]) and np.array_equal(st["iterations"], ref[2])
        b.close()


def test_kernel_name_follows_method_speculation_and_batch_size():
    """pantr_batch_kernel_name: what measurement code names its dominant kernel with."""
    b = pa.PantrBatch("hanging_chain", 60, 1024, opts=pa.DODO_PANTR_OPTS)
    assert b.kernel_name(600) == "pantr_solve_spec_kernel<ChainModel, 3>"   # auto: small batches speculate (n = 180: three elements per lane)
    assert b.kernel_name(1024) == "pantr_solve_kernel<ChainModel, 0, 3>"    # auto: large ones do not
    lds_auto, ws = b.footprint()
    assert b.footprint(1024) == (lds_auto, ws) and b.footprint(600)[0] > lds_auto   # the figure follows the kernel of B
    b.forget_buffers()
    b.close()
    b = pa.PantrBatch("hanging_chain", 60, 600, opts=tuple(pa.DODO_PANTR_OPTS) + ("speculation=off",))
    assert b.kernel_name() == "pantr_solve_kernel<ChainModel, 0, 3>"
    lds_off, _ = b.footprint()
    b.close()
    b = pa.PantrBatch("hanging_chain", 60, 600, opts=tuple(pa.DODO_PANTR_OPTS) + ("speculation=on",))
    assert b.kernel_name() == "pantr_solve_spec_kernel<ChainModel, 3>"
    lds_on, _ = b.footprint()
    b.close()
    assert lds_off == lds_auto < lds_on <= 160 * 1024 // 3   # (1024 instances: one wavefront each; three two-wavefront workgroups fit a CU)
    b = pa.PantrBatch("hanging_chain", 120, 1024, opts=pa.DODO_PANTR_OPTS)
    assert b.kernel_name(256) == "pantr_solve_spec_kernel<ChainModel, 6>"   # auto: two per CU would fit up to 512 instances ...
    assert b.kernel_name(512) == "pantr_solve_kernel<ChainModel, 0, 6>"     # ... but 512 x 120 stages load the memory system
    b.close()
    # the elements of a decision vector a lane owns, nj = ceil(3 N / 64), are a compile-time constant of the kernel up to six
    for N, nj in ((10, 1), (21, 1), (22, 2), (30, 2), (50, 3), (70, 4), (100, 5), (128, 6), (129, 0), (200, 0)):
        b = pa.PantrBatch("hanging_chain", N, 4, opts=pa.DODO_PANTR_OPTS)
        assert b.kernel_name() == f"pantr_solve_spec_kernel<ChainModel, {nj}>", (N, b.kernel_name())
        b.close()
    b = pa.PantrBatch("hanging_chain", 60, 8, opts=pa.DODO_OPTS["panoc"])
    assert b.kernel_name() == "pantr_solve_kernel<ChainModel, 1, 0>"
    b.close()
    b = pa.PantrBatch("realistic_quadcopter", 12, 2)
    assert b.kernel_name() == "pantr_alm_spec_kernel<QuadModelT<true>, 4, 0>"       # auto, a small batch (n <= 256: four elements per lane)
    b.close()
    b = pa.PantrBatch("realistic_quadcopter", 12, 2, opts=tuple(pa.DODO_PANTR_OPTS) + ("speculation=off",))
    assert b.kernel_name() == "pantr_alm_kernel<QuadModelT<true>, 0, 4, 0>"
    b.close()
    b = pa.PantrBatch("quadcopter", 12, 2)
    assert b.kernel_name() == "pantr_alm_spec_kernel<QuadModelT<false>, 1, 1>"      # up to 16 stages: one element per lane, the product for short horizons
    b.close()
    b = pa.PantrBatch("quadcopter", 70, 2, opts=tuple(pa.DODO_PANTR_OPTS) + ("speculation=off",))
    assert b.kernel_name() == "pantr_alm_kernel<QuadModelT<false>, 0, 0, 0>"         # n = 280: the generic passes
    b.close()


def test_a_helper_that_is_late_in_the_middle_of_a_solve_changes_nothing_but_time():
    """Two wavefronts per instance with more workgroups than the SIMDs can run side by side (768 instances: three per
    CU, helpers and solvers sharing SIMDs) and a patience short enough to run out in the middle of solves
    (`spec_patience=50us`, `5us`): wherever a solver wavefront stops waiting it tells its helper to stop, evaluates the
    step itself and goes on alone - the solutions and counters are those of `speculation=off`, `spec_timeouts` says how
    often it happened, every workgroup ends (the launch returns)."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpc_hanging_chain_N60_warm300.npz"))
    x0 = np.vstack([d["x0"], d["x0"], d["x0"][:168]])
    u0 = np.vstack([d["guess"], np.zeros_like(d["guess"]), d["guess"][:168]])
    assert len(x0) == 768
    base = tuple(pa.DODO_PANTR_OPTS) + ("alm.max_time=5s",)
    b = pa.PantrBatch("hanging_chain", 60, 768, opts=base + ("speculation=off",))
    u_off, st_off = b.solve(x0, u0)
    b.close()
    assert np.all(st_off["status"] == 1)
    timeouts = {}
    for patience in ("50us", "5us"):
        b = pa.PantrBatch("hanging_chain", 60, 768, opts=base + ("speculation=on", f"spec_patience={patience}"))
        assert "spec" in b.kernel_name(768)
        u, st = b.solve(x0, u0)
        b.close()
        assert np.array_equal(u, u_off)
        for k in ("status", "iterations", "cg_iterations", "accepted", "rejected", "qub_backtracks", "n_psi", "n_psi_grad_psi",
                  "n_grad_L", "n_hess_psi_prod", "n_hess_psi_prod_calls", "n_prox_grad_step", "eps", "final_psi", "final_fbe"):
            assert np.array_equal(st[k], st_off[k]), (patience, k)
        timeouts[patience] = int((st["spec_timeouts"] > 0).sum())
    # (how many solvers ran out of patience depends on who gets the SIMDs when: anything from none to most - whenever the
    # solver asks, the helper has usually had the candidate's 70 us for its 40 us product)
    assert all(0 <= t <= 768 for t in timeouts.values())


def test_non_finite_instances_end_like_the_oracle(rng):
    """Instances driven to non-finite values - two chain points on top of each other (a spring of length zero: L / r is
    infinite), a NaN in the initial guess, an infinite state - end with the oracle's status, counters and solution slot,
    and their neighbours in the batch are not touched.  (The lane constants that replaced selects in the tangent sweeps spread a
    NaN over more lanes of a product than the oracle's loops do; every sum over a vector is NaN either way.)"""
    N, B = 12, 5
    x0 = np.tile(disturbed_state(), (B, 1))
    u0 = random_inputs(rng, B, N, -0.3, 0.3)
    x0[1, 3:6] = x0[1, 0:3]            # point 1 on point 0
    u0[2, 7] = np.nan
    x0[3, 40] = np.inf
    b = pa.PantrBatch("hanging_chain", N, B, opts=pa.DODO_PANTR_OPTS)
    u, st = b.solve(x0, u0)
    b.close()
    for i in range(B):
        u_o, st_o, _ = ob.solve(N, x0[i], u0[i])
        assert st["status"][i] == st_o["status"], (i, st["status"][i], st_o["status"])
        for k in ("iterations", "n_psi", "n_psi_grad_psi", "n_grad_psi", "n_grad_L", "n_hess_psi_prod", "cg_iterations"):
            assert st[k][i] == st_o[k], (i, k)
        assert np.array_equal(u[i], u_o, equal_nan=True)
    # (what they end as is the specification's business: fmin / fmax drop a NaN, so the stop criterion of such an instance
    # is tiny and it "converges" in one iteration with a cost that is not a number - on both sides alike)
    assert st["status"][0] == 1 and st["status"][4] == 1 and np.all(np.isfinite(st["final_psi"][[0, 4]]))
    assert not np.any(np.isfinite(st["final_psi"][1:4]))
