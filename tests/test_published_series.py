"""The only numbers the reference holds for this path: the per-MPC-step solver times of its N = 60 runs, published as
plots (README.md:36,69,106 -> images/*.svg; tests/golden/published_series.json, extracted by
tests/golden/extract_published_series.py in the build container).

Time on an unstated CPU is not an iteration count, so this pins nothing bit for bit - parity stays "unpinned" - but a
restatement of alpaqa's PANTR that walks other trajectories than the reference's would not reproduce WHICH steps of a
closed loop are easy and which are hard, how much harder a cold start is than a warm one, or how flat the warm
quadcopter runs are.  The oracle's evaluation counts do (scripts/adjudicate_choices.py prints the same figures for the
variants of the six "recollection" choices of DESIGN.md section 5)."""
import json
import os

import numpy as np
import pytest
from scipy.stats import pearsonr, spearmanr

import oracle_binding as ob

HERE = os.path.dirname(os.path.abspath(__file__))
PUB = json.load(open(os.path.join(HERE, "golden", "published_series.json")))


def work(s):
    """Sweeps over the horizon an evaluation costs: a product two, a gradient two, a value one; products in the
    reference's convention (alpaqa counts the call where the product is known to be zero)."""
    return 2 * s["n_hess_psi_prod_calls"] + 2 * (s["n_psi_grad_psi"] + s["n_grad_L"] + s["n_grad_psi"]) + s["n_psi"]


@pytest.fixture(scope="module")
def closed_loops():
    out = {}
    for problem in ("hanging_chain", "quadcopter", "realistic_quadcopter"):
        for warm in (False, True):
            if problem == "hanging_chain":
                stats = ob.mpc(60, 60, warm)[3]
            else:
                stats = ob.quad_mpc(problem, 60, 60, warm)[2]
            assert all(s["status"] == 1 for s in stats)       # the reference's N = 60 runs converged too (no x markers)
            out[(problem, warm)] = np.array([work(s) for s in stats], float)
    return out


def published(problem, warm):
    return np.array(PUB["series"][problem]["pantr"]["warm" if warm else "cold"])


def test_fixture_is_the_published_record():
    """means of the 60 points of the hanging-chain series; 60 points per series, all four solvers."""
    assert PUB["horizon"] == 60 and PUB["steps"] == 60
    assert abs(published("hanging_chain", False).mean() - 0.3454) < 5e-4
    assert abs(published("hanging_chain", True).mean() - 0.1301) < 5e-4
    for problem, solvers in PUB["series"].items():
        for solver, starts in solvers.items():
            for start, v in starts.items():
                assert len(v) == 60 and min(v) > 0, (problem, solver, start)


@pytest.mark.parametrize("problem,warm", [("hanging_chain", False), ("hanging_chain", True), ("quadcopter", False),
                                          ("realistic_quadcopter", False)])
def test_easy_and_hard_steps_are_the_published_ones(closed_loops, problem, warm):
    """The four series with structure: rank correlation of the oracle's per-step work with the published per-step
    time >= 0.6 (measured 0.69, 0.79, 0.75, 0.72), correlation of the logarithms >= 0.8."""
    w, p = closed_loops[(problem, warm)], published(problem, warm)
    assert spearmanr(w, p)[0] >= 0.6
    assert pearsonr(np.log(w), np.log(p))[0] >= 0.8


@pytest.mark.parametrize("problem", ["quadcopter", "realistic_quadcopter"])
def test_warm_quadcopter_runs_are_as_flat_as_published(closed_loops, problem):
    """The warm quadcopter series carry no structure to correlate with (published: 27 +- 5 ms and 79 +- 15 ms per step):
    what can be compared is how flat they are - coefficient of variation 0.18 / 0.19 published, 0.17 / 0.15 here."""
    w, p = closed_loops[(problem, True)], published(problem, True)
    cv = lambda v: v.std() / v.mean()
    assert cv(p) < 0.25 and cv(w) < 0.25 and abs(cv(w) - cv(p)) < 0.08
    assert w.max() / np.median(w) < 2.0 and p.max() / np.median(p) < 2.0


@pytest.mark.parametrize("problem", ["hanging_chain", "quadcopter", "realistic_quadcopter"])
def test_cold_start_costs_what_it_cost_the_reference(closed_loops, problem):
    """Ratio of cold to warm: total work here against mean solver time there, within 30 % (published 2.65 / 5.04 / 4.01;
    here 2.32 / 5.55 / 4.49)."""
    ratio = closed_loops[(problem, False)].sum() / closed_loops[(problem, True)].sum()
    ref = published(problem, False).mean() / published(problem, True).mean()
    assert abs(ratio / ref - 1) < 0.30, (ratio, ref)


# ---- the two comparators of row f4 (method=panoc, strucpanoc; mpc-driver.cpp:371-384) against their published series ----
@pytest.fixture(scope="module")
def comparator_loops():
    out = {}
    for method in ("panoc", "strucpanoc"):
        for warm in (False, True):
            stats = ob.mpc(60, 60, warm, params=ob.dodo_params(method))[3]
            assert all(s["status"] == 1 for s in stats)
            out[(method, warm)] = np.array([work(s) for s in stats], float)     # (no products: gradients and values)
    return out


@pytest.mark.parametrize("method,warm", [("panoc", False), ("panoc", True), ("strucpanoc", False), ("strucpanoc", True)])
def test_comparators_rank_their_steps_like_the_reference(comparator_loops, method, warm):
    """PANOC+ and structured PANOC+ (L-BFGS 50) on the hanging chain, N = 60: per-step work against the published
    per-step time - Spearman 0.79 / 0.92 (PANOC+ cold / warm), 0.77 / 0.87 (structured)."""
    w = comparator_loops[(method, warm)]
    p = np.array(PUB["series"]["hanging_chain"][method]["warm" if warm else "cold"])
    assert spearmanr(w, p)[0] >= 0.7


@pytest.mark.parametrize("warm", [False, True])
def test_comparators_relative_cost_is_the_published_one(comparator_loops, warm):
    """Both comparators spend their time in the same sweeps, so their work ratio is comparable with the ratio of the
    published mean times: 2.01 against 1.86 (cold), 1.37 against 1.35 (warm)."""
    key = "warm" if warm else "cold"
    ratio = comparator_loops[("panoc", warm)].sum() / comparator_loops[("strucpanoc", warm)].sum()
    ref = np.mean(PUB["series"]["hanging_chain"]["panoc"][key]) / np.mean(PUB["series"]["hanging_chain"]["strucpanoc"][key])
    assert abs(ratio / ref - 1) < 0.15, (ratio, ref)


# ---- the 36 published averages per horizon (README.md:33,66,103) ------------------------------
AVG_H = PUB["average_horizons"]
# the reference's realistic_quadcopter N = 50 cold run has non-converged steps (703 ms, marked x in the
# plot: twice the N = 60 average); the oracle's run converges (60 of 60) - that one point is compared with nothing
KNOWN_OUTLIER = ("realistic_quadcopter", False, 50)


@pytest.fixture(scope="module")
def work_per_horizon(closed_loops):
    """Mean per-step work x N (an evaluation sweeps over N stages) of the oracle's closed loops at the six horizons."""
    out = {}
    for problem in ("hanging_chain", "quadcopter", "realistic_quadcopter"):
        for warm in (False, True):
            w = []
            for N in AVG_H:
                if N == 60:
                    w.append(closed_loops[(problem, warm)].mean() * N)
                    continue
                stats = ob.mpc(N, 60, warm)[3] if problem == "hanging_chain" else ob.quad_mpc(problem, N, 60, warm)[2]
                assert all(s["status"] == 1 for s in stats)
                w.append(np.mean([work(s) for s in stats]) * N)
            out[(problem, warm)] = np.array(w)
    return out


def test_averages_fixture_is_the_published_record():
    """hanging_chain cold 2.21 / 7.23 / 22.1 / 56.9 / 176 / 345 ms, warm 1.76 ... 130 ms; the N = 60
    average is the mean of the 60 per-step points of the other plot (the extraction checks it to 0.2 % for every solver)."""
    assert AVG_H == [10, 20, 30, 40, 50, 60]
    a = PUB["averages"]["hanging_chain"]["pantr"]
    assert np.allclose(a["cold"], [2.21e-3, 7.23e-3, 22.1e-3, 56.9e-3, 176e-3, 345e-3], rtol=5e-3)
    assert np.allclose(a["warm"], [1.76e-3, 5.80e-3, 13.2e-3, 23.5e-3, 50.4e-3, 130e-3], rtol=5e-3)
    for problem in PUB["averages"]:
        for start in ("cold", "warm"):
            assert abs(PUB["averages"][problem]["pantr"][start][5] / published(problem, start == "warm").mean() - 1) < 2e-3
    assert sum(len(v) for p in PUB["averages"].values() for v in p["pantr"].values()) == 36


@pytest.mark.parametrize("problem", ["hanging_chain", "quadcopter", "realistic_quadcopter"])
@pytest.mark.parametrize("warm", [False, True])
def test_cost_grows_with_the_horizon_as_published(work_per_horizon, problem, warm):
    """work(N) / work(60) of the oracle against time(N) / time(60) of the reference, six horizons, cold and warm: a
    machine-independent shape.  Measured (work slope / time slope of the log-log fit): hanging chain 2.47 / 2.84 cold,
    1.94 / 2.27 warm; quadcopter 2.53 / 2.74, 1.69 / 1.92; realistic_quadcopter 2.32 / 2.79 (2.32 / 2.50 without the known
    outlier), 1.47 / 1.73; every point's ratio within [0.72, 1.95].  The work grows a little slower than the time on all
    six curves (by 0.1 to 0.4 in the exponent): an evaluation of the reference is straight-line generated code of O(N)
    instructions, whose cost per instruction rises with the code's size (the full-Hessian variant "fails to compile (too
    big)", problems/CMakeLists.txt:38-39) - the count of sweeps x N cannot show that.  Asserted: the exponents differ by at
    most 0.45 and never the other way by more than 0.1; every point within a factor of two.  Evidence, not a pin."""
    w, t = work_per_horizon[(problem, warm)], np.array(PUB["averages"][problem]["pantr"]["warm" if warm else "cold"])
    keep = np.array([(problem, warm, N) != KNOWN_OUTLIER for N in AVG_H])
    h = np.log(np.array(AVG_H, float))
    slope_w = np.polyfit(h[keep], np.log(w[keep]), 1)[0]
    slope_t = np.polyfit(h[keep], np.log(t[keep]), 1)[0]
    assert -0.45 <= slope_w - slope_t <= 0.10, (slope_w, slope_t)
    ratio = (w / w[-1]) / (t / t[-1])
    assert np.all((ratio[keep] > 0.5) & (ratio[keep] < 2.0)), ratio
