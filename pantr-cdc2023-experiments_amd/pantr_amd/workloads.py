"""The batches of BASELINE.json / SURVEY.md section 8(d), as host-side enumerators.

  dodo_tasks            the rows of new-benchmarks-paper/dodo.py:69-111 for one solver (problem x
                        {warm, cold} x horizon 10..60) with their option lists and output names;
                        dodo_pantr_tasks: the PANTR rows
  config5_disturbances  the 1024 synthetic disturbances of config 5 (std::mt19937_64(12345) ->
                        uniform_real_distribution<double>(-1, 1), three draws per instance)
"""
import numpy as np

from . import DODO_HANGING_CHAIN_OPTS, DODO_OPTS

NUM_SIM = 60          # dodo.py:10
HORIZONS = tuple(range(10, 61, 10))  # dodo.py:5,9,71


def problem_name(problem, warm, solver, horizon, num_sim=NUM_SIM):
    """Output file stem, dodo.py:62-66."""
    return f"mpc-{problem}-{horizon},{num_sim}-{'warm' if warm else 'cold'}-{solver}"


def dodo_tasks(problem="hanging_chain", method="pantr", horizons=HORIZONS, num_sim=NUM_SIM):
    """Tasks of mpc_benchmark(problem) for one solver (pantr, panoc, strucpanoc), dodo.py:69-111,
    in its enumeration order."""
    tasks = []
    for warm in (True, False):
        for horizon in horizons:
            opts = list(DODO_OPTS[method][1:])                  # dodo.py:24-40 (method= goes first below)
            if problem == "hanging_chain":
                opts += list(DODO_HANGING_CHAIN_OPTS)          # dodo.py:94-95
            else:
                opts += ["num_dist=0", "solver.max_iter=250"]   # dodo.py:97-100
            if problem == "realistic_quadcopter":
                if method == "pantr":
                    opts += ["accel.tol_scale=1"]               # dodo.py:101-103
                opts += ["alm.initial_tolerance=1e-1"]          # dodo.py:104-105
            name = problem_name(problem, warm, method, horizon, num_sim)
            # run-single-mpc-exp.sh:19
            argv = [problem, str(horizon), "ss2p", f"method={method}", f"num_sim={num_sim}",
                    f"results_name={name}", f"warm={'true' if warm else 'false'}"] + opts
            tasks.append(dict(name=name, problem=problem, method=method, horizon=horizon, warm=warm,
                              num_sim=num_sim, argv=argv))
    return tasks


def dodo_pantr_tasks(problem="hanging_chain", horizons=HORIZONS, num_sim=NUM_SIM):
    return dodo_tasks(problem, "pantr", horizons, num_sim)


class MT19937_64:
    """std::mt19937_64 (the generator SURVEY.md section 8(d) config 5 prescribes)."""
    NN, MM = 312, 156
    MATRIX_A, UM, LM = 0xB5026F5AA96619E9, 0xFFFFFFFF80000000, 0x7FFFFFFF
    MASK = (1 << 64) - 1

    def __init__(self, seed=5489):
        mt = [seed & self.MASK]
        for i in range(1, self.NN):
            mt.append((6364136223846793005 * (mt[-1] ^ (mt[-1] >> 62)) + i) & self.MASK)
        self.mt, self.idx = mt, self.NN

    def _twist(self):
        mt, NN, MM = self.mt, self.NN, self.MM
        for i in range(NN):
            x = (mt[i] & self.UM) | (mt[(i + 1) % NN] & self.LM)
            mt[i] = mt[(i + MM) % NN] ^ (x >> 1) ^ (self.MATRIX_A if x & 1 else 0)
        self.idx = 0

    def __call__(self):
        if self.idx >= self.NN:
            self._twist()
        x = self.mt[self.idx]
        self.idx += 1
        x ^= (x >> 29) & 0x5555555555555555
        x ^= (x << 17) & 0x71D67FFFEDA60000
        x ^= (x << 37) & 0xFFF7EEE000000000
        x ^= x >> 43
        return x & self.MASK

    def uniform(self, a, b):
        """std::uniform_real_distribution<double>(a, b): generate_canonical<double, 53> with one
        64-bit draw, x / 2^64 (libstdc++ computes it in long double), clamped below 1."""
        x = self()
        hi, lo = x >> 11, x & 0x7FF           # round-to-nearest-even of x * 2^-64 to 53 bits
        if lo > 0x400 or (lo == 0x400 and (hi & 1)):
            hi += 1
        r = hi * 2.0 ** -53
        if r >= 1.0:
            r = np.nextafter(1.0, 0.0)
        return r * (b - a) + a


def config5_disturbances(B=1024, seed=12345):
    """d_j in [-1,1]^3, three draws per instance in order."""
    g = MT19937_64(seed)
    return np.array([[g.uniform(-1.0, 1.0) for _ in range(3)] for _ in range(B)])


# ---- replay batches of BASELINE.json's configs, produced by the product's own closed loops --------
# Each returns a dict: problem, horizon, x0 (B, nx), u0 (B, n), y0 (B, m) or None, sol (B, n) the
# solutions the closed loops recorded (a replay must reproduce them bit for bit), warm (B,) flags,
# opts (solver options), label.

def _quad_solver_opts(problem):
    """dodo.py:12-33 + :96-105, solver options only."""
    t = dodo_pantr_tasks(problem, horizons=(10,))[0]
    return tuple(o for o in t["argv"][3:] if o.startswith(("solver.", "alm.", "dir.", "accel.")))


def closed_loop_batch(problem, horizon, num_sim=NUM_SIM, starts=(True, False), device=0, opts=None):
    """The timed steps of the closed loops of one (problem, horizon) for the given starts (True =
    warm), recorded on the device with PantrBatch.mpc: the replay batch of SURVEY.md 8(d)."""
    from . import PantrBatch, DODO_PANTR_OPTS
    chain = problem == "hanging_chain"
    opts = tuple(opts) if opts is not None else (tuple(DODO_PANTR_OPTS) if chain else _quad_solver_opts(problem))
    b = PantrBatch(problem, horizon, 1, device=device, opts=opts)
    u_dist = np.array([[-1.0, 1.0, 1.0]]) if chain else np.zeros((1, b.nu))      # dodo.py:94-100
    num_dist = 5 if chain else 0
    x0, u0, y0, sol, warm_flags, cost = [], [], [], [], [], []
    for warm in starts:
        states, inputs, stats, rec = b.mpc(num_sim, warm, u_dist, num_dist=num_dist, record=True)
        if not np.all(stats["status"] == 1):
            raise RuntimeError(f"{problem} N={horizon} {'warm' if warm else 'cold'}: a closed-loop step did not converge")
        x0.append(states[:num_sim, 0]); u0.append(rec["guess_u"][:, 0]); y0.append(rec["guess_y"][:, 0])
        sol.append(inputs[:, 0]); warm_flags += [warm] * num_sim
        cost.append(stats["n_hess_psi_prod"][:num_sim, 0].astype(np.float64) * horizon)
    m = b.m
    b.close()
    return dict(problem=problem, horizon=horizon, x0=np.vstack(x0), u0=np.vstack(u0),
                y0=np.vstack(y0) if m else None, sol=np.vstack(sol), warm=np.array(warm_flags), opts=opts,
                # what every instance cost when it was recorded: products x horizon (sharding deals longest first)
                cost=np.concatenate(cost),
                label=f"{problem} N={horizon}: {num_sim} time steps x {len(starts)} starts")


def headline_batch(device=0, num_sim=300, horizon=60):
    """BASELINE.json configs[1] + its cold twin: the num_sim timed steps of ONE warm-started closed
    loop (num_dist=5, u_dist=-1,1,1), each to be solved from its warm guess and from the cold guess
    u = 0 (mpc-driver.cpp:168-175 | 192-197): 2 num_sim instances, warm half first."""
    w = closed_loop_batch("hanging_chain", horizon, num_sim, starts=(True,), device=device)
    B = len(w["x0"])
    return dict(w, x0=np.vstack([w["x0"], w["x0"]]), u0=np.vstack([w["u0"], np.zeros_like(w["u0"])]),
                sol=w["sol"], warm=np.r_[np.ones(B, bool), np.zeros(B, bool)], cost=None,  # (the cold twins have no record)
                label=f"hanging_chain N={horizon}: {num_sim} time steps of one warm closed loop, each solved warm and cold")


def stress_batch(B=1024, device=0, horizon=120):
    """BASELINE.json configs[4]: synthetic initial states x0_j = f_d^5(x_rest, d_j), cold starts."""
    from . import PantrBatch, DODO_PANTR_OPTS
    d = config5_disturbances(1024)[:B]
    b = PantrBatch("hanging_chain", horizon, B, device=device)
    x0 = np.tile(b.initial_state(), (B, 1))
    for _ in range(5):
        x0 = b.simulate_dynamics(x0, d)
    n = b.n
    b.close()
    return dict(problem="hanging_chain", horizon=horizon, x0=x0, u0=np.zeros((B, n)), y0=None, sol=None, cost=None,
                warm=np.zeros(B, bool), opts=tuple(DODO_PANTR_OPTS),
                label=f"hanging_chain N={horizon}: {B} synthetic initial states (mt19937_64(12345)), cold")


def expected_cost(horizon, warm):
    """Relative cost of one instance for longest-first dealing where a batch holds no recorded solves (the replay
    batches carry `cost`, their own recorded product counts): the published mean solver times grow like N^2.7
    and a cold start costs about 2.5 warm ones."""
    return float(horizon) ** 2.7 * (1.0 if warm else 2.5)
