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
