"""Shared helpers for the tests (seeded inputs, the reference's benchmark settings)."""
import numpy as np

import oracle_binding as ob

NX, NU = 57, 3


def disturbed_state(num_dist=5, u_dist=(-1.0, 1.0, 1.0)):
    """x0 after the initial disturbance of mpc-driver.cpp:153-154 with dodo.py:94-95 settings."""
    x = ob.initial_state()
    for _ in range(num_dist):
        x = ob.simulate(x, u_dist)
    return x


def random_states(rng, B, scale=0.05):
    """Physically sensible random states: disturbed chain + small perturbation."""
    base = disturbed_state()
    return base[None, :] + scale * rng.standard_normal((B, NX))


def random_inputs(rng, B, N, lo=-1.2, hi=1.2):
    return rng.uniform(lo, hi, size=(B, NU * N))
