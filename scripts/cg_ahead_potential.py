#!/usr/bin/env python3
"""What would deeper speculation be worth on the headline?  (CPU; uses the oracle - a development aid.)

Steihaug's CG iterates do not depend on the trust radius until they cross it, so the helper wavefront of the
two-wavefront kernel, which idles through most of a CG loop, could run the NEXT iteration's CG ahead (assuming the
candidate is rejected: the next point is the prox iterate, known early) and hand its state over.  This script walks
the oracle's iterations of the instance that decides the bench pass (and of the whole batch) and bounds what that
would save, with the measured times of round 5: a product 38 us, the candidate's (psi, grad psi) 70 us, the rest of
an iteration 19 us, the helper's first-order work 32 us, the model-value product it serves 38 us, half a product of
waiting at every hand-over.

    python scripts/cg_ahead_potential.py > profiles/<tag>_cg_ahead_potential.txt"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob

T_PROD, T_CAND, T_REST, T_FIRST, T_MODEL = 38.0, 70.0, 19.0, 32.0, 38.0


def walk(acc, cg):
    """(microseconds now, microseconds with the next CG run ahead on the helper after every rejection)."""
    now = ahead_t = 0.0
    ahead = 0
    for k in range(len(acc)):
        n = int(cg[k])
        now += n * T_PROD + T_CAND + T_REST
        mine = max(0, n - ahead)
        wait = T_PROD / 2 if 0 < ahead < n else 0.0
        dur = mine * T_PROD + wait + T_CAND + T_REST
        ahead_t += dur
        nxt = int(cg[k + 1]) if k + 1 < len(acc) else 0
        ahead = 0 if acc[k] else min(nxt, int(max(0.0, dur - T_FIRST - T_MODEL) // T_PROD))
    return now, ahead_t


d = np.load(os.path.join(ROOT, "tests/golden/mpc_hanging_chain_N60_warm300.npz"))
N = 60
print("# Upper bound of what running the next iteration's CG ahead on the helper wavefront would save (oracle iterations,")
print("# round-5 times per piece; scripts/cg_ahead_potential.py).  A = candidate accepted, r = rejected; CG products per iteration.")
u, st, tr = ob.solve(N, d["x0"][119], np.zeros(3 * N), trace_cap=1000)
acc, cg = tr[:, 10].astype(int), tr[:, 9].astype(int)
print(f"instance 419 (cold twin of time step 119): {st['iterations']} iterations, {int(cg.sum())} CG products, "
      f"{int(acc.sum())} accepted")
print("  " + "".join("A" if a else "r" for a in acc))
print("  " + " ".join(str(c) for c in cg))
after_acc = [int(cg[k]) for k in range(1, len(cg)) if acc[k - 1]]
after_rej = [int(cg[k]) for k in range(1, len(cg)) if not acc[k - 1]]
print(f"  CG products of an iteration that follows an ACCEPTED candidate: mean {np.mean(after_acc):.1f} (sum {sum(after_acc)}); "
      f"that follows a REJECTED one: mean {np.mean(after_rej):.1f} (sum {sum(after_rej)})")
a, b = walk(acc, cg)
print(f"  modelled pass of this instance: {a / 1e3:.1f} ms now, {b / 1e3:.1f} ms with the CG run ahead: -{100 * (1 - b / a):.0f} %")
tot_a = tot_b = 0.0
for cold in (False, True):
    for i in range(0, 300, 10):
        u0 = np.zeros(3 * N) if cold else d["guess"][i]
        _, _, tr = ob.solve(N, d["x0"][i], u0, trace_cap=1000)
        a, b = walk(tr[:, 10].astype(int), tr[:, 9].astype(int))
        tot_a += a
        tot_b += b
print(f"every tenth instance of the batch (30 warm + 30 cold): summed solve time -{100 * (1 - tot_b / tot_a):.0f} %")
print("# The long CG loops follow ACCEPTED candidates - the one case in which the next point is not known ahead - and are")
print("# themselves mostly rejected; what follows a rejection is short.  Not built: at most this much for a second CG state")
print("# machine, a hand-over protocol and twice the tape traffic while the batch is busy.")
