#!/usr/bin/env python3
"""Which variant of the oracle's "recollection" choices agrees better with the only numbers
the reference publishes for this path: the per-step solver times of its N = 60 runs
(tests/golden/published_series.json, extracted from /root/reference/images/*.svg)?

CPU only; uses the oracle (test infrastructure) - a measurement aid, not on the product path.  For every variant
(PANTRParams::variant of oracle/pantr_oracle.hpp) and each of the six series (3 problems x cold/warm) it runs the
oracle's closed loop and prints
  * Spearman rank correlation between the per-step work (evaluation-weighted sweeps: 2 Hv + 2 gradients + psi) and
    the published per-step time, and the Pearson correlation of their logarithms,
  * the cold/warm ratio of total work next to the published ratio of mean times,
  * converged steps.
Time is not iterations: a published second is alpaqa on CasADi code on an unstated CPU.  What the comparison can
show is whether a variant walks through the same easy and hard steps."""
import json, os, sys
import numpy as np
from scipy.stats import spearmanr, pearsonr
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob

pub = json.load(open(os.path.join(ROOT, "tests", "golden", "published_series.json")))["series"]
VARIANTS = [(0, "specification"), (1, "Lipschitz step x + h"), (2, "initial radius ||grad||"), (4, "q_K = 0"),
            (8, "model value of the whole step"),
            (16, "psi(xhat) + bound test at the prox iterate BEFORE the direction (SURVEY Appendix A step 2)")]
only = [int(a) for a in sys.argv[1:]] or None


def work(s):
    return 2 * s["n_hess_psi_prod_calls"] + 2 * (s["n_psi_grad_psi"] + s["n_grad_L"] + s["n_grad_psi"]) + s["n_psi"]


def run(problem, warm, variant):
    if problem == "hanging_chain":
        return ob.mpc(60, 60, warm, params=ob.dodo_params(variant=variant))[3]
    return ob.quad_mpc(problem, 60, 60, warm, params=ob.quad_dodo_params(problem, variant=variant))[2]


rows = []
for v, name in VARIANTS:
    if only and v not in only:
        continue
    print(f"variant {v}: {name}")
    sp, lp = [], []
    for problem in ("hanging_chain", "quadcopter", "realistic_quadcopter"):
        tot = {}
        for warm in (False, True):
            st = run(problem, warm, v)
            w = np.array([work(s) for s in st], float)
            p = np.array(pub[problem]["pantr"]["warm" if warm else "cold"])
            rho, r = spearmanr(w, p)[0], pearsonr(np.log(w + 1), np.log(p))[0]
            tot[warm] = w.sum()
            conv = sum(s["status"] == 1 for s in st)
            qub, qub_prox, its = (sum(s[k] for s in st) for k in ("qub_backtracks", "qub_at_prox", "iterations"))
            sp.append(rho); lp.append(r)
            print(f"   {problem:22s} {'warm' if warm else 'cold'}: spearman {rho:+.2f}  log-pearson {r:+.2f}  work {w.sum():9.0f} sweeps  "
                  f"converged {conv}/60  bound test failed {qub}x ({qub_prox}x at a prox iterate) in {its} iterations")
        pr = np.mean(pub[problem]["pantr"]["cold"]) / np.mean(pub[problem]["pantr"]["warm"])
        print(f"   {problem:22s} cold/warm: work {tot[False]/tot[True]:.2f}, published time {pr:.2f}")
    print(f"   mean spearman over the six series {np.mean(sp):+.3f}, over the four with structure (not the two flat warm "
          f"quadcopter series) {np.mean([sp[0], sp[1], sp[2], sp[4]]):+.3f}; mean log-pearson {np.mean(lp):+.3f}")
