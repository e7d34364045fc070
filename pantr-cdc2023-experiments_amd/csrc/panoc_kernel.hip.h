// PANOC+ inner solver with L-BFGS and structured L-BFGS directions on gfx950 (SURVEY.md section 8
// row f4: the comparators of the reference's benchmark), generic over the wave-cooperative problem.
//
// What it replaces (paths relative to /root/reference):
//   alpaqa::PANOCSolver<LBFGSDirection>            method=panoc       drivers/mpc-driver.cpp:378-384
//   alpaqa::PANOCSolver<StructuredLBFGSDirection>  method=strucpanoc  drivers/mpc-driver.cpp:371-377
//   options accel.memory=50, dir.hessian_vec=0                        new-benchmarks-paper/dodo.py:34-40
// alpaqa itself is an un-vendored submodule; the algorithm is restated in oracle/panoc_oracle.hpp,
// which this file follows operation for operation.
//
// One wavefront per instance as for PANTR.  The L-BFGS pairs (memory x 2 vectors of n_pad doubles)
// do not fit the LDS next to the iterates, so they live in an HBM ring per instance; lane l owns the
// elements l, l+64, ... of every vector, in LDS and in HBM alike, so a pair is written and read back
// by the same lanes (coalesced 512-byte rows, no cross-lane ordering needed).  rho and alpha are
// small LDS arrays (in the vector slots PANOC leaves free), written by lane 0 and read by all (one
// wavefront's LDS operations execute in order).
#pragma once
#include "pantr_kernel.hip.h"

namespace pantr {

/// The L-BFGS ring of one instance.
struct LbfgsRing {
    double *S, *Y;       ///< HBM [memory][npad]
    double *rho, *alpha; ///< LDS [memory]
    int memory, npad, idx;
    bool full;
    __device__ __forceinline__ void reset() {
        idx  = 0;
        full = false;
    }
    __device__ __forceinline__ int succ(int i) const { return i + 1 < memory ? i + 1 : 0; }
    __device__ __forceinline__ int pred(int i) const { return i > 0 ? i - 1 : memory - 1; }
    __device__ __forceinline__ int count() const { return full ? memory : idx; }
    /// c-th pair counted from the newest / from the oldest
    __device__ __forceinline__ int from_newest(int c) const {
        int i = idx - 1 - c;
        return i < 0 ? i + memory : i;
    }
    __device__ __forceinline__ int from_oldest(int c) const { return from_newest(count() - 1 - c); }
    __device__ __forceinline__ double *s(int i) const { return S + (size_t)i * npad; }
    __device__ __forceinline__ double *y(int i) const { return Y + (size_t)i * npad; }
};

/// One PANOC inner solve; same contract as pantr_inner().  (-DPANTR_PHASE_TIMING: stats.phase_s[] receives
/// the seconds spent in the L-BFGS application / psi_grad / psi(xhat) / the L-BFGS update.)  lbfgs_ws: 2 * memory * npad doubles of
/// HBM for this instance.
template <class PB, bool STRUCTURED>
__device__ __forceinline__ InnerResult panoc_inner(PB &pb, const KernelParams &P, double tolerance,
                                                   double deadline_s, long long t_start, double *x_io,
                                                   DeviceStats &st, double *trace, int trace_cap,
                                                   double *lbfgs_ws) {
    const int lane = threadIdx.x;
    const int n = pb.n, nj = pb.nj;
#ifdef PANTR_REST_TIMING // (development build of pantr_inner's bookkeeping timing: unused here)
    double rest_bucket_[4] = {0, 0, 0, 0};
    long long rest_last_ = 0;
    int rest_cur_ = 0;
#endif
    enum { S_IT0 = 0, S_IT1 = 4, S_Q = 8, S_WORK = 9, S_RA = 10 }; // S_RA: slots 10..18, rho | alpha
    IterateRef curr{S_IT0, 0}, next{S_IT1, 1};
    auto X_  = [&](const IterateRef &it) { return pb.vec(it.base + 0); };
    auto XH_ = [&](const IterateRef &it) { return pb.vec(it.base + 1); };
    auto G_  = [&](const IterateRef &it) { return pb.vec(it.base + 2); };
    auto P_  = [&](const IterateRef &it) { return pb.vec(it.base + 3); };
    double *q = pb.vec(S_Q);

    LbfgsRing lb;
    lb.memory = P.lbfgs_memory;
    lb.npad   = pb.npad;
    lb.S      = lbfgs_ws;
    lb.Y      = lbfgs_ws + (size_t)lb.memory * lb.npad;
    lb.rho    = pb.vec(S_RA);
    lb.alpha  = pb.vec(S_RA) + lb.memory;
    lb.reset();

    // ---- L-BFGS (oracle/panoc_oracle.hpp: LBFGS) ----
    // pair <- (xn - x, sign (bn - b)); not forced: cautious acceptance test first
    auto lbfgs_update = [&](const double *x, const double *xn, const double *b, const double *bn, bool positive,
                            bool forced) {
        double a_ys = 0.0, a_ss = 0.0;
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + lane;
            if (i < n) {
                double s = xn[i] - x[i];
                double y = positive ? bn[i] - b[i] : b[i] - bn[i];
                a_ys     = a_ys + y * s;
                a_ss     = a_ss + s * s;
            }
        }
        double yTs = uniform(wave_sum(a_ys));
        if (!forced) {
            double sTs = uniform(wave_sum(a_ss));
            if (sTs < P.lbfgs_min_abs_s)
                return false;
            if (!is_finite(yTs))
                return false;
            double a_yTs = P.lbfgs_force_pos_def ? yTs : fabs(yTs);
            if (a_yTs < P.lbfgs_min_div_fac * sTs)
                return false;
        }
        double *srow = lb.s(lb.idx), *yrow = lb.y(lb.idx);
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + lane;
            if (i < n) {
                srow[i] = xn[i] - x[i];
                yrow[i] = positive ? bn[i] - b[i] : b[i] - bn[i];
            }
        }
        if (lane == 0)
            lb.rho[lb.idx] = 1.0 / yTs;
        lb.idx         = lb.succ(lb.idx);
        lb.full        = lb.full || lb.idx == 0;
        return true;
    };
    // A pair's rows in registers (lane l holds elements l, l+64, ...): the two-loop recursion is a
    // chain of 2 x count row operations, each a reduction that needs the previous one's result, so
    // the rows of the next RING pairs are in flight from HBM while the current pair is reduced (RING
    // register sets, loops unrolled by RING so that no set is copied; an exposed fetch costs ~1 us,
    // the arithmetic of a pair ~0.3 us).
    constexpr int NJR  = 4; // register rows cover n <= 256; longer vectors take the row-by-row path
    constexpr int RING = 4;
    struct PairRegs {
        double s[NJR], y[NJR];
    };
    // element j of this lane: index (clamped into the vector so that loads need no lane mask) and
    // whether it exists; selects instead of lane-masked branches keep a row operation one basic block
    int eidx[NJR];
    bool valid[NJR];
#pragma unroll
    for (int j = 0; j < NJR; ++j) {
        int e    = j * WAVE + lane;
        valid[j] = j < nj && e < n;
        eidx[j]  = e < n ? e : n - 1;
    }
    auto fetch = [&](int i, PairRegs &r) {
        const double *srow = lb.s(i), *yrow = lb.y(i);
#pragma unroll
        for (int j = 0; j < NJR; ++j)
            if (j < nj) { // wave-uniform
                r.s[j] = srow[eidx[j]];
                r.y[j] = yrow[eidx[j]];
            }
    };
    // sum over the mask of a[j] * b[j]
    auto dot_rr = [&](const double (&a)[NJR], const double (&b)[NJR], uint32_t mask) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < NJR; ++j)
            if (j < nj) {
                double t = acc + a[j] * ((mask >> j) & 1u ? b[j] : 0.0);
                acc      = valid[j] ? t : acc;
            }
        return uniform(wave_sum(acc));
    };
    auto axmy_r = [&](double c, const double (&a)[NJR], double (&v)[NJR], uint32_t mask) { // v <- v - c a on the mask
#pragma unroll
        for (int j = 0; j < NJR; ++j)
            if (j < nj) {
                double t = v[j] - c * a[j];
                v[j]     = valid[j] && ((mask >> j) & 1u) ? t : v[j];
            }
    };
    // the same operations on rows left in HBM and the vector left in LDS (n > 256)
    auto dot_row = [&](const double *row, const double *v, uint32_t mask) {
        double acc = 0.0;
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + lane;
            if (i < n)
                acc = acc + row[i] * ((mask >> j) & 1u ? v[i] : 0.0);
        }
        return uniform(wave_sum(acc));
    };
    auto axmy_row = [&](double c, const double *row, double *v, uint32_t mask) {
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + lane;
            if (i < n && ((mask >> j) & 1u))
                v[i] = v[i] - c * row[i];
        }
    };
    auto scale = [&](double c, double *v, uint32_t mask) {
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + lane;
            if (i < n && ((mask >> j) & 1u))
                v[i] = c * v[i];
        }
    };
    // v <- H v by the two-loop recursion; masked: on the index set J only, with the curvature of
    // every pair recomputed on J (and kept) and non-positive pairs skipped
    auto lbfgs_apply = [&](double *v, uint32_t mask, bool masked) {
        const int cnt = lb.count();
        if (cnt == 0)
            return false;
        double gamma = -1.0;
        if (nj > NJR) { // row-by-row from HBM
            if (!masked) {
                int newest = lb.pred(lb.idx);
                gamma      = 1.0 / (lb.rho[newest] * dot_row(lb.y(newest), lb.y(newest), mask));
            }
            for (int c = 0; c < cnt; ++c) {
                int i = lb.from_newest(c);
                if (masked) {
                    lb.rho[i] = 1.0 / dot_row(lb.s(i), lb.y(i), mask);
                    if (!(lb.rho[i] > 0))
                        continue;
                }
                double a    = lb.rho[i] * dot_row(lb.s(i), v, mask);
                if (lane == 0)
                    lb.alpha[i] = a;
                axmy_row(a, lb.y(i), v, mask);
                if (masked && gamma < 0)
                    gamma = 1.0 / (lb.rho[i] * dot_row(lb.y(i), lb.y(i), mask));
            }
            if (masked && gamma < 0) // no pair with positive curvature on J
                return false;
            scale(gamma, v, mask);
            for (int c = 0; c < cnt; ++c) {
                int i = lb.from_oldest(c);
                if (masked && !(lb.rho[i] > 0))
                    continue;
                double beta = lb.rho[i] * dot_row(lb.y(i), v, mask);
                axmy_row(beta - lb.alpha[i], lb.s(i), v, mask);
            }
            return true;
        }
        PairRegs R[RING];
        double qr[NJR]; // the vector being transformed stays in registers for the whole recursion
#pragma unroll
        for (int j = 0; j < NJR; ++j)
            qr[j] = j < nj ? v[eidx[j]] : 0.0;
        // op(c, rows of the c-th pair in the order idx(c)) for c = 0..cnt-1
        auto sweep = [&](auto &&idx, auto &&op) {
#pragma unroll
            for (int j = 0; j < RING; ++j)
                if (j < cnt)
                    fetch(idx(j), R[j]);
            for (int c0 = 0; c0 < cnt; c0 += RING) {
#pragma unroll
                for (int j = 0; j < RING; ++j) {
                    const int c = c0 + j;
                    if (c < cnt) {
                        op(c, R[j]);
                        if (c + RING < cnt)
                            fetch(idx(c + RING), R[j]);
                    }
                }
            }
        };
        // ---- newest to oldest ----
        sweep([&](int c) { return lb.from_newest(c); },
              [&](int c, const PairRegs &r) {
                  int i = lb.from_newest(c);
                  if (!masked && c == 0)
                      gamma = 1.0 / (lb.rho[i] * dot_rr(r.y, r.y, mask));
                  double rho_i;
                  if (masked) {
                      rho_i = 1.0 / dot_rr(r.s, r.y, mask);
                      if (lane == 0)
                          lb.rho[i] = rho_i;
                      if (!(rho_i > 0))
                          return;
                  } else {
                      rho_i = lb.rho[i];
                  }
                  double a = rho_i * dot_rr(r.s, qr, mask);
                  if (lane == 0) // one writer: a 64-lane write to one LDS address serialises
                      lb.alpha[i] = a;
                  axmy_r(a, r.y, qr, mask);
                  if (masked && gamma < 0)
                      gamma = 1.0 / (rho_i * dot_rr(r.y, r.y, mask));
              });
        if (masked && gamma < 0) // no pair with positive curvature on J
            return false;
#pragma unroll
        for (int j = 0; j < NJR; ++j)
            if (j < nj) {
                double t = gamma * qr[j];
                qr[j]    = valid[j] && ((mask >> j) & 1u) ? t : qr[j];
            }
        // ---- oldest to newest ----
        sweep([&](int c) { return lb.from_oldest(c); },
              [&](int c, const PairRegs &r) {
                  int i = lb.from_oldest(c);
                  double rho_i = lb.rho[i];
                  if (masked && !(rho_i > 0))
                      return;
                  double beta = rho_i * dot_rr(r.y, qr, mask);
                  axmy_r(beta - lb.alpha[i], r.s, qr, mask);
              });
#pragma unroll
        for (int j = 0; j < NJR; ++j)
            if (valid[j])
                v[j * WAVE + lane] = qr[j];
        return true;
    };

    auto qub_violated = [&](const IterateRef &i) {
        double margin = (1.0 + fabs(i.psi_x)) * P.quadratic_upperbound_tolerance_factor;
        return i.psi_xh > ((i.psi_x + i.gradTp) + (0.5 * i.L) * i.pTp) + margin;
    };

    int status   = Busy;
    double eps_k = __builtin_nan(""), tau = __builtin_nan("");

    // ---- initial iterate and Lipschitz estimate (as pantr_inner) ----
    for (int j = 0; j < nj; ++j) {
        int i = j * WAVE + lane;
        if (i < n)
            X_(curr)[i] = x_io[i];
    }
    curr.psi_x = pb.eval_psi_grad(X_(curr), G_(curr));
    ++st.n_psi_grad_psi;
    if (P.L_0 <= 0) {
        double *h = pb.vec(S_WORK);
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + lane;
            if (i < n) {
                double g  = G_(curr)[i];
                double hh = g > 0 ? fmax(g * P.lip_eps, P.lip_delta) : fmin(g * P.lip_eps, -P.lip_delta);
                h[i]         = hh;
                XH_(curr)[i] = X_(curr)[i] - hh;
            }
        }
        double norm_h = __builtin_sqrt(pb.dot(h, h));
        pb.eval_psi_grad(XH_(curr), G_(next));
        ++st.n_grad_psi;
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + lane;
            if (i < n)
                h[i] = G_(next)[i] - G_(curr)[i];
        }
        double Lest = __builtin_sqrt(pb.dot(h, h)) / norm_h;
        curr.L      = fmin(fmax(Lest, P.L_min), P.L_max);
    } else {
        curr.L = P.L_0;
    }
    int k = 0, no_progress = 0;
    if (!is_finite(curr.L)) {
        status = NotFinite;
    } else {
        curr.gamma = P.Lgamma_factor / curr.L;
        pb.prox_grad_step(curr.gamma, X_(curr), G_(curr), XH_(curr), P_(curr), curr.pTp, curr.gradTp);
        ++st.n_prox_grad_step;
        curr.psi_xh = pb.eval_psi_xhat(XH_(curr), curr.yh);
        ++st.n_psi;
        while (curr.L < P.L_max && qub_violated(curr)) {
            curr.gamma = curr.gamma / 2.0;
            curr.L     = curr.L * 2.0;
            pb.prox_grad_step(curr.gamma, X_(curr), G_(curr), XH_(curr), P_(curr), curr.pTp, curr.gradTp);
            ++st.n_prox_grad_step;
            curr.psi_xh = pb.eval_psi_xhat(XH_(curr), curr.yh);
            ++st.n_psi;
            ++st.qub_backtracks;
        }
    }

    while (status == Busy) {
        // ---- stop criterion ----
        if (P.stop_crit == ProjGradUnitNorm) {
            eps_k = pb.proj_grad_unit_norm(X_(curr), G_(curr));
            ++st.n_prox_grad_step;
        } else if (P.stop_crit == FPRNorm) {
            eps_k = pb.norm_inf(P_(curr)) / curr.gamma;
        } else {
            eps_k = __builtin_nan("");
        }
        if (trace && k < trace_cap && lane == 0) {
            double *o = trace + (size_t)k * TRACE_COLS;
            o[0] = eps_k; o[1] = curr.psi_x; o[2] = __builtin_sqrt(curr.pTp); o[3] = curr.gamma;
            o[4] = tau; o[5] = curr.fbe(); o[6] = lb.idx + (lb.full ? lb.memory : 0);
        }
        double elapsed   = (double)(wall_clock64() - t_start) * P.tick_s;
        bool converged   = eps_k <= tolerance;
        bool out_of_time = elapsed > deadline_s;
        bool out_of_iter = k == P.max_iter;
        bool not_finite  = !is_finite(eps_k);
        bool stalled     = no_progress > P.max_no_progress;
        status = converged ? Converged : out_of_time ? MaxTime : out_of_iter ? MaxIter
                 : not_finite ? NotFinite : stalled ? NoProgress : Busy;
        status = uniform(status);
        if (status != Busy)
            break;

        // ---- quasi-Newton direction ----
        double tau_init = 0;
        if (k > 0) {
            bool ok;
            if (STRUCTURED) {
                int nJ        = 0;
                uint32_t mask = pb.inactive_mask(curr.gamma, X_(curr), G_(curr), nJ);
                for (int j = 0; j < nj; ++j) {
                    int i = j * WAVE + lane;
                    if (i < n)
                        q[i] = (mask >> j) & 1u ? -G_(curr)[i] : P_(curr)[i];
                }
                PANTR_PHASE(PH_DIRECTION, ok = nJ > 0 && lbfgs_apply(q, nJ == n ? ~0u : mask, nJ != n));
            } else {
                for (int j = 0; j < nj; ++j) {
                    int i = j * WAVE + lane;
                    if (i < n)
                        q[i] = P_(curr)[i];
                }
                PANTR_PHASE(PH_DIRECTION, ok = lbfgs_apply(q, ~0u, false));
            }
            bool finite = true;
            for (int j = 0; j < nj; ++j) {
                int i = j * WAVE + lane;
                if (i < n)
                    finite = finite && is_finite(q[i]);
            }
            bool all_finite = __builtin_amdgcn_readfirstlane(__all(finite ? 1 : 0)) != 0;
            tau_init        = ok && all_finite ? 1 : 0;
            if (tau_init != 1) {
                ++st.lbfgs_failures;
                lb.reset();
            }
        }

        // ---- line search on the forward-backward envelope ----
        next.gamma      = curr.gamma;
        next.L          = curr.L;
        tau             = tau_init;
        double tau_prev = -1;
        while (true) {
            if (tau != tau_prev) {
                // x+ = x + (1 - tau) p + tau q, or xhat for tau = 0 (there psi and the multipliers are
                // those of xhat again: the evaluation below reproduces them bit for bit)
                for (int j = 0; j < nj; ++j) {
                    int i = j * WAVE + lane;
                    if (i < n) {
                        double xi = X_(curr)[i];
                        X_(next)[i] = tau == 0   ? XH_(curr)[i]
                                      : tau == 1 ? xi + q[i]
                                                 : (xi + (1.0 - tau) * P_(curr)[i]) + tau * q[i];
                    }
                }
                PANTR_PHASE(PH_PSI_GRAD, next.psi_x = pb.eval_psi_grad(X_(next), G_(next)));
                if (tau != 0)
                    ++st.n_psi_grad_psi;
                else
                    ++st.n_grad_L;
                tau_prev = tau;
            }
            bool fail = next.L >= P.L_max || !is_finite(next.psi_x);
            if (tau > 0 && fail) {
                next.L     = curr.L;
                next.gamma = curr.gamma;
                tau        = 0;
                lb.reset();
                continue;
            }
            pb.prox_grad_step(next.gamma, X_(next), G_(next), XH_(next), P_(next), next.pTp, next.gradTp);
            ++st.n_prox_grad_step;
            PANTR_PHASE(PH_PSI_XHAT, next.psi_xh = pb.eval_psi_xhat(XH_(next), next.yh));
            ++st.n_psi;
            if (next.L < P.L_max && qub_violated(next)) {
                next.gamma = next.gamma / 2.0;
                next.L     = next.L * 2.0;
                if (tau > 0)
                    tau = tau_init;
                ++st.qub_backtracks;
                continue;
            }
            if (tau > 0 && !P.force_linesearch) {
                double sigma  = (P.linesearch_strictness_factor * (1.0 - curr.gamma * curr.L)) / (2.0 * curr.gamma);
                double phi    = curr.fbe();
                double margin = (1.0 + fabs(phi)) * P.linesearch_tolerance_factor;
                if (next.fbe() > (phi - sigma * curr.pTp) + margin) {
                    tau = tau / 2.0;
                    if (tau < P.min_linesearch_coefficient)
                        tau = 0;
                    ++st.linesearch_backtracks;
                    continue;
                }
            }
            break;
        }
        st.linesearch_failures += tau == 0 && tau_init > 0;
        st.tau_1_accepted += tau == 1;
        st.count_tau += tau_init > 0;
        st.sum_tau += tau_init > 0 ? tau : 0.0;

        if (no_progress > 0 || k % P.max_no_progress == 0) {
            bool same = true;
            for (int j = 0; j < nj; ++j) {
                int i = j * WAVE + lane;
                if (i < n)
                    same = same && X_(curr)[i] == X_(next)[i];
            }
            bool all_same = __builtin_amdgcn_readfirstlane(__all(same ? 1 : 0)) != 0;
            no_progress   = all_same ? no_progress + 1 : 0;
        }
        bool accepted;
        if (STRUCTURED) {
            PANTR_PHASE(PH_GRAD_XHAT, accepted = lbfgs_update(X_(curr), X_(next), G_(curr), G_(next), true, true));
        } else {
            if (curr.gamma != next.gamma)
                lb.reset();
            PANTR_PHASE(PH_GRAD_XHAT, accepted = lbfgs_update(X_(curr), X_(next), P_(curr), P_(next), false, false));
        }
        st.lbfgs_rejected += accepted ? 0 : 1;
        swap_iter(curr, next);
        ++k;
    }

    // ---- results ----
    if ((status == Converged || status == Interrupted || P.always_overwrite_results) && is_finite(curr.L)) {
        pb.update_multipliers(curr.yh);
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + lane;
            if (i < n)
                x_io[i] = XH_(curr)[i];
        }
        wave_sync();
    }
    return {status, k, eps_k, curr.gamma, curr.psi_xh, curr.fbe(), __builtin_nan("")};
}

/// pantr_inner() or panoc_inner() by method (0 = pantr, 1 = panoc, 2 = strucpanoc).
template <int METHOD, class PB>
__device__ __forceinline__ InnerResult inner_solve(PB &pb, const KernelParams &P, double tolerance,
                                                   double deadline_s, long long t_start, double *x_io,
                                                   DeviceStats &st, double *trace, int trace_cap,
                                                   double *lbfgs_ws) {
    if constexpr (METHOD == 0)
        return pantr_inner(pb, P, tolerance, deadline_s, t_start, x_io, st, trace, trace_cap);
    else
        return panoc_inner<PB, METHOD == 2>(pb, P, tolerance, deadline_s, t_start, x_io, st, trace, trace_cap,
                                            lbfgs_ws);
}

} // namespace pantr
