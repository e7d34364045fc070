// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/README.md).
//
// Plain C++ restatement of the comparator inner solvers of the reference's benchmark
// (SURVEY.md section 8 row f4):
//   method=panoc ....... alpaqa::PANOCSolver<LBFGSDirection>            drivers/mpc-driver.cpp:378-384
//   method=strucpanoc .. alpaqa::PANOCSolver<StructuredLBFGSDirection>  drivers/mpc-driver.cpp:371-377
//   options ............ accel.memory=50, dir.hessian_vec=0             new-benchmarks-paper/dodo.py:34-40
// The algorithms live in the un-vendored submodule kul-optec/alpaqa (branch "pantr", pinned commit
// unrecoverable), so they are restated from the published method (Stella, Themelis, Sopasakis,
// Patrinos: "A simple and efficient algorithm for nonlinear model predictive control", CDC 2017;
// De Marchi, Themelis: "Proximal gradient algorithms under local Lipschitz gradient continuity",
// JOTA 2022, for the adaptive step size that makes it PANOC+; Pas, Schuurmans, Patrinos: "Alpaqa",
// ECC 2022, for the structured direction) and from upstream alpaqa's public structure
// (implementation/inner/panoc.tpp, inner/directions/panoc/lbfgs.hpp, structured-lbfgs.tpp,
// accelerators/lbfgs.hpp).  Every reduction follows the kernels' lane order (chain_model.hpp: dot).
#pragma once
#include "pantr_oracle.hpp"

namespace oracle {

/// alpaqa::PANOCParams defaults (mpc-driver.cpp:51-53 then overrides max_iter and stop_crit).
struct PANOCParams {
    double L_0 = 0, lip_eps = 1e-6, lip_delta = 1e-12, Lgamma_factor = 0.95; // Lipschitz.*
    int max_iter      = 100;
    double max_time_s = 300;
    double min_linesearch_coefficient   = 1.0 / 256;
    int force_linesearch                = 0;
    double linesearch_strictness_factor = 0.95;
    double L_min = 1e-5, L_max = 1e20;
    int stop_crit       = ApproxKKT;
    int max_no_progress = 10;
    double quadratic_upperbound_tolerance_factor = 10 * eps_mach;
    double linesearch_tolerance_factor           = 10 * eps_mach;
};
/// alpaqa::LBFGSParams ("accel.*"; dodo.py:35,38 set memory=50)
struct LBFGSParams {
    int memory         = 10;
    double min_div_fac = 1e-10, min_abs_s = 1e-32;
    double cbfgs_alpha = 1, cbfgs_eps = 0; ///< cautious-BFGS condition, inactive when eps = 0
    int force_pos_def  = 1;
};

/// Limited-memory BFGS inverse-Hessian approximation (alpaqa::LBFGS), step size from the most
/// recent curvature pair (LBFGSStepSize::BasedOnCurvature, the default).
struct LBFGS {
    LBFGSParams params;
    int n = 0, idx = 0;
    bool full = false;
    std::vector<std::vector<double>> S, Y;
    std::vector<double> rho, alpha;

    void resize(int n_) {
        n = n_;
        S.assign(params.memory, std::vector<double>(n));
        Y.assign(params.memory, std::vector<double>(n));
        rho.assign(params.memory, NaN);
        alpha.assign(params.memory, NaN);
        reset();
    }
    void reset() {
        idx  = 0;
        full = false;
    }
    int succ(int i) const { return i + 1 < params.memory ? i + 1 : 0; }
    int pred(int i) const { return i > 0 ? i - 1 : params.memory - 1; }
    template <class F>
    void foreach_rev(F &&f) const { // newest to oldest
        for (int i = idx - 1; i >= 0; --i)
            f(i);
        if (full)
            for (int i = params.memory - 1; i >= idx; --i)
                f(i);
    }
    template <class F>
    void foreach_fwd(F &&f) const { // oldest to newest
        if (full)
            for (int i = idx; i < params.memory; ++i)
                f(i);
        for (int i = 0; i < idx; ++i)
            f(i);
    }
    bool update_valid(double yTs, double sTs, double pTp) const {
        if (sTs < params.min_abs_s)
            return false;
        if (!std::isfinite(yTs))
            return false;
        double a_yTs = params.force_pos_def ? yTs : std::fabs(yTs);
        if (a_yTs < params.min_div_fac * sTs)
            return false;
        if (params.cbfgs_eps > 0) { // y's / s's >= eps |p|^alpha
            if (!(a_yTs >= (sTs * params.cbfgs_eps) * std::pow(pTp, params.cbfgs_alpha / 2)))
                return false;
        }
        return true;
    }
    /// s = xn - x, y = sign (pn - p); pn: the vector of the cautious-BFGS condition.
    bool update(const double *x, const double *xn, const double *p, const double *pn, bool positive,
                bool forced) {
        std::vector<double> &s = S[idx], &y = Y[idx];
        std::vector<double> s_keep, y_keep;
        if (!forced) { // the slot is only overwritten when the pair is accepted
            s_keep = s;
            y_keep = y;
        }
        for (int i = 0; i < n; ++i) {
            s[i] = xn[i] - x[i];
            y[i] = positive ? pn[i] - p[i] : p[i] - pn[i];
        }
        double yTs = dot(y.data(), s.data(), n);
        if (!forced) {
            double sTs = dot(s.data(), s.data(), n);
            double pTp = params.cbfgs_eps > 0 ? dot(pn, pn, n) : 0.0;
            if (!update_valid(yTs, sTs, pTp)) {
                s = s_keep;
                y = y_keep;
                return false;
            }
        }
        rho[idx] = 1.0 / yTs;
        idx      = succ(idx);
        full     = full || idx == 0;
        return true;
    }
    /// q <- H q (two-loop recursion).
    bool apply(double *q) {
        if (idx == 0 && !full)
            return false;
        int newest   = pred(idx);
        double yTy   = dot(Y[newest].data(), Y[newest].data(), n);
        double gamma = 1.0 / (rho[newest] * yTy);
        foreach_rev([&](int i) {
            alpha[i] = rho[i] * dot(S[i].data(), q, n);
            for (int j = 0; j < n; ++j)
                q[j] = q[j] - alpha[i] * Y[i][j];
        });
        for (int j = 0; j < n; ++j)
            q[j] = gamma * q[j];
        foreach_fwd([&](int i) {
            double beta = rho[i] * dot(Y[i].data(), q, n);
            double c    = beta - alpha[i];
            for (int j = 0; j < n; ++j)
                q[j] = q[j] - c * S[i][j];
        });
        return true;
    }
    /// The same on the index set J (mask[i] = 1 on J); q is only touched on J.  The curvature of
    /// every pair is recomputed on J (and kept), pairs with non-positive curvature are skipped.
    bool apply_masked(double *q, const unsigned char *mask, int nJ) {
        if (idx == 0 && !full)
            return false;
        if (nJ == n)
            return apply(q);
        auto dotJ = [&](const double *a, const double *b) {
            double lane[64] = {};
            for (int i = 0; i < n; ++i)
                lane[i % 64] = lane[i % 64] + a[i] * (mask[i] ? b[i] : 0.0);
            return wave_sum(lane);
        };
        double gamma = -1;
        foreach_rev([&](int i) {
            rho[i] = 1.0 / dotJ(S[i].data(), Y[i].data());
            if (!(rho[i] > 0))
                return;
            alpha[i] = rho[i] * dotJ(S[i].data(), q);
            for (int j = 0; j < n; ++j)
                if (mask[j])
                    q[j] = q[j] - alpha[i] * Y[i][j];
            if (gamma < 0)
                gamma = 1.0 / (rho[i] * dotJ(Y[i].data(), Y[i].data()));
        });
        if (gamma < 0)
            return false;
        for (int j = 0; j < n; ++j)
            if (mask[j])
                q[j] = gamma * q[j];
        foreach_fwd([&](int i) {
            if (!(rho[i] > 0))
                return;
            double beta = rho[i] * dotJ(Y[i].data(), q);
            double c    = beta - alpha[i];
            for (int j = 0; j < n; ++j)
                if (mask[j])
                    q[j] = q[j] - c * S[i][j];
        });
        return true;
    }
};

/// alpaqa::LBFGSDirection: q = H p with pairs (dx, -dp); the memory is dropped when gamma changes.
struct LBFGSDirection {
    static constexpr bool structured = false;
    LBFGS lbfgs;
    template <class Problem>
    void initialize(const Problem &problem) { lbfgs.resize(problem.n()); }
    template <class Problem>
    bool apply(const Problem &, double, const double *, const double *p, const double *, double *q, int n) {
        std::copy(p, p + n, q);
        return lbfgs.apply(q);
    }
    bool update(const double *x, const double *xn, const double *p, const double *pn, const double *,
                const double *) {
        return lbfgs.update(x, xn, p, pn, /*positive=*/false, /*forced=*/false);
    }
    void changed_gamma() { lbfgs.reset(); }
    void reset() { lbfgs.reset(); }
};

/// alpaqa::StructuredLBFGSDirection with hessian_vec_factor = 0 (dodo.py:39): the prox step on the
/// active constraints, L-BFGS (pairs (dx, dgrad psi), always stored) on the inactive set J.
struct StructuredLBFGSDirection {
    static constexpr bool structured = true;
    LBFGS lbfgs;
    std::vector<unsigned char> mask;
    template <class Problem>
    void initialize(const Problem &problem) {
        lbfgs.resize(problem.n());
        mask.assign(problem.n(), 0);
    }
    template <class Problem>
    bool apply(const Problem &problem, double gamma, const double *x, const double *p, const double *grad,
               double *q, int n) {
        int nJ = 0;
        for (int i = 0; i < n; ++i) {
            double gd = x[i] - gamma * grad[i];
            bool in   = problem.lb(i) < gd && gd < problem.ub(i);
            mask[i]   = in;
            nJ += in;
            q[i] = in ? -grad[i] : p[i];
        }
        if (nJ == 0)
            return false; // no free variable: the projected-gradient step is taken
        return lbfgs.apply_masked(q, mask.data(), nJ);
    }
    bool update(const double *x, const double *xn, const double *, const double *, const double *grad,
                const double *gradn) {
        return lbfgs.update(x, xn, grad, gradn, /*positive=*/true, /*forced=*/true);
    }
    void changed_gamma() {} // the pairs do not depend on the step size
    void reset() { lbfgs.reset(); }
};

/// One record per PANOC iteration (iterate-parity tests).
struct PanocTraceRow {
    double eps, psi, pnorm, gamma, tau, fbe;
    int lbfgs_idx;
};

template <class Problem, class Direction>
struct PANOCSolver {
    PANOCParams params;
    Direction direction;
    mutable std::vector<PanocTraceRow> *trace = nullptr;

    struct Iterate {
        std::vector<double> x, xh, grad, p, yh;
        double psi_x = NaN, psi_xh = NaN, gamma = NaN, L = NaN, pTp = NaN, gradTp = NaN;
        Iterate(int n, int m) : x(n), xh(n), grad(n), p(n), yh(m) {}
        double fbe() const { return (psi_x + pTp / (2.0 * gamma)) + gradTp; }
    };

    /// Same calling convention as PANTRSolver::operator().
    Stats operator()(Problem &problem, double tolerance, double max_time_s, bool always_overwrite_results,
                     double *x, double *y = nullptr, const double *Sigma = nullptr,
                     double *err_z = nullptr) const {
        using clk = std::chrono::steady_clock;
        auto t0   = clk::now();
        Direction dir = direction; // fresh memory per solve
        const int n = problem.n(), m = problem.m();
        problem.set_al(y, Sigma);
        Stats s;
        const EvalCounters c0 = problem.cnt;
        Iterate it0(n, m), it1(n, m);
        Iterate *curr = &it0, *next = &it1;
        std::vector<double> q(n), work(n), work2(n);

        auto prox_grad_step = [&](Iterate &i) {
            problem.eval_prox_grad_step(i.gamma, i.x.data(), i.grad.data(), i.xh.data(), i.p.data());
            i.pTp    = dot(i.p.data(), i.p.data(), n);
            i.gradTp = dot(i.p.data(), i.grad.data(), n);
        };
        auto eval_psi_xh = [&](Iterate &i) { i.psi_xh = problem.eval_psi(i.xh.data(), i.yh.data()); };
        auto qub_violated = [&](const Iterate &i) {
            double margin = (1.0 + std::fabs(i.psi_x)) * params.quadratic_upperbound_tolerance_factor;
            return i.psi_xh > ((i.psi_x + i.gradTp) + (0.5 * i.L) * i.pTp) + margin;
        };
        auto linesearch_violated = [&](const Iterate &c, const Iterate &nx) {
            if (params.force_linesearch)
                return false;
            double sigma  = (params.linesearch_strictness_factor * (1.0 - c.gamma * c.L)) / (2.0 * c.gamma);
            double phi    = c.fbe();
            double margin = (1.0 + std::fabs(phi)) * params.linesearch_tolerance_factor;
            return nx.fbe() > (phi - sigma * c.pTp) + margin;
        };
        auto finish = [&](int status, double eps_k, int k) {
            if (status == Converged || status == Interrupted || always_overwrite_results) {
                for (int j = 0; j < m; ++j) {
                    err_z[j] = (curr->yh[j] - y[j]) / Sigma[j];
                    y[j]     = curr->yh[j];
                }
                std::copy(curr->xh.begin(), curr->xh.end(), x);
            }
            s.status       = status;
            s.iterations   = k;
            s.eps          = eps_k;
            s.final_gamma  = curr->gamma;
            s.final_psi    = curr->psi_xh;
            s.final_fbe    = curr->fbe();
            s.final_radius = NaN;
            const EvalCounters &c = problem.cnt;
            s.n_psi            = c.psi - c0.psi;
            s.n_grad_psi       = c.grad_psi - c0.grad_psi;
            s.n_psi_grad_psi   = c.psi_grad_psi - c0.psi_grad_psi;
            s.n_grad_L         = c.grad_L - c0.grad_L;
            s.n_hess_psi_prod  = c.hess_psi_prod - c0.hess_psi_prod;
            s.n_prox_grad_step = c.prox_grad_step - c0.prox_grad_step;
            s.elapsed_s        = std::chrono::duration<double>(clk::now() - t0).count();
            return s;
        };

        // ---- initial iterate and Lipschitz estimate (as PANTRSolver) ----
        std::copy(x, x + n, curr->x.begin());
        if (params.L_0 <= 0) {
            curr->psi_x = problem.eval_psi_grad_psi(curr->x.data(), curr->grad.data());
            for (int i = 0; i < n; ++i) {
                double g = curr->grad[i];
                double h = g > 0 ? std::fmax(g * params.lip_eps, params.lip_delta)
                                 : std::fmin(g * params.lip_eps, -params.lip_delta);
                work[i]     = h;
                curr->xh[i] = curr->x[i] - h;
            }
            double norm_h = std::sqrt(dot(work.data(), work.data(), n));
            problem.eval_grad_psi(curr->xh.data(), next->grad.data());
            for (int i = 0; i < n; ++i)
                work[i] = next->grad[i] - curr->grad[i];
            double L = std::sqrt(dot(work.data(), work.data(), n)) / norm_h;
            curr->L  = std::fmin(std::fmax(L, params.L_min), params.L_max);
        } else {
            curr->L     = params.L_0;
            curr->psi_x = problem.eval_psi_grad_psi(curr->x.data(), curr->grad.data());
        }
        if (!std::isfinite(curr->L)) {
            s.status = NotFinite;
            return s;
        }
        curr->gamma = params.Lgamma_factor / curr->L;
        prox_grad_step(*curr);
        eval_psi_xh(*curr);
        while (curr->L < params.L_max && qub_violated(*curr)) {
            curr->gamma = curr->gamma / 2.0;
            curr->L     = curr->L * 2.0;
            prox_grad_step(*curr);
            eval_psi_xh(*curr);
            ++s.qub_backtracks;
        }
        dir.initialize(problem);

        int k = 0, no_progress = 0;
        double tau = NaN;
        while (true) {
            // ---- stop criterion ----
            double eps_k;
            switch (params.stop_crit) {
            case ProjGradUnitNorm:
                problem.eval_prox_grad_step(1.0, curr->x.data(), curr->grad.data(), work.data(), work2.data());
                eps_k = norm_inf(work2.data(), n);
                break;
            case FPRNorm: eps_k = norm_inf(curr->p.data(), n) / curr->gamma; break;
            default: eps_k = NaN; break;
            }
            if (trace)
                trace->push_back({eps_k, curr->psi_x, std::sqrt(curr->pTp), curr->gamma, tau, curr->fbe(),
                                  dir.lbfgs.idx + (dir.lbfgs.full ? dir.lbfgs.params.memory : 0)});
            double elapsed   = std::chrono::duration<double>(clk::now() - t0).count();
            bool converged   = eps_k <= tolerance;
            bool out_of_time = elapsed > std::fmin(params.max_time_s, max_time_s);
            bool out_of_iter = k == params.max_iter;
            bool not_finite  = !std::isfinite(eps_k);
            bool stalled     = no_progress > params.max_no_progress;
            int status = converged ? Converged : out_of_time ? MaxTime : out_of_iter ? MaxIter
                         : not_finite ? NotFinite : stalled ? NoProgress : Busy;
            if (status != Busy)
                return finish(status, eps_k, k);

            // ---- quasi-Newton direction ----
            double tau_init = 0;
            if (k > 0) {
                bool ok = dir.apply(problem, curr->gamma, curr->x.data(), curr->p.data(), curr->grad.data(),
                                    q.data(), n);
                bool finite = true;
                for (int i = 0; i < n; ++i)
                    finite = finite && std::isfinite(q[i]);
                tau_init = ok && finite ? 1 : 0;
                if (tau_init != 1) {
                    ++s.lbfgs_failures;
                    dir.reset();
                }
            }

            // ---- line search on the forward-backward envelope ----
            next->gamma = curr->gamma;
            next->L     = curr->L;
            tau         = tau_init;
            double tau_prev = -1;
            while (true) {
                if (tau != tau_prev) {
                    if (tau != 0) { // x+ = x + (1 - tau) p + tau q
                        for (int i = 0; i < n; ++i)
                            next->x[i] = tau == 1 ? curr->x[i] + q[i]
                                                  : (curr->x[i] + (1.0 - tau) * curr->p[i]) + tau * q[i];
                        next->psi_x = problem.eval_psi_grad_psi(next->x.data(), next->grad.data());
                    } else { // x+ = xhat: psi known, gradient with the multipliers of xhat
                        next->x     = curr->xh;
                        next->psi_x = curr->psi_xh;
                        problem.eval_grad_L(next->x.data(), next->grad.data(), /*record=*/false);
                    }
                    tau_prev = tau;
                }
                bool fail = next->L >= params.L_max || !std::isfinite(next->psi_x);
                if (tau > 0 && fail) { // abandon the direction, keep the step size of the current iterate
                    next->L     = curr->L;
                    next->gamma = curr->gamma;
                    tau         = 0;
                    dir.reset();
                    continue;
                }
                prox_grad_step(*next);
                eval_psi_xh(*next);
                if (next->L < params.L_max && qub_violated(*next)) {
                    next->gamma = next->gamma / 2.0;
                    next->L     = next->L * 2.0;
                    if (tau > 0)
                        tau = tau_init;
                    ++s.qub_backtracks;
                    continue;
                }
                if (tau > 0 && linesearch_violated(*curr, *next)) {
                    tau = tau / 2.0;
                    if (tau < params.min_linesearch_coefficient)
                        tau = 0;
                    ++s.linesearch_backtracks;
                    continue;
                }
                break;
            }
            s.linesearch_failures += tau == 0 && tau_init > 0;
            s.tau_1_accepted += tau == 1;
            s.count_tau += tau_init > 0;
            s.sum_tau += tau_init > 0 ? tau : 0.0;

            if (no_progress > 0 || k % params.max_no_progress == 0) {
                bool same = true;
                for (int i = 0; i < n; ++i)
                    same = same && curr->x[i] == next->x[i];
                no_progress = same ? no_progress + 1 : 0;
            }
            if (curr->gamma != next->gamma)
                dir.changed_gamma();
            s.lbfgs_rejected += !dir.update(curr->x.data(), next->x.data(), curr->p.data(), next->p.data(),
                                            curr->grad.data(), next->grad.data());
            std::swap(curr, next);
            ++k;
        }
    }
};

} // namespace oracle
