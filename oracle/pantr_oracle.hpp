// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/README.md).
//
// Plain C++ restatement of the PANTR inner solver, its Newton trust-region
// direction, the Steihaug-Toint CG subproblem solver and the (m = 0) ALM shell.
// These live in the un-vendored submodule kul-optec/alpaqa (.gitmodules:1-4,
// branch "pantr", pinned commit unrecoverable), so they are restated from the
// published algorithm (Bodard, Pas, Patrinos: "PANTR: A proximal algorithm with
// trust-region updates for nonconvex constrained optimization", arXiv 2306.17119)
// and from upstream alpaqa's public structure (include/alpaqa/inner/pantr.hpp,
// implementation/inner/pantr.tpp, inner/directions/pantr/newton-tr.hpp,
// accelerators/steihaugcg.hpp, implementation/inner/panoc-helpers.tpp,
// implementation/outer/alm.tpp), anchored on the reference's own call sites:
//   solver construction ...... drivers/mpc-driver.cpp:47-73,104-115,364-370
//   solve call / status ...... drivers/mpc-driver.cpp:216-222
//   option names and values .. new-benchmarks-paper/dodo.py:12-33,94-105
//   stop criterion semantics . drivers/results.hpp:268-271
// The Steihaug CG loop has the structure of SciPy's CGSteihaugSubproblem
// (scipy/optimize/_trustregion_ncg.py:44-126), which tests/ use as an
// independent cross-check of that subroutine.
#pragma once
#include "chain_model.hpp"
#include <chrono>
#include <limits>
#include <utility>

namespace oracle {

inline constexpr double eps_mach = std::numeric_limits<double>::epsilon();
inline constexpr double NaN      = std::numeric_limits<double>::quiet_NaN();
inline constexpr double inf      = std::numeric_limits<double>::infinity();

/// alpaqa::SolverStatus values (printed by mpc-driver.cpp:218)
enum Status : int {
    Busy = 0, Converged = 1, MaxTime = 2, MaxIter = 3, NotFinite = 4, NoProgress = 5,
    Interrupted = 6, Exception = 7,
};
/// alpaqa::PANOCStopCrit (mpc-driver.cpp:53, dodo.py:19)
enum StopCrit : int { ApproxKKT = 0, ApproxKKT2, ProjGradNorm, ProjGradNorm2, ProjGradUnitNorm,
                      ProjGradUnitNorm2, FPRNorm, FPRNorm2, Ipopt, LBFGSBpp };

/// PANTRParams: alpaqa defaults, then mpc-driver.cpp:51-53 overrides
/// (max_iter 50000, stop_crit FPRNorm) are applied by the caller.
struct PANTRParams {
    double L_0 = 0, lip_eps = 1e-6, lip_delta = 1e-12, Lgamma_factor = 0.95; // Lipschitz.*
    int max_iter      = 100;
    double max_time_s = 300;
    double L_min = 1e-5, L_max = 1e20;
    int stop_crit       = ApproxKKT;
    int max_no_progress = 10;
    double quadratic_upperbound_tolerance_factor = 10 * eps_mach;
    double TR_tolerance_factor                   = 10 * eps_mach;
    double ratio_threshold_acceptable = 0.2;  // dodo.py:29 sets 0.3
    double ratio_threshold_good       = 0.8;  // dodo.py:30 sets 0.5
    double radius_factor_rejected     = 0.35; // dodo.py:26
    double radius_factor_acceptable   = 0.999; // dodo.py:27 sets 0.99
    double radius_factor_good         = 2.5;  // dodo.py:28 sets 10
    double initial_radius = NaN;
    double min_radius     = 100 * eps_mach;
    int disable_acceleration = 0;
    /// Variants of the six places where recollections of upstream alpaqa differ, for
    /// scripts/adjudicate_choices.py only - 0 is the specification the kernels follow.  Bits:
    ///   1  Lipschitz estimate: finite-difference step x + h instead of x - h
    ///   2  initial trust radius ||grad psi(x0)|| instead of 0.1 ||grad psi(x0)||
    ///   4  q_K = 0 (the active set does not move with the Newton step) instead of q_K = p_K
    ///   8  the model value the ratio test divides by also counts the active set's part of the step:
    ///      + grad_K' p_K + 1/2 p_K' H_KK p_K + p_K' H_KJ q_J  (one more product)
    int variant = 0;
};
struct NewtonTRDirectionParams {
    double hessian_vec_factor = 0.5; // dodo.py:31 sets 1
};
struct SteihaugCGParams {
    double tol_scale = 1, tol_scale_root = 0.5, tol_max = inf, max_iter_factor = 1; // dodo.py:32,103
};

struct Stats {
    int status = Busy;
    int iterations = 0, cg_iterations = 0, accepted = 0, rejected = 0, qub_backtracks = 0;
    int qub_at_prox = 0; ///< of those, the backtracking steps taken at a prox iterate xbar (adjudication aid)
    long n_psi = 0, n_grad_psi = 0, n_psi_grad_psi = 0, n_grad_L = 0, n_hess_psi_prod = 0,
         n_prox_grad_step = 0;
    /// n_hess_psi_prod + the products with q_K = 0 the direction skips: what alpaqa's EvalCounter would show
    /// (drivers/results.hpp:63-87 writes it, benchmark_util.py:12-27 sums it)
    long n_hess_psi_prod_calls = 0;
    double eps = NaN, final_gamma = NaN, final_psi = NaN, final_fbe = NaN, final_radius = NaN;
    double elapsed_s = 0;
    // ALM
    int outer_iterations = 1, inner_convergence_failures = 0;
    double delta = 0, penalty_norm = 0;
    // PANOC (panoc_oracle.hpp)
    int linesearch_failures = 0, linesearch_backtracks = 0, lbfgs_failures = 0, lbfgs_rejected = 0,
        tau_1_accepted = 0, count_tau = 0;
    double sum_tau = 0;
};

/// One record per PANTR iteration, used by the iterate-parity tests.
struct TraceRow {
    double eps, psi, pnorm, gamma, radius, rho, q_model, fbe;
    int nJ, cg_its, accepted;
};

// ---------------------------------------------------------------------------
// Steihaug-Toint CG on the index set J, vectors stored full length with zeros
// outside J (so every reduction has the kernel's lane order).
// ---------------------------------------------------------------------------
inline std::pair<double, double> boundaries_intersections(double zz, double zd, double dd,
                                                          double radius) {
    // Solve ||z + t d|| = radius: a t^2 + b t + c = 0, cancellation-free roots
    double a = dd, b = 2.0 * zd, c = zz - radius * radius;
    double sqd = std::sqrt(b * b - (4.0 * a) * c);
    double aux = b + std::copysign(sqd, b);
    double ta = (-aux) / (2.0 * a), tb = (-(2.0 * c)) / aux;
    return {std::fmin(ta, tb), std::fmax(ta, tb)};
}

template <class HessProd>
double steihaug_solve(const SteihaugCGParams &prm, int n, int nJ, const double *grad /*r_J*/,
                      HessProd &&hess_prod, double radius, double *step,
                      std::vector<double> &wz, std::vector<double> &wr, std::vector<double> &wd,
                      std::vector<double> &wBd, std::vector<double> &wev, int &cg_its) {
    double *z = wz.data(), *r = wr.data(), *d = wd.data(), *Bd = wBd.data(), *ev = wev.data();
    for (int i = 0; i < n; ++i) {
        z[i] = 0.0;
        r[i] = grad[i];
        d[i] = -grad[i];
    }
    double r_sq     = dot(r, r, n);
    double grad_mag = std::sqrt(r_sq);
    cg_its          = 0;
    if (!(grad_mag > 0.0)) { // zero (or NaN) model gradient: no step
        std::fill(step, step + n, 0.0);
        return grad_mag == 0.0 ? 0.0 : NaN;
    }
    // tol_scale_root = 1/2 (the only value the reference uses) goes through sqrt, which is
    // correctly rounded on every platform; other exponents fall back to pow.
    double rooted    = prm.tol_scale_root == 0.5 ? std::sqrt(grad_mag)
                                                 : std::pow(grad_mag, prm.tol_scale_root);
    double tolerance = std::fmin(prm.tol_max, (prm.tol_scale * grad_mag) * std::fmin(1.0, rooted));
    auto eval = [&](const double *p) {
        hess_prod(p, ev);
        return dot(p, grad, n) + 0.5 * dot(p, ev, n);
    };
    const int max_iter = static_cast<int>(std::lround(static_cast<double>(nJ) * prm.max_iter_factor));
    int i = 0;
    while (true) {
        hess_prod(d, Bd);
        ++cg_its;
        double dBd = dot(d, Bd, n);
        if (dBd <= 0.0) {
            // negative curvature: both boundary points along d, keep the lower model value
            auto [ta, tb] = boundaries_intersections(dot(z, z, n), dot(z, d, n), dot(d, d, n), radius);
            for (int j = 0; j < n; ++j) {
                double pa = z[j] + ta * d[j], pb = z[j] + tb * d[j];
                r[j] = pa;
                d[j] = pb;
            }
            double qa = eval(r), qb = eval(d);
            const double *best = qa < qb ? r : d;
            std::copy(best, best + n, step);
            return qa < qb ? qa : qb;
        }
        double alpha = r_sq / dBd;
        for (int j = 0; j < n; ++j)
            step[j] = z[j] + alpha * d[j]; // z_next
        if (std::sqrt(dot(step, step, n)) >= radius) {
            auto [ta, tb] = boundaries_intersections(dot(z, z, n), dot(z, d, n), dot(d, d, n), radius);
            (void)ta;
            for (int j = 0; j < n; ++j)
                step[j] = z[j] + tb * d[j];
            return eval(step);
        }
        for (int j = 0; j < n; ++j)
            r[j] = r[j] + alpha * Bd[j];
        double r_next_sq = dot(r, r, n);
        if (std::sqrt(r_next_sq) < tolerance || i > max_iter)
            return eval(step);
        double beta = r_next_sq / r_sq;
        r_sq        = r_next_sq;
        for (int j = 0; j < n; ++j) {
            d[j] = (-r[j]) + beta * d[j];
            z[j] = step[j];
        }
        ++i;
    }
}

// ---------------------------------------------------------------------------
// PANTR
// ---------------------------------------------------------------------------
template <class Problem>
struct PANTRSolver {
    PANTRParams params;
    NewtonTRDirectionParams dir_params;
    SteihaugCGParams accel_params;
    std::vector<TraceRow> *trace = nullptr;

    struct Iterate {
        std::vector<double> x, xh, grad, p, yh; ///< yh = candidate multipliers yhat(xhat)
        double psi_x = NaN, psi_xh = NaN, gamma = NaN, L = NaN, pTp = NaN, gradTp = NaN;
        Iterate(int n, int m) : x(n), xh(n), grad(n), p(n), yh(m) {}
        /// forward-backward envelope (h(xhat) = 0 for a box)
        double fbe() const { return (psi_x + pTp / (2.0 * gamma)) + gradTp; }
    };

    /// x in/out (n); y in/out and Sigma (m) the multipliers/penalties of the ALM iteration
    /// (nullptr when m = 0); err_z out (m): Sigma^{-1} (yhat - y).
    Stats operator()(Problem &problem, double tolerance, double max_time_s,
                     bool always_overwrite_results, double *x, double *y = nullptr,
                     const double *Sigma = nullptr, double *err_z = nullptr) const {
        using clk = std::chrono::steady_clock;
        auto t0   = clk::now();
        const int n = problem.n(), m = problem.m();
        problem.set_al(y, Sigma);
        Stats s;
        const EvalCounters c0 = problem.cnt;
        long skipped_products = 0; // H q_K with q_K = 0: alpaqa would call the function (n_hess_psi_prod_calls)
        Iterate it0(n, m), it1(n, m), it2(n, m);
        Iterate *curr = &it0, *prox = &it1, *cand = &it2;
        std::vector<double> q(n), work(n), work2(n), rJ(n), wz(n), wr(n), wd(n), wBd(n), wev(n),
            vfull(n);
        std::vector<unsigned char> mask(n);

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
        auto backtrack_qub = [&](Iterate &i) {
            while (i.L < params.L_max && qub_violated(i)) {
                i.gamma = i.gamma / 2.0;
                i.L     = i.L * 2.0;
                prox_grad_step(i);
                eval_psi_xh(i);
                ++s.qub_backtracks;
            }
        };
        auto finish = [&](int status, double eps_k, double radius) {
            if (status == Converged || status == Interrupted || always_overwrite_results) {
                for (int j = 0; j < m; ++j) { // err_z = Sigma^{-1} (yhat - y), then y <- yhat
                    err_z[j] = (curr->yh[j] - y[j]) / Sigma[j];
                    y[j]     = curr->yh[j];
                }
                std::copy(curr->xh.begin(), curr->xh.end(), x);
            }
            s.status       = status;
            s.eps          = eps_k;
            s.final_gamma  = curr->gamma;
            s.final_psi    = curr->psi_xh;
            s.final_fbe    = curr->fbe();
            s.final_radius = radius;
            const EvalCounters &c = problem.cnt;
            s.n_psi            = c.psi - c0.psi;
            s.n_grad_psi       = c.grad_psi - c0.grad_psi;
            s.n_psi_grad_psi   = c.psi_grad_psi - c0.psi_grad_psi;
            s.n_grad_L         = c.grad_L - c0.grad_L;
            s.n_hess_psi_prod  = c.hess_psi_prod - c0.hess_psi_prod;
            s.n_hess_psi_prod_calls = s.n_hess_psi_prod + skipped_products;
            s.n_prox_grad_step = c.prox_grad_step - c0.prox_grad_step;
            s.elapsed_s        = std::chrono::duration<double>(clk::now() - t0).count();
            return s;
        };

        // ---- initial iterate and Lipschitz estimate (finite difference of the gradient) ----
        std::copy(x, x + n, curr->x.begin());
        if (params.L_0 <= 0) {
            curr->psi_x = problem.eval_psi_grad_psi(curr->x.data(), curr->grad.data());
            // step h along the gradient, |h_i| >= delta
            for (int i = 0; i < n; ++i) {
                double g = curr->grad[i];
                double h = g > 0 ? std::fmax(g * params.lip_eps, params.lip_delta)
                                 : std::fmin(g * params.lip_eps, -params.lip_delta);
                work[i]     = h;
                curr->xh[i] = (params.variant & 1) ? curr->x[i] + h : curr->x[i] - h;
            }
            double norm_h = std::sqrt(dot(work.data(), work.data(), n));
            problem.eval_grad_psi(curr->xh.data(), cand->grad.data());
            for (int i = 0; i < n; ++i)
                work[i] = cand->grad[i] - curr->grad[i];
            double L = std::sqrt(dot(work.data(), work.data(), n)) / norm_h;
            curr->L  = std::fmin(std::fmax(L, params.L_min), params.L_max);
        } else {
            curr->L     = params.L_0;
            curr->psi_x = problem.eval_psi_grad_psi(curr->x.data(), curr->grad.data());
        }
        if (!std::isfinite(curr->L)) { // nothing usable to return: leave x untouched
            s.status = NotFinite;
            return s;
        }
        curr->gamma = params.Lgamma_factor / curr->L;
        prox_grad_step(*curr);
        eval_psi_xh(*curr);
        backtrack_qub(*curr);

        int k = 0, no_progress = 0;
        double radius = params.initial_radius;
        if (!std::isfinite(radius) || radius == 0)
            radius = ((params.variant & 2) ? 1.0 : 0.1) * std::sqrt(dot(curr->grad.data(), curr->grad.data(), n));
        radius = std::fmax(radius, params.min_radius);

        while (true) {
            // ---- stop criterion ----
            double eps_k;
            switch (params.stop_crit) {
            case ProjGradUnitNorm: // || Pi_C(x - grad psi(x)) - x ||_inf   (results.hpp:268-271)
                problem.eval_prox_grad_step(1.0, curr->x.data(), curr->grad.data(), work.data(),
                                            work2.data());
                eps_k = norm_inf(work2.data(), n);
                break;
            case FPRNorm: // || p ||_inf / gamma
                eps_k = norm_inf(curr->p.data(), n) / curr->gamma;
                break;
            default: eps_k = NaN; break; // only the two criteria the reference selects
            }
            double elapsed   = std::chrono::duration<double>(clk::now() - t0).count();
            bool converged   = eps_k <= tolerance;
            bool out_of_time = elapsed > std::fmin(params.max_time_s, max_time_s);
            bool out_of_iter = k == params.max_iter;
            bool not_finite  = !std::isfinite(eps_k);
            bool stalled     = no_progress > params.max_no_progress;
            int status = converged ? Converged : out_of_time ? MaxTime : out_of_iter ? MaxIter
                         : not_finite ? NotFinite : stalled ? NoProgress : Busy;
            if (status != Busy) {
                s.iterations = k;
                return finish(status, eps_k, radius);
            }

            // ---- forward-backward step: the prox iterate xbar = xhat_k ----
            prox->x     = curr->xh;
            prox->psi_x = curr->psi_xh;
            problem.eval_grad_L(prox->x.data(), prox->grad.data());
            prox->gamma = curr->gamma;
            prox->L     = curr->L;
            prox_grad_step(*prox);
            // variant 16 (read literally): the prox iterate gets its psi(xhat) and its
            // quadratic-upper-bound backtracking BEFORE the direction, so that J, r_J and phi_gamma(xbar) see a step
            // size halved here; the specification (variant 0) defers both until the candidate has been rejected - the
            // step size is then the one the direction was built with, and an accepted candidate never pays for
            // psi(xhat(xbar)).  The two differ only where the bound test fails at xbar (s.qub_at_prox counts it).
            if (params.variant & 16) {
                const int before = s.qub_backtracks;
                eval_psi_xh(*prox);
                backtrack_qub(*prox);
                s.qub_at_prox += s.qub_backtracks - before;
            }

            // ---- Newton trust-region direction at xbar ----
            bool accept = false;
            double rho = NaN, q_model = NaN;
            int nJ = 0, cg_its = 0;
            if (!params.disable_acceleration) {
                nJ = problem.eval_inactive_indices_res_lna(prox->gamma, prox->x.data(),
                                                           prox->grad.data(), mask.data());
                double inv_gamma = -1.0 / prox->gamma;
                bool any_K       = nJ < n;
                for (int i = 0; i < n; ++i) {
                    rJ[i] = mask[i] ? inv_gamma * prox->p[i] : 0.0;
                    q[i]  = mask[i] ? 0.0 : ((params.variant & 4) ? 0.0 : prox->p[i]);
                }
                // The product with q_K is skipped when q_K = 0 (active bounds the prox step does not leave:
                // nine iterations in ten of the hanging-chain benchmark): its result is exactly zero.
                bool qK_zero = true;
                for (int i = 0; i < n; ++i)
                    qK_zero = qK_zero && q[i] == 0.0;
                if (dir_params.hessian_vec_factor != 0 && any_K && qK_zero)
                    ++skipped_products;
                if (dir_params.hessian_vec_factor != 0 && any_K && !qK_zero) {
                    problem.eval_hess_psi_prod(prox->x.data(), q.data(), work.data());
                    for (int i = 0; i < n; ++i)
                        if (mask[i])
                            rJ[i] = rJ[i] + dir_params.hessian_vec_factor * work[i];
                }
                auto hess_prod = [&](const double *p, double *Bp) {
                    problem.eval_hess_psi_prod(prox->x.data(), p, work.data());
                    for (int i = 0; i < n; ++i)
                        Bp[i] = mask[i] ? work[i] : 0.0;
                };
                q_model = steihaug_solve(accel_params, n, nJ, rJ.data(), hess_prod, radius,
                                         vfull.data(), wz, wr, wd, wBd, wev, cg_its);
                s.cg_iterations += cg_its;
                for (int i = 0; i < n; ++i)
                    q[i] = mask[i] ? vfull[i] : ((params.variant & 4) ? 0.0 : prox->p[i]);
                if ((params.variant & 8) && any_K) { // model value of the whole step q = (q_J, p_K), gradient = grad psi(xbar)
                    problem.eval_hess_psi_prod(prox->x.data(), q.data(), work.data());
                    double lin = 0, quad = 0;
                    for (int i = 0; i < n; ++i) {
                        lin += prox->grad[i] * q[i];
                        quad += q[i] * work[i];
                    }
                    q_model = lin + 0.5 * quad;
                }
                bool q_finite = true;
                for (int i = 0; i < n; ++i)
                    q_finite = q_finite && std::isfinite(q[i]);
                if (q_finite && q_model < 0) {
                    // candidate x+ = xbar + q
                    for (int i = 0; i < n; ++i)
                        cand->x[i] = prox->x[i] + q[i];
                    cand->psi_x = problem.eval_psi_grad_psi(cand->x.data(), cand->grad.data());
                    cand->gamma = prox->gamma;
                    cand->L     = prox->L;
                    prox_grad_step(*cand);
                    double phi_k = prox->fbe(), phi_n = cand->fbe();
                    double tr_eps = (1.0 + std::fabs(phi_k)) * params.TR_tolerance_factor;
                    rho    = ((phi_k - phi_n) + tr_eps) / (-q_model);
                    accept = rho >= params.ratio_threshold_acceptable;
                    double qn = std::sqrt(dot(q.data(), q.data(), n));
                    double nr = rho >= params.ratio_threshold_good
                                    ? std::fmax(params.radius_factor_good * qn, radius)
                                : rho >= params.ratio_threshold_acceptable
                                    ? radius * params.radius_factor_acceptable
                                    : params.radius_factor_rejected * qn;
                    if (std::isnan(rho)) // failed candidate evaluation: treat as rejected
                        nr = params.radius_factor_rejected * qn;
                    radius = std::fmax(nr, params.min_radius);
                }
            }
            if (trace)
                trace->push_back({eps_k, curr->psi_x, std::sqrt(curr->pTp), curr->gamma, radius, rho,
                                  q_model, curr->fbe(), nJ, cg_its, accept ? 1 : 0});

            // ---- next iterate ----
            Iterate *old = curr;
            if (accept) {
                eval_psi_xh(*cand);
                backtrack_qub(*cand);
                std::swap(curr, cand);
                ++s.accepted;
            } else {
                if (!(params.variant & 16)) {
                    const int before = s.qub_backtracks;
                    eval_psi_xh(*prox);
                    backtrack_qub(*prox);
                    s.qub_at_prox += s.qub_backtracks - before;
                }
                std::swap(curr, prox);
                ++s.rejected;
            }
            bool same = true;
            for (int i = 0; i < n; ++i)
                same = same && curr->x[i] == old->x[i];
            no_progress = same ? no_progress + 1 : 0;
            ++k;
        }
    }
};

/// ALMParams: alpaqa defaults; mpc-driver.cpp:109-111 and dodo.py:12-22 override.
struct ALMParams {
    int max_iter = 200;              // mpc-driver.cpp:109, dodo.py:13 sets 300
    double tolerance = 1e-8, dual_tolerance = 1e-8; // mpc-driver.cpp:110-111
    double max_time_s = 300;         // dodo.py:14 sets 30 s
    double initial_penalty = 1, penalty_update_factor = 10, initial_tolerance = 1,
           tolerance_update_factor = 1e-1; // dodo.py:15-18
    double rel_penalty_increase_threshold = 0.1, max_multiplier = 1e9, max_penalty = 1e9;
    int single_penalty_factor = 0;
};

/// ALM outer loop (alpaqa::ALMSolver::operator(), named at mpc-driver.cpp:107,114,217).
/// m == 0: a single inner solve at alm.tolerance with always_overwrite_results = true.
/// m > 0: Sigma_0 = initial_penalty, eps_0 = initial_tolerance; per outer iteration: project the
/// multipliers, inner solve (y <- yhat, err = Sigma^{-1}(yhat - y)), stop when the inner solver
/// converged with eps <= tolerance and ||err||_inf <= dual_tolerance; otherwise raise the
/// penalty of the components whose violation did not shrink by the threshold factor,
/// Sigma_i <- min(max_penalty, max(Delta |e_i| / ||e||_inf, 1) Sigma_i), and tighten eps.
template <class Inner, class Problem>
Stats alm_solve(const ALMParams &alm, const Inner &inner, Problem &problem, double *x, double *y = nullptr) {
    using clk = std::chrono::steady_clock;
    auto t0   = clk::now();
    const int m = problem.m();
    if (m == 0)
        return inner(problem, alm.tolerance, alm.max_time_s, /*always_overwrite_results=*/true, x);
    std::vector<double> Sigma(m, alm.initial_penalty), e1(m, NaN), e2(m, NaN);
    double norm_e1 = NaN, eps = alm.initial_tolerance;
    const double Delta = alm.penalty_update_factor, rho = alm.tolerance_update_factor;
    Stats total;
    int num_successful = 0;
    for (int i = 0; i < alm.max_iter; ++i) {
        for (int j = 0; j < m; ++j) { // eval_proj_multipliers(y, M)
            double ylb = problem.d_lb(j) == -inf ? 0.0 : -alm.max_multiplier;
            double yub = problem.d_ub(j) == +inf ? 0.0 : +alm.max_multiplier;
            y[j]       = std::fmin(std::fmax(y[j], ylb), yub);
        }
        double elapsed = std::chrono::duration<double>(clk::now() - t0).count();
        Stats ps = inner(problem, eps, std::fmax(alm.max_time_s - elapsed, 0.0), true, x, y, Sigma.data(),
                         e2.data());
        bool inner_converged = ps.status == Converged;
        total.iterations += ps.iterations;
        total.cg_iterations += ps.cg_iterations;
        total.accepted += ps.accepted;
        total.rejected += ps.rejected;
        total.qub_backtracks += ps.qub_backtracks;
        total.qub_at_prox += ps.qub_at_prox;
        total.n_psi += ps.n_psi; total.n_grad_psi += ps.n_grad_psi; total.n_psi_grad_psi += ps.n_psi_grad_psi;
        total.n_grad_L += ps.n_grad_L; total.n_hess_psi_prod += ps.n_hess_psi_prod;
        total.n_hess_psi_prod_calls += ps.n_hess_psi_prod_calls;
        total.n_prox_grad_step += ps.n_prox_grad_step;
        total.linesearch_failures += ps.linesearch_failures; total.linesearch_backtracks += ps.linesearch_backtracks;
        total.lbfgs_failures += ps.lbfgs_failures; total.lbfgs_rejected += ps.lbfgs_rejected;
        total.tau_1_accepted += ps.tau_1_accepted; total.count_tau += ps.count_tau; total.sum_tau += ps.sum_tau;
        total.inner_convergence_failures += inner_converged ? 0 : 1;
        double norm_e2 = norm_inf(e2.data(), m);
        elapsed        = std::chrono::duration<double>(clk::now() - t0).count();
        bool alm_converged = ps.eps <= alm.tolerance && inner_converged && norm_e2 <= alm.dual_tolerance;
        bool out_of_time = elapsed > alm.max_time_s, out_of_iter = i + 1 == alm.max_iter;
        if (alm_converged || out_of_iter || out_of_time) {
            total.status = alm_converged ? Converged : out_of_time ? MaxTime : MaxIter;
            total.eps = ps.eps; total.delta = norm_e2; total.outer_iterations = i + 1;
            total.final_gamma = ps.final_gamma; total.final_psi = ps.final_psi; total.final_fbe = ps.final_fbe;
            total.final_radius = ps.final_radius; total.elapsed_s = elapsed;
            double sq = 0;
            for (int j = 0; j < m; ++j)
                sq += Sigma[j] * Sigma[j];
            total.penalty_norm = std::sqrt(sq / m);
            return total;
        }
        // update_penalty_weights: e2 = current error, e1 = previous
        if (norm_e2 > alm.dual_tolerance) {
            bool first = num_successful == 0;
            if (alm.single_penalty_factor) {
                if (first || norm_e2 > alm.rel_penalty_increase_threshold * norm_e1) {
                    double ns = std::fmin(alm.max_penalty, Delta * Sigma[0]);
                    std::fill(Sigma.begin(), Sigma.end(), ns);
                }
            } else {
                for (int j = 0; j < m; ++j)
                    if (first || std::fabs(e2[j]) > alm.rel_penalty_increase_threshold * std::fabs(e1[j]))
                        Sigma[j] = std::fmin(alm.max_penalty,
                                             std::fmax((Delta * std::fabs(e2[j])) / norm_e2, 1.0) * Sigma[j]);
            }
        }
        eps = std::fmax(rho * eps, alm.tolerance);
        ++num_successful;
        std::swap(e1, e2);
        norm_e1 = norm_e2;
    }
    total.status = MaxIter;
    return total;
}

} // namespace oracle
