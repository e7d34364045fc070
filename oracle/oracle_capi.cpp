// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/README.md).
//
// C entry points (ctypes-friendly) over the scalar restatement in
// chain_model.hpp / pantr_oracle.hpp, plus the closed-loop MPC experiment of
// drivers/mpc-driver.cpp:123-210 and a threaded batch driver used as the CPU
// baseline ("port") by bench.py.
#include "panoc_oracle.hpp"
#include "quad_models.hpp"
#include <atomic>
#include <thread>

using namespace oracle;
using Prob = SingleShooting<Chain>;

extern "C" {

/// Flat parameter block (doubles first, then ints) mirrored by tests/oracle_binding.py.
struct OracleParams {
    // solver.* (PANTRParams)
    double L_0, lip_eps, lip_delta, Lgamma_factor, max_time_s, L_min, L_max,
        quadratic_upperbound_tolerance_factor, TR_tolerance_factor, ratio_threshold_acceptable,
        ratio_threshold_good, radius_factor_rejected, radius_factor_acceptable, radius_factor_good,
        initial_radius, min_radius;
    // dir.*, accel.*
    double hessian_vec_factor, tol_scale, tol_scale_root, tol_max, max_iter_factor;
    // alm.*
    double alm_tolerance, alm_max_time_s;
    double alm_dual_tolerance, alm_initial_penalty, alm_penalty_update_factor, alm_initial_tolerance,
        alm_tolerance_update_factor;
    // solver.* of PANOCParams that PANTRParams lacks, accel.* of LBFGSParams (method = panoc, strucpanoc)
    double min_linesearch_coefficient, linesearch_strictness_factor, linesearch_tolerance_factor,
        lbfgs_min_div_fac, lbfgs_min_abs_s, cbfgs_alpha, cbfgs_eps;
    int32_t max_iter, stop_crit, max_no_progress, disable_acceleration;
    int32_t alm_max_iter;
    int32_t method; ///< 0 = pantr, 1 = panoc, 2 = strucpanoc (mpc-driver.cpp:364-384)
    int32_t force_linesearch, lbfgs_memory, lbfgs_force_pos_def, variant; ///< variant: PANTRParams::variant (0 = the specification)
};
enum Method : int32_t { MethodPANTR = 0, MethodPANOC = 1, MethodStrucPANOC = 2 };

struct OracleStats {
    double eps, final_gamma, final_psi, final_fbe, final_radius, elapsed_s;
    int32_t status, iterations, cg_iterations, accepted, rejected, qub_backtracks;
    int32_t n_psi, n_grad_psi, n_psi_grad_psi, n_grad_L, n_hess_psi_prod, n_prox_grad_step;
    int32_t outer_iterations, inner_convergence_failures;
    int32_t n_hess_psi_prod_calls, qub_at_prox; ///< qub_at_prox: quadratic-upper-bound backtracking steps taken at a prox iterate
    double delta, penalty_norm;
    int32_t linesearch_failures, linesearch_backtracks, lbfgs_failures, lbfgs_rejected, tau_1_accepted,
        count_tau;
    double sum_tau;
};

/// Defaults = alpaqa defaults + mpc-driver.cpp:51-53,109-111.
void oracle_default_params(OracleParams *p) {
    PANTRParams a;
    NewtonTRDirectionParams d;
    SteihaugCGParams c;
    ALMParams alm;
    p->L_0 = a.L_0; p->lip_eps = a.lip_eps; p->lip_delta = a.lip_delta;
    p->Lgamma_factor = a.Lgamma_factor; p->max_time_s = a.max_time_s;
    p->L_min = a.L_min; p->L_max = a.L_max;
    p->quadratic_upperbound_tolerance_factor = a.quadratic_upperbound_tolerance_factor;
    p->TR_tolerance_factor = a.TR_tolerance_factor;
    p->ratio_threshold_acceptable = a.ratio_threshold_acceptable;
    p->ratio_threshold_good = a.ratio_threshold_good;
    p->radius_factor_rejected = a.radius_factor_rejected;
    p->radius_factor_acceptable = a.radius_factor_acceptable;
    p->radius_factor_good = a.radius_factor_good;
    p->initial_radius = a.initial_radius; p->min_radius = a.min_radius;
    p->hessian_vec_factor = d.hessian_vec_factor;
    p->tol_scale = c.tol_scale; p->tol_scale_root = c.tol_scale_root; p->tol_max = c.tol_max;
    p->max_iter_factor = c.max_iter_factor;
    p->alm_tolerance = alm.tolerance; p->alm_max_time_s = alm.max_time_s;
    p->alm_dual_tolerance = alm.dual_tolerance; p->alm_initial_penalty = alm.initial_penalty;
    p->alm_penalty_update_factor = alm.penalty_update_factor;
    p->alm_initial_tolerance = alm.initial_tolerance;
    p->alm_tolerance_update_factor = alm.tolerance_update_factor;
    p->alm_max_iter = alm.max_iter; p->variant = 0;
    PANOCParams pn;
    LBFGSParams lb;
    p->min_linesearch_coefficient = pn.min_linesearch_coefficient;
    p->linesearch_strictness_factor = pn.linesearch_strictness_factor;
    p->linesearch_tolerance_factor = pn.linesearch_tolerance_factor;
    p->lbfgs_min_div_fac = lb.min_div_fac; p->lbfgs_min_abs_s = lb.min_abs_s;
    p->cbfgs_alpha = lb.cbfgs_alpha; p->cbfgs_eps = lb.cbfgs_eps;
    p->method = MethodPANTR; p->force_linesearch = pn.force_linesearch; p->lbfgs_memory = lb.memory;
    p->lbfgs_force_pos_def = lb.force_pos_def;
    p->max_iter = 50000;          // mpc-driver.cpp:51
    p->stop_crit = FPRNorm;       // mpc-driver.cpp:53
    p->max_no_progress = a.max_no_progress;
    p->disable_acceleration = 0;
}

/// The PANTR option set of new-benchmarks-paper/dodo.py:12-33 on top of the defaults.
void oracle_dodo_params(OracleParams *p) {
    oracle_default_params(p);
    p->alm_max_time_s = 30;               // alm.max_time=30s
    p->alm_max_iter = 300;                // alm.max_iter=300
    p->alm_initial_penalty = 1e4;         // alm.initial_penalty=1e4
    p->alm_penalty_update_factor = 5;     // alm.penalty_update_factor=5
    p->alm_initial_tolerance = 1e2;       // alm.initial_tolerance=1e2
    p->alm_tolerance_update_factor = 0.1; // alm.tolerance_update_factor=0.1
    p->alm_dual_tolerance = 1e-8;         // alm.dual_tolerance
    p->stop_crit = ProjGradUnitNorm;      // solver.stop_crit=ProjGradUnitNorm
    p->alm_tolerance = 1e-8;              // alm.tolerance
    p->radius_factor_rejected = 0.35;
    p->radius_factor_acceptable = 0.99;
    p->radius_factor_good = 10;
    p->ratio_threshold_acceptable = 0.3;
    p->ratio_threshold_good = 0.5;
    p->hessian_vec_factor = 1;
    p->tol_scale = 1e-2;
}
/// dodo.py:34-40: the comparators' option sets (OPT_ALM + accel.memory=50 [+ dir.hessian_vec=0]).
void oracle_dodo_params_method(OracleParams *p, int method) {
    oracle_dodo_params(p);
    p->method = method;
    if (method != MethodPANTR)
        p->lbfgs_memory = 50;
}

static ALMParams make_alm(const OracleParams &p) {
    ALMParams a;
    a.max_iter = p.alm_max_iter; a.tolerance = p.alm_tolerance; a.dual_tolerance = p.alm_dual_tolerance;
    a.max_time_s = p.alm_max_time_s; a.initial_penalty = p.alm_initial_penalty;
    a.penalty_update_factor = p.alm_penalty_update_factor; a.initial_tolerance = p.alm_initial_tolerance;
    a.tolerance_update_factor = p.alm_tolerance_update_factor;
    return a;
}

extern "C++" {
template <class P = Prob>
static PANTRSolver<P> make_solver(const OracleParams &p) {
    PANTRSolver<P> s;
    auto &a = s.params;
    a.L_0 = p.L_0; a.lip_eps = p.lip_eps; a.lip_delta = p.lip_delta;
    a.Lgamma_factor = p.Lgamma_factor; a.max_time_s = p.max_time_s;
    a.L_min = p.L_min; a.L_max = p.L_max;
    a.quadratic_upperbound_tolerance_factor = p.quadratic_upperbound_tolerance_factor;
    a.TR_tolerance_factor = p.TR_tolerance_factor;
    a.ratio_threshold_acceptable = p.ratio_threshold_acceptable;
    a.ratio_threshold_good = p.ratio_threshold_good;
    a.radius_factor_rejected = p.radius_factor_rejected;
    a.radius_factor_acceptable = p.radius_factor_acceptable;
    a.radius_factor_good = p.radius_factor_good;
    a.initial_radius = p.initial_radius; a.min_radius = p.min_radius;
    a.max_iter = p.max_iter; a.stop_crit = p.stop_crit; a.max_no_progress = p.max_no_progress;
    a.disable_acceleration = p.disable_acceleration;
    a.variant = p.variant;
    s.dir_params.hessian_vec_factor = p.hessian_vec_factor;
    s.accel_params.tol_scale = p.tol_scale; s.accel_params.tol_scale_root = p.tol_scale_root;
    s.accel_params.tol_max = p.tol_max; s.accel_params.max_iter_factor = p.max_iter_factor;
    return s;
}
template <class P, class Dir>
static PANOCSolver<P, Dir> make_panoc(const OracleParams &p) {
    PANOCSolver<P, Dir> s;
    auto &a = s.params;
    a.L_0 = p.L_0; a.lip_eps = p.lip_eps; a.lip_delta = p.lip_delta;
    a.Lgamma_factor = p.Lgamma_factor; a.max_time_s = p.max_time_s;
    a.L_min = p.L_min; a.L_max = p.L_max;
    a.quadratic_upperbound_tolerance_factor = p.quadratic_upperbound_tolerance_factor;
    a.max_iter = p.max_iter; a.stop_crit = p.stop_crit; a.max_no_progress = p.max_no_progress;
    a.min_linesearch_coefficient = p.min_linesearch_coefficient;
    a.linesearch_strictness_factor = p.linesearch_strictness_factor;
    a.linesearch_tolerance_factor = p.linesearch_tolerance_factor;
    a.force_linesearch = p.force_linesearch;
    auto &l = s.direction.lbfgs.params;
    l.memory = p.lbfgs_memory; l.min_div_fac = p.lbfgs_min_div_fac; l.min_abs_s = p.lbfgs_min_abs_s;
    l.cbfgs_alpha = p.cbfgs_alpha; l.cbfgs_eps = p.cbfgs_eps; l.force_pos_def = p.lbfgs_force_pos_def;
    return s;
}
/// f(inner solver) with the solver `method` names (mpc-driver.cpp:364-384).
template <class P, class F>
static auto with_solver(const OracleParams &p, F &&f) {
    if (p.method == MethodPANOC)
        return f(make_panoc<P, LBFGSDirection>(p));
    if (p.method == MethodStrucPANOC)
        return f(make_panoc<P, StructuredLBFGSDirection>(p));
    return f(make_solver<P>(p));
}
} // extern "C++"

static void export_stats(const Stats &s, OracleStats *o) {
    o->eps = s.eps; o->final_gamma = s.final_gamma; o->final_psi = s.final_psi;
    o->final_fbe = s.final_fbe; o->final_radius = s.final_radius; o->elapsed_s = s.elapsed_s;
    o->status = s.status; o->iterations = s.iterations; o->cg_iterations = s.cg_iterations;
    o->accepted = s.accepted; o->rejected = s.rejected; o->qub_backtracks = s.qub_backtracks;
    o->n_psi = (int32_t)s.n_psi; o->n_grad_psi = (int32_t)s.n_grad_psi;
    o->n_psi_grad_psi = (int32_t)s.n_psi_grad_psi; o->n_grad_L = (int32_t)s.n_grad_L;
    o->n_hess_psi_prod = (int32_t)s.n_hess_psi_prod;
    o->n_prox_grad_step = (int32_t)s.n_prox_grad_step;
    o->outer_iterations = s.outer_iterations; o->inner_convergence_failures = s.inner_convergence_failures;
    o->n_hess_psi_prod_calls = (int32_t)s.n_hess_psi_prod_calls; o->qub_at_prox = s.qub_at_prox;
    o->delta = s.delta; o->penalty_norm = s.penalty_norm;
    o->linesearch_failures = s.linesearch_failures; o->linesearch_backtracks = s.linesearch_backtracks;
    o->lbfgs_failures = s.lbfgs_failures; o->lbfgs_rejected = s.lbfgs_rejected;
    o->tau_1_accepted = s.tau_1_accepted; o->count_tau = s.count_tau; o->sum_tau = s.sum_tau;
}

// ---- problem functions --------------------------------------------------------------------
int oracle_chain_nx(void) { return Chain::nx; }
int oracle_chain_nu(void) { return Chain::nu; }
void oracle_chain_initial_state(double *x0) { Chain::initial_state(x0); }
void oracle_chain_box(double *lb, double *ub) {
    for (int c = 0; c < Chain::nu; ++c) {
        lb[c] = Chain::u_lb(c);
        ub[c] = Chain::u_ub(c);
    }
}
void oracle_chain_fc(const double *s, const double *u, double *ds) {
    Chain::Springs<double> sp;
    Chain::fc(s, u, ds, sp);
}
void oracle_chain_simulate(const double *x, const double *u, double *xn) {
    Prob::simulate_dynamics(x, u, xn);
}
double oracle_chain_psi(int N, const double *x0, const double *u) {
    Prob p(N);
    std::copy(x0, x0 + Chain::nx, p.x0.begin());
    return p.eval_psi(u);
}
double oracle_chain_psi_grad(int N, const double *x0, const double *u, double *grad) {
    Prob p(N);
    std::copy(x0, x0 + Chain::nx, p.x0.begin());
    return p.eval_psi_grad_psi(u, grad);
}
void oracle_chain_hess_prod(int N, const double *x0, const double *u, const double *v, double *Hv) {
    Prob p(N);
    std::copy(x0, x0 + Chain::nx, p.x0.begin());
    p.eval_hess_psi_prod(u, v, Hv);
}
void oracle_chain_hess_prod_dual(int N, const double *x0, const double *u, const double *v, double *Hv) {
    Prob p(N);
    std::copy(x0, x0 + Chain::nx, p.x0.begin());
    p.eval_hess_psi_prod_dual(u, v, Hv);
}
void oracle_chain_prox_grad_step(int N, double gamma, const double *x, const double *grad,
                                 double *xh, double *p, double *pTp, double *gradTp) {
    Prob pr(N);
    pr.eval_prox_grad_step(gamma, x, grad, xh, p);
    *pTp    = dot(p, p, pr.n());
    *gradTp = dot(p, grad, pr.n());
}
double oracle_dot(const double *a, const double *b, int n) { return dot(a, b, n); }

// ---- one solve --------------------------------------------------------------------------
int oracle_chain_solve(int N, const OracleParams *prm, const double *x0, double *u,
                       OracleStats *st, double *trace /*cap x 11*/, int trace_cap, int *trace_len) {
    Prob p(N);
    std::copy(x0, x0 + Chain::nx, p.x0.begin());
    if (prm->method != MethodPANTR) { // trace columns: eps psi |p| gamma tau fbe lbfgs-pairs
        std::vector<PanocTraceRow> ptr;
        Stats s = with_solver<Prob>(*prm, [&](auto solver) {
            if constexpr (requires { solver.direction; })
                if (trace)
                    solver.trace = &ptr;
            return solver(p, prm->alm_tolerance, prm->alm_max_time_s, true, u);
        });
        export_stats(s, st);
        if (trace) {
            int len = std::min<int>((int)ptr.size(), trace_cap);
            for (int i = 0; i < len; ++i) {
                const PanocTraceRow &r = ptr[i];
                double *o = trace + 11 * i;
                std::fill(o, o + 11, NaN);
                o[0] = r.eps; o[1] = r.psi; o[2] = r.pnorm; o[3] = r.gamma; o[4] = r.tau; o[5] = r.fbe;
                o[6] = r.lbfgs_idx;
            }
            *trace_len = (int)ptr.size();
        }
        return s.status;
    }
    auto solver = make_solver(*prm);
    std::vector<TraceRow> tr;
    if (trace)
        solver.trace = &tr;
    Stats s = solver(p, prm->alm_tolerance, prm->alm_max_time_s, true, u);
    export_stats(s, st);
    if (trace) {
        int len = std::min<int>((int)tr.size(), trace_cap);
        for (int i = 0; i < len; ++i) {
            const TraceRow &r = tr[i];
            double *o = trace + 11 * i;
            o[0] = r.eps; o[1] = r.psi; o[2] = r.pnorm; o[3] = r.gamma; o[4] = r.radius;
            o[5] = r.rho; o[6] = r.q_model; o[7] = r.fbe; o[8] = r.nJ; o[9] = r.cg_its;
            o[10] = r.accepted;
        }
        *trace_len = (int)tr.size();
    }
    return s.status;
}

/// B independent solves over `nthreads` host threads (dynamic work queue).
void oracle_chain_solve_batch(int N, const OracleParams *prm, int B, const double *x0s,
                              double *us, OracleStats *stats, int nthreads) {
    const int n = Chain::nu * N;
    std::atomic<int> next{0};
    auto worker = [&] {
        with_solver<Prob>(*prm, [&](auto solver) {
            for (int b; (b = next.fetch_add(1)) < B;) {
                Prob p(N);
                std::copy(x0s + b * Chain::nx, x0s + (b + 1) * Chain::nx, p.x0.begin());
                Stats s = solver(p, prm->alm_tolerance, prm->alm_max_time_s, true, us + (size_t)b * n);
                export_stats(s, &stats[b]);
            }
            return 0;
        });
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t)
        th.emplace_back(worker);
    worker();
    for (auto &t : th)
        t.join();
}

/// Closed-loop MPC experiment, drivers/mpc-driver.cpp:123-210 (single shooting, m = 0).
/// Outputs: states (num_sim+1) x nx, inputs num_sim x n (row per step: the whole optimal u),
/// guesses num_sim x n (the warm/cold start each timed solve began from), stats num_sim.
void oracle_chain_mpc(int N, const OracleParams *prm, int num_sim, int warm, int num_dist,
                      const double *u_dist, double *states, double *inputs, double *guesses,
                      OracleStats *stats) {
    const int nx = Chain::nx, nu = Chain::nu, n = nu * N;
    Prob p(N);
    with_solver<Prob>(*prm, [&](auto solver) {
    std::vector<double> x0(nx), xn(nx), u(n, 0.0), u_initial(n, 0.0); // hanging_chain.cpp:135
    Chain::initial_state(x0.data());
    for (int i = 0; i < num_dist; ++i) { // mpc-driver.cpp:153-154
        Prob::simulate_dynamics(x0.data(), u_dist, xn.data());
        x0 = xn;
    }
    p.x0 = x0;
    solver(p, prm->alm_tolerance, prm->alm_max_time_s, true, u.data()); // untimed, :158
    Prob::simulate_dynamics(x0.data(), u.data(), xn.data());           // :160-161
    x0 = xn;
    for (int i = 0; i < num_sim; ++i) {
        if (warm) // shift by nu, last block kept (:170)
            std::copy(u.begin() + nu, u.end(), u.begin());
        else
            u = u_initial;
        std::copy(u.begin(), u.end(), guesses + (size_t)i * n);
        p.x0 = x0;
        Stats s = solver(p, prm->alm_tolerance, prm->alm_max_time_s, true, u.data());
        export_stats(s, &stats[i]);
        std::copy(x0.begin(), x0.end(), states + (size_t)i * nx);
        std::copy(u.begin(), u.end(), inputs + (size_t)i * n);
        Prob::simulate_dynamics(x0.data(), u.data(), xn.data()); // :205
        x0 = xn;
    }
    std::copy(x0.begin(), x0.end(), states + (size_t)num_sim * nx);
    return 0;
    });
}

// ---- quadcopter problems (general constraints, ALM) -------------------------------------------
extern "C++" {
template <class F>
static auto with_quad(int realistic, int N, F &&f) {
    if (realistic) {
        SingleShootingAL<RealisticQuadcopter> p(N);
        return f(p);
    }
    SingleShootingAL<Quadcopter> p(N);
    return f(p);
}
} // extern "C++"
void oracle_sincos(double x, double *s, double *c) { sincos_det(x, *s, *c); }
void oracle_quad_initial_state(double *x0) { Quadcopter::initial_state(x0); }
void oracle_quad_bounds(double *ulb, double *uub, double *dlb, double *dub) {
    for (int c = 0; c < 4; ++c) {
        ulb[c] = Quadcopter::u_lb(c); uub[c] = Quadcopter::u_ub(c);
        dlb[c] = Quadcopter::d_lb(c); dub[c] = Quadcopter::d_ub(c);
    }
}
void oracle_quad_simulate(int realistic, const double *x, const double *u, double *xn) {
    with_quad(realistic, 1, [&](auto &p) { p.simulate_dynamics(x, u, xn); return 0; });
}
void oracle_quad_fc(int realistic, const double *s, const double *u, double *ds) {
    if (realistic) RealisticQuadcopter::fc(s, u, ds); else Quadcopter::fc(s, u, ds);
}
double oracle_quad_f_g(int realistic, int N, const double *x0, const double *u, double *g) {
    return with_quad(realistic, N, [&](auto &p) {
        std::copy(x0, x0 + 9, p.x0.begin());
        p.eval_g(u, g);
        return p.eval_f(u);
    });
}
double oracle_quad_psi_grad(int realistic, int N, const double *x0, const double *y, const double *Sig,
                            const double *u, double *grad, double *yhat) {
    return with_quad(realistic, N, [&](auto &p) {
        std::copy(x0, x0 + 9, p.x0.begin());
        p.set_al(y, Sig);
        p.eval_psi(u, yhat);
        return p.eval_psi_grad_psi(u, grad);
    });
}
/// The two scan orders of the quadcopter-family specification (tests/test_oracle.py pins them).
void oracle_prefix_scan(const double *r, int n, double init, double *excl) {
    oracle::SingleShootingAL<oracle::Quadcopter>::prefix_scan(r, n, init, excl);
}
/// maps: n x 6 doubles (A00, A01, A10, A11, b0, b1); out: 2 (n + 1)
void oracle_affine_scan(const double *maps, int n, const double *y0, double *out) {
    using P = oracle::SingleShootingAL<oracle::Quadcopter>;
    std::vector<P::Affine2> m(n);
    for (int k = 0; k < n; ++k)
        m[k] = {{{maps[6 * k], maps[6 * k + 1]}, {maps[6 * k + 2], maps[6 * k + 3]}}, {maps[6 * k + 4], maps[6 * k + 5]}};
    P::affine_scan(m.data(), n, y0, out);
}
/// grad psi through the stage-by-stage adjoint sweep (the check of the affine sweep realistic_quadcopter uses)
void oracle_quad_grad_generic(int realistic, int N, const double *x0, const double *y, const double *Sig,
                              const double *u, double *grad) {
    with_quad(realistic, N, [&](auto &p) {
        std::copy(x0, x0 + 9, p.x0.begin());
        p.set_al(y, Sig);
        std::vector<double> X((N + 1) * 9), yh(4 * N + 4);
        p.forward(u, X.data(), yh.data());
        p.reverse_generic(u, X.data(), yh.data(), grad);
        return 0;
    });
}
void oracle_quad_hess_prod(int realistic, int N, const double *x0, const double *y, const double *Sig,
                           const double *u, const double *v, double *Hv) {
    with_quad(realistic, N, [&](auto &p) {
        std::copy(x0, x0 + 9, p.x0.begin());
        p.set_al(y, Sig);
        p.eval_hess_psi_prod(u, v, Hv);
        return 0;
    });
}
void oracle_quad_hess_prod_dual(int realistic, int N, const double *x0, const double *y, const double *Sig,
                                const double *u, const double *v, double *Hv) {
    with_quad(realistic, N, [&](auto &p) {
        std::copy(x0, x0 + 9, p.x0.begin());
        p.set_al(y, Sig);
        p.eval_hess_psi_prod_dual(u, v, Hv);
        return 0;
    });
}
/// ALM solve of one instance: u (n) and y (m) in/out.
int oracle_quad_solve(int realistic, int N, const OracleParams *prm, const double *x0, double *u, double *y,
                      OracleStats *st) {
    return with_quad(realistic, N, [&](auto &p) {
        using P = std::decay_t<decltype(p)>;
        std::copy(x0, x0 + 9, p.x0.begin());
        Stats s = with_solver<P>(*prm, [&](auto solver) { return alm_solve(make_alm(*prm), solver, p, u, y); });
        export_stats(s, st);
        return s.status;
    });
}
/// Closed loop of mpc-driver.cpp:123-210 with multipliers: warm start shifts u by nu and y by nc.
/// Closed loop of drivers/mpc-driver.cpp:123-210 for the quadcopter family.  Optional outputs (may
/// be null): the guesses every step was solved from (u: [num_sim][n], y: [num_sim][m]) and the
/// multipliers it returned ([num_sim][m]) - what a replay batch needs.
void oracle_quad_mpc_full(int realistic, int N, const OracleParams *prm, int num_sim, int warm, int num_dist,
                          const double *u_dist, double *states, double *inputs, OracleStats *stats,
                          double *guesses_u, double *guesses_y, double *multipliers) {
    with_quad(realistic, N, [&](auto &p) {
        using P = std::decay_t<decltype(p)>;
        const int nx = P::nx, nu = P::nu, nc = P::nc, n = p.n(), m = p.m();
        ALMParams alm = make_alm(*prm);
        return with_solver<P>(*prm, [&](auto solver) {
        std::vector<double> x0(nx), xn(nx), u(n), u_initial(n), y(m, 0.0);
        Quadcopter::initial_state(x0.data());
        for (int i = 0; i < n; ++i)
            u_initial[i] = Quadcopter::initial_guess(i % nu);
        u = u_initial;
        for (int i = 0; i < num_dist; ++i) {
            P::simulate_dynamics(x0.data(), u_dist, xn.data());
            x0 = xn;
        }
        p.x0 = x0;
        alm_solve(alm, solver, p, u.data(), y.data());
        P::simulate_dynamics(x0.data(), u.data(), xn.data());
        x0 = xn;
        for (int i = 0; i < num_sim; ++i) {
            if (warm) { // mpc-driver.cpp:168-175
                std::copy(u.begin() + nu, u.end(), u.begin());
                std::copy(y.begin() + nc, y.end(), y.begin());
            } else {
                u = u_initial;
                std::fill(y.begin(), y.end(), 0.0);
            }
            p.x0 = x0;
            if (guesses_u)
                std::copy(u.begin(), u.end(), guesses_u + (size_t)i * n);
            if (guesses_y)
                std::copy(y.begin(), y.end(), guesses_y + (size_t)i * m);
            Stats s = alm_solve(alm, solver, p, u.data(), y.data());
            if (multipliers)
                std::copy(y.begin(), y.end(), multipliers + (size_t)i * m);
            export_stats(s, &stats[i]);
            std::copy(x0.begin(), x0.end(), states + (size_t)i * nx);
            std::copy(u.begin(), u.end(), inputs + (size_t)i * n);
            P::simulate_dynamics(x0.data(), u.data(), xn.data());
            x0 = xn;
        }
        std::copy(x0.begin(), x0.end(), states + (size_t)num_sim * nx);
        return 0;
        });
    });
}

void oracle_quad_mpc(int realistic, int N, const OracleParams *prm, int num_sim, int warm, int num_dist,
                     const double *u_dist, double *states, double *inputs, OracleStats *stats) {
    oracle_quad_mpc_full(realistic, N, prm, num_sim, warm, num_dist, u_dist, states, inputs, stats, nullptr, nullptr,
                         nullptr);
}

/// L-BFGS two-loop recursion on given pairs (cross-check against the dense BFGS recursion):
/// pairs[i] = (S[i], Y[i]), oldest first; mask == nullptr: all indices.  Returns 0 when the
/// application failed (no usable pair).
int oracle_lbfgs_apply(int n, int memory, int count, const double *S, const double *Y, double *q,
                       const unsigned char *mask) {
    LBFGS l;
    l.params.memory = memory;
    l.resize(n);
    std::vector<double> zero(n, 0.0), s(n), y(n);
    for (int i = 0; i < count; ++i) // update(x, xn, p, pn, positive, forced): s = xn - x, y = pn - p
        l.update(zero.data(), S + (size_t)i * n, zero.data(), Y + (size_t)i * n, true, true);
    if (!mask)
        return l.apply(q);
    int nJ = 0;
    for (int i = 0; i < n; ++i)
        nJ += mask[i] != 0;
    return l.apply_masked(q, mask, nJ);
}

/// Steihaug CG on a dense symmetric matrix (cross-check against SciPy).
double oracle_steihaug_dense(int n, const double *H /*n x n row-major*/, const double *g,
                             double radius, double tol_scale, double tol_scale_root,
                             double max_iter_factor, double *step, int *cg_its) {
    SteihaugCGParams prm;
    prm.tol_scale = tol_scale; prm.tol_scale_root = tol_scale_root;
    prm.max_iter_factor = max_iter_factor;
    std::vector<double> wz(n), wr(n), wd(n), wBd(n), wev(n);
    auto hp = [&](const double *p, double *Bp) {
        for (int i = 0; i < n; ++i) {
            double acc = 0;
            for (int j = 0; j < n; ++j)
                acc += H[i * n + j] * p[j];
            Bp[i] = acc;
        }
    };
    return steihaug_solve(prm, n, n, g, hp, radius, step, wz, wr, wd, wBd, wev, *cg_its);
}

} // extern "C"
