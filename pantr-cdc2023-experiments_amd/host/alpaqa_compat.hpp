// The slice of alpaqa's C++ surface that drivers/mpc-driver.cpp's "pantr", "panoc" and "strucpanoc"
// branches (mpc-driver.cpp:364-384) use,
// re-implemented over the C ABI of include/pantr_mi355x.h.  Same names, argument meaning and
// error behaviour, so a driver written against alpaqa's types reads the same against these:
//
//   alpaqa::TypeErasedProblem<config_t>   get_n/get_m/get_box_C/eval_*   casadi-dll-wrapper.hpp:18, mpc-driver.cpp:131,140
//   typename InnerSolver::Params, ::Direction, Direction::DirectionParams / AcceleratorParams
//   InnerSolver{params, Direction{{.accelerator = ..., .direction = ...}}}  mpc-driver.cpp:50-69
//   ALMSolver{alm_param, inner};  stats = solver(problem, u, y);  stats.status, stats.elapsed_time
//                                                                          mpc-driver.cpp:104-115,217-221
//   solver.get_name()                                                      mpc-driver.cpp:295-296
//   alpaqa::params::set_params(x, "prefix", opts)                          mpc-driver.cpp:54
//
// Vectors are std::vector<double> / std::span<double> (no Eigen in this pipeline).  Errors
// of the C ABI are turned back into the exception types the reference throws.
#pragma once
#include "../../include/pantr_mi355x.h"
#include "params.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <iostream>
#include <memory>
#include <ostream>
#include <span>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

namespace alpaqa {

using real_t   = double;
using length_t = std::ptrdiff_t;
using index_t  = std::ptrdiff_t;
using vec      = std::vector<real_t>;
using rvec     = std::span<real_t>;
using crvec    = std::span<const real_t>;
struct DefaultConfig {};

using pantr::host::ALMParams;
using pantr::host::LBFGSDirectionParams;
using pantr::host::LBFGSParams;
using pantr::host::NewtonTRDirectionParams;
using pantr::host::PANOCParams;
using pantr::host::PANOCStopCrit;
using pantr::host::PANTRParams;
using pantr::host::SolverStatus;
using pantr::host::SteihaugCGParams;
using pantr::host::enum_name;

/// Map a C-ABI error code to the exception the reference would have thrown.
inline void check(int rc) {
    if (rc == PANTR_OK)
        return;
    std::string msg = pantr_last_error();
    switch (rc) {
    case PANTR_ERR_INVALID_ARGUMENT: throw std::invalid_argument(msg);
    case PANTR_ERR_OUT_OF_RANGE: throw std::out_of_range(msg);
    default: throw std::runtime_error(msg);
    }
}

struct Box {
    vec lowerbound, upperbound;
};

/// alpaqa::EvalCounter fields the reference writes (drivers/results.hpp:63-87).
struct EvalCounter {
    unsigned proj_diff_g = 0, proj_multipliers = 0, prox_grad_step = 0, f = 0, grad_f = 0,
             f_grad_f = 0, f_g = 0, grad_f_grad_g_prod = 0, g = 0, grad_g_prod = 0, grad_gi = 0,
             grad_L = 0, hess_L_prod = 0, hess_L = 0, hess_psi_prod = 0, hess_psi = 0, psi = 0,
             grad_psi = 0, psi_grad_psi = 0;
    /// not an alpaqa field: the products the device evaluated (pantr_stats_t::n_hess_psi_prod)
    unsigned hess_psi_prod_evaluated = 0;
    void reset() { *this = {}; }
    void add(const pantr_stats_t &s) {
        prox_grad_step += s.n_prox_grad_step;
        grad_L += s.n_grad_L;
        hess_psi_prod += s.n_hess_psi_prod_calls; // alpaqa counts the call, also where the product is known to be zero
        hess_psi_prod_evaluated += s.n_hess_psi_prod;
        psi += s.n_psi;
        grad_psi += s.n_grad_psi;
        psi_grad_psi += s.n_psi_grad_psi;
    }
};

/// The problem as the solver sees it.  Owns a pantr_batch (device code + workspace) and the
/// problem parameter (initial state) the driver writes through (mpc-driver.cpp:126).
template <class Conf = DefaultConfig>
class TypeErasedProblem {
  public:
    TypeErasedProblem(std::string_view name, length_t horizon, int max_instances = 1, int device = 0) {
        pantr_batch_t *h = nullptr;
        check(pantr_batch_create(std::string(name).c_str(), static_cast<int32_t>(horizon), max_instances,
                                 device, &h));
        adopt(h);
    }
    /// load_benchmark_casadi_problem (drivers/casadi-dll-wrapper.cpp:25-67), data half: the problem
    /// "<dir>/<name>" with the bounds, initial state and initial guess of its sidecar files
    /// "<dir>/<name>_<formulation>_<N>.csv" / ".guess.csv" in place of the compiled-in values.
    struct FromFiles {};
    TypeErasedProblem(FromFiles, const std::string &path, std::string_view formulation, length_t horizon,
                      int max_instances = 1, int device = 0) {
        pantr_batch_t *h = nullptr;
        check(pantr_batch_create_from_files(path.c_str(), std::string(formulation).c_str(),
                                            static_cast<int32_t>(horizon), max_instances, device, &h));
        adopt(h);
    }

  private:
    void adopt(pantr_batch_t *h) {
        handle.reset(h, pantr_batch_destroy);
        int32_t n_, m_, nx_, nu_, nc_, ncN_;
        check(pantr_batch_dims(h, &n_, &m_, &nx_, &nu_, &nc_, &ncN_));
        n = n_; m = m_; nx = nx_; nu = nu_; nc = nc_; nc_N = ncN_;
        param.resize(nx);
        check(pantr_batch_get_initial_state(h, param.data()));
        evaluations = std::make_shared<EvalCounter>();
    }

  public:
    length_t get_n() const { return n; }
    length_t get_m() const { return m; }
    Box get_box_C() const {
        Box b{vec(n), vec(n)};
        check(pantr_batch_get_box_C(handle.get(), b.lowerbound.data(), b.upperbound.data()));
        return b;
    }
    vec get_initial_guess() const {
        vec u(n);
        check(pantr_batch_get_initial_guess(handle.get(), u.data()));
        return u;
    }
    Box get_box_D() const {
        Box b{vec(m), vec(m)};
        check(pantr_batch_get_box_D(handle.get(), b.lowerbound.data(), b.upperbound.data()));
        return b;
    }
    // Problem functions (single instance, evaluated on the device), with alpaqa's argument lists
    // (problems/problems/formulation.hpp:127-176; y and Sigma have size m, empty for m = 0).
    /// psi(x) = f + 1/2 dist^2_Sigma(g + y/Sigma, D); yhat receives Sigma (zeta - clamp(zeta))
    real_t eval_psi(crvec x, crvec y, crvec Sigma, rvec yhat) const {
        real_t v;
        if (m == 0) {
            check(pantr_batch_eval_psi(handle.get(), 1, param.data(), x.data(), &v));
            return v;
        }
        need(y, m, "y"), need(Sigma, m, "Sigma"), need(yhat, m, "yhat");
        vec grad(n);
        check(pantr_batch_eval_psi_grad_psi_al(handle.get(), 1, param.data(), x.data(), y.data(), Sigma.data(), &v,
                                               grad.data(), yhat.data()));
        return v;
    }
    real_t eval_psi(crvec x) const { return eval_psi(x, {}, {}, {}); }
    real_t eval_psi_grad_psi(crvec x, crvec y, crvec Sigma, rvec grad, rvec /*work_n*/, rvec /*work_m*/) const {
        real_t v;
        need(grad, n, "grad");
        if (m == 0) {
            check(pantr_batch_eval_psi_grad_psi(handle.get(), 1, param.data(), x.data(), &v, grad.data()));
            return v;
        }
        need(y, m, "y"), need(Sigma, m, "Sigma");
        check(pantr_batch_eval_psi_grad_psi_al(handle.get(), 1, param.data(), x.data(), y.data(), Sigma.data(), &v,
                                               grad.data(), nullptr));
        return v;
    }
    real_t eval_psi_grad_psi(crvec x, rvec grad) const { return eval_psi_grad_psi(x, {}, {}, grad, {}, {}); }
    /// grad_x L(x, y) = grad f(x) + grad g(x)' y
    void eval_grad_L(crvec x, crvec y, rvec grad, rvec /*work_n*/) const {
        need(grad, n, "grad"), need(y, m, "y");
        check(pantr_batch_eval_grad_L(handle.get(), 1, param.data(), x.data(), m ? y.data() : nullptr, grad.data()));
    }
    /// Hessian-vector product of psi; only scale = 1 exists on the device (formulation.hpp:166 passes 1)
    void eval_hess_psi_prod(crvec x, crvec y, crvec Sigma, real_t scale, crvec v, rvec Hv) const {
        if (scale != 1)
            throw std::invalid_argument("eval_hess_psi_prod: only scale = 1 is available");
        need(v, n, "v"), need(Hv, n, "Hv");
        if (m == 0) {
            check(pantr_batch_eval_hess_psi_prod(handle.get(), 1, param.data(), x.data(), v.data(), Hv.data()));
            return;
        }
        need(y, m, "y"), need(Sigma, m, "Sigma");
        check(pantr_batch_eval_hess_psi_prod_al(handle.get(), 1, param.data(), x.data(), y.data(), Sigma.data(),
                                                v.data(), Hv.data()));
    }
    void eval_hess_psi_prod(crvec x, crvec v, rvec Hv) const { eval_hess_psi_prod(x, {}, {}, 1, v, Hv); }
    /// g(x)
    void eval_g(crvec x, rvec g) const {
        need(g, m, "g");
        check(pantr_batch_eval_g(handle.get(), 1, param.data(), x.data(), g.data()));
    }
    /// p = z - Pi_D(z)
    void eval_proj_diff_g(crvec z, rvec p) const {
        need(z, m, "z"), need(p, m, "p");
        const Box D = get_box_D();
        for (length_t j = 0; j < m; ++j)
            p[j] = z[j] - std::fmax(D.lowerbound[j], std::fmin(z[j], D.upperbound[j]));
    }
    real_t eval_prox_grad_step(real_t gamma, crvec x, crvec grad, rvec xh, rvec p) const {
        real_t s[4];
        check(pantr_batch_eval_prox_grad_step(handle.get(), 1, gamma, x.data(), grad.data(), xh.data(),
                                              p.data(), s));
        return 0; // h(xhat) of a box
    }
    vec simulate(crvec x, crvec u) const {
        vec xn(nx);
        check(pantr_batch_simulate_dynamics(handle.get(), 1, x.data(), u.data(), xn.data()));
        return xn;
    }

    pantr_batch_t *get() const { return handle.get(); }

  private:
    template <class V>
    static void need(const V &v, length_t size, const char *what) {
        if (static_cast<length_t>(v.size()) != size)
            throw std::invalid_argument(std::string("TypeErasedProblem: ") + what + " has the wrong dimension");
    }

  public:
    length_t n = 0, m = 0, nx = 0, nu = 0, nc = 0, nc_N = 0;
    vec param; ///< initial state: CasADiProblem::param in the reference (casadi-dll-wrapper.cpp:64)
    std::shared_ptr<EvalCounter> evaluations;

  private:
    std::shared_ptr<pantr_batch_t> handle;
};

template <class Conf = DefaultConfig>
struct NewtonTRDirection {
    using DirectionParams   = NewtonTRDirectionParams;
    using AcceleratorParams = SteihaugCGParams;
    struct Params {
        AcceleratorParams accelerator{};
        DirectionParams direction{};
    } params;
    NewtonTRDirection() = default;
    NewtonTRDirection(const Params &p) : params(p) {}
    std::string get_name() const { return "NewtonTRDirection"; }
};

template <class DirectionT>
class PANTRSolver {
  public:
    using Params    = PANTRParams;
    using Direction = DirectionT;
    PANTRSolver(const Params &p, Direction d = {}) : params(p), direction(std::move(d)) {}
    std::string get_name() const { return "PANTRSolver<" + direction.get_name() + ">"; }
    /// The typed parameter structs as one option set (what crosses the C ABI as strings).
    void to_options(pantr::host::SolverOptions &o) const {
        o.method = pantr::host::Method::PANTR;
        o.solver = params;
        o.dir    = direction.params.direction;
        o.accel  = direction.params.accelerator;
    }
    Params params;
    Direction direction;
};

/// alpaqa::LBFGSDirection / alpaqa::StructuredLBFGSDirection (mpc-driver.cpp:373,380)
template <class Conf = DefaultConfig>
struct LBFGSDirection {
    using DirectionParams   = LBFGSDirectionParams;
    using AcceleratorParams = LBFGSParams;
    static constexpr auto method = pantr::host::Method::PANOC;
    struct Params {
        AcceleratorParams accelerator{};
        DirectionParams direction{};
    } params;
    LBFGSDirection() = default;
    LBFGSDirection(const Params &p) : params(p) {}
    std::string get_name() const { return "LBFGSDirection"; }
};
template <class Conf = DefaultConfig>
struct StructuredLBFGSDirection {
    using DirectionParams   = LBFGSDirectionParams;
    using AcceleratorParams = LBFGSParams;
    static constexpr auto method = pantr::host::Method::StrucPANOC;
    struct Params {
        AcceleratorParams accelerator{};
        DirectionParams direction{};
    } params;
    StructuredLBFGSDirection() = default;
    StructuredLBFGSDirection(const Params &p) : params(p) {}
    std::string get_name() const { return "StructuredLBFGSDirection"; }
};

/// alpaqa::PANOCSolver<Direction> (mpc-driver.cpp:374,381)
template <class DirectionT>
class PANOCSolver {
  public:
    using Params    = PANOCParams;
    using Direction = DirectionT;
    PANOCSolver(const Params &p, Direction d = {}) : params(p), direction(std::move(d)) {}
    std::string get_name() const { return "PANOCSolver<" + direction.get_name() + ">"; }
    void to_options(pantr::host::SolverOptions &o) const {
        o.method = Direction::method;
        params.split(o.solver, o.panoc);
        o.ldir  = direction.params.direction;
        o.lbfgs = direction.params.accelerator;
    }
    Params params;
    Direction direction;
};

/// solver.print_interval (mpc-driver.cpp:52): the per-iteration values alpaqa's solvers print while they run -
/// here the rows of the device trace (pantr_batch_get_trace), printed after the solve in a fixed table, every
/// `interval`-th iteration and the last one.  One row per iteration k, the values AT THE START of iteration k
/// (eps, psi, ||p||, gamma, phi_gamma of the current iterate) and of the step taken in it (|J|, CG iterations,
/// q_model, rho, accepted / rejected, the radius after its update) - the quantities alpaqa's PANTR progress
/// lines show (phi_gamma, psi, ||p||, gamma, Delta, eps; ||q|| and ||grad psi|| are not traced), so a reference
/// run with the same print_interval can be laid next to this one iteration by iteration.
/// `rows`: [cap][11] doubles (include/pantr_mi355x.h); a row never written has gamma = 0.
inline void print_iterates(std::ostream &os, bool pantr, const double *rows, int cap, int interval, int precision,
                           const char *note, const pantr_stats_t &stats) {
    int count = 0;
    while (count < cap && rows[(size_t)count * 11 + 3] != 0)
        ++count;
    const int w = precision + 8;
    auto num = [&](double v) {
        char buf[64];
        std::snprintf(buf, sizeof buf, "%+*.*e", w, precision, v);
        return std::string(buf);
    };
    auto head = [&](const char *h) {
        char buf[64];
        std::snprintf(buf, sizeof buf, "%*s", w, h);
        return std::string(buf);
    };
    const char *tag = pantr ? "[PANTR] " : "[PANOC] ";
    os << tag << "  iter " << head("phi_gamma") << ' ' << head("psi") << ' ' << head("||p||") << ' ' << head("gamma")
       << ' ' << head("eps");
    if (pantr)
        os << ' ' << head("Delta") << ' ' << head("rho") << ' ' << head("q_model") << "  |J|   cg";
    else
        os << ' ' << head("tau") << " lbfgs";
    os << (note ? note : "") << '\n';
    for (int k = 0; k < count; ++k) {
        if (k % interval != 0 && k != count - 1)
            continue;
        const double *r = rows + (size_t)k * 11;
        char it[16];
        std::snprintf(it, sizeof it, "%6d ", k);
        os << tag << it << num(pantr ? r[7] : r[5]) << ' ' << num(r[1]) << ' ' << num(r[2]) << ' ' << num(r[3]) << ' '
           << num(r[0]);
        if (pantr) {
            char jc[32];
            std::snprintf(jc, sizeof jc, " %4d %4d", (int)r[8], (int)r[9]);
            os << ' ' << num(r[4]) << ' ' << num(r[5]) << ' ' << num(r[6]) << jc
               << (r[10] != 0 ? "  accepted" : "  rejected");
        } else {
            char lb[32];
            std::snprintf(lb, sizeof lb, " %5d", (int)r[6]);
            os << ' ' << num(r[4]) << lb;
        }
        os << '\n';
    }
    if (count == cap)
        os << tag << "(trace full: " << cap << " iterations kept)\n";
    // the exit: the iterate the stop criterion accepted takes no step and has no row
    os << tag << "  exit " << enum_name(static_cast<SolverStatus>(stats.status)) << " after " << stats.iterations
       << " iterations" << (note ? " (all inner solves)" : "") << ", eps = " << num(stats.eps) << ", psi = "
       << num(stats.final_psi) << ", gamma = " << num(stats.final_gamma) << '\n';
    os.flush();
}

template <class InnerSolverT>
class ALMSolver {
  public:
    using Params      = ALMParams;
    using InnerSolver = InnerSolverT;
    struct Stats {
        SolverStatus status = SolverStatus::Busy;
        std::chrono::nanoseconds elapsed_time{};
        unsigned outer_iterations = 0;
        real_t eps = 0, delta = 0;
        pantr_stats_t inner{};
    };
    ALMSolver(const Params &p, InnerSolver inner) : params(p), inner_solver(std::move(inner)) {}
    std::string get_name() const { return "ALMSolver<" + inner_solver.get_name() + ">"; }

    /// stats = solver(problem, x, y)   (mpc-driver.cpp:217)
    template <class Conf>
    Stats operator()(TypeErasedProblem<Conf> &problem, rvec x, rvec y) {
        if (static_cast<length_t>(x.size()) != problem.get_n())
            throw std::invalid_argument("ALMSolver: x has the wrong dimension");
        if (static_cast<length_t>(y.size()) != problem.get_m())
            throw std::invalid_argument("ALMSolver: y has the wrong dimension");
        push_options(problem.get());
        pantr_stats_t s{};
        // solver.print_interval: trace this solve on the device and print its rows afterwards (print_iterates)
        const int interval = inner_solver.params.print_interval;
        const int cap      = interval > 0 ? std::min<int>(inner_solver.params.max_iter, 100000) + 1 : 0;
        if (interval > 0)
            check(pantr_batch_set_trace(problem.get(), 1, cap)); // (also clears the rows of the solve before)
        check(pantr_batch_solve(problem.get(), 1, problem.param.data(), x.data(), y.data(), &s));
        if (interval > 0) {
            std::vector<double> rows((size_t)cap * 11);
            check(pantr_batch_get_trace(problem.get(), rows.data()));
            pantr::host::SolverOptions o;
            inner_solver.to_options(o);
            print_iterates(std::cout, o.method == pantr::host::Method::PANTR, rows.data(), cap, interval,
                           inner_solver.params.print_precision,
                           problem.get_m() > 0 ? "   (first inner solve of the ALM)" : nullptr, s);
        }
        problem.evaluations->add(s);
        return convert(s);
    }

    /// Batched form: B independent instances (x0s[B][nx], xs[B][n] and ys[B][m] in/out).
    template <class Conf>
    std::vector<Stats> solve_batch(TypeErasedProblem<Conf> &problem, length_t B, crvec x0s, rvec xs, rvec ys = {}) {
        push_options(problem.get());
        std::vector<pantr_stats_t> s(B);
        check(pantr_batch_solve(problem.get(), static_cast<int32_t>(B), x0s.data(), xs.data(),
                                ys.empty() ? nullptr : ys.data(), s.data()));
        std::vector<Stats> out;
        for (auto &e : s) {
            problem.evaluations->add(e);
            out.push_back(convert(e));
        }
        return out;
    }

    Params params;
    InnerSolver inner_solver;

  private:
    static Stats convert(const pantr_stats_t &s) {
        Stats r;
        r.status           = static_cast<SolverStatus>(s.status);
        r.elapsed_time     = std::chrono::nanoseconds(static_cast<long long>(s.elapsed_s * 1e9));
        r.outer_iterations = static_cast<unsigned>(s.outer_iterations);
        r.eps              = s.eps;
        r.delta            = s.delta;
        r.inner            = s;
        return r;
    }
    /// Hand the typed parameter structs to the handle through the string interface of the ABI.
    void push_options(pantr_batch_t *b) const {
        pantr::host::SolverOptions o;
        inner_solver.to_options(o);
        o.alm = params;
        check(pantr_batch_set_param(b, (std::string("method=") + enum_name(o.method)).c_str()));
        for (const char *key : pantr::host::kOptionKeys)
            check(pantr_batch_set_param(b, pantr::host::format_option(o, key).c_str()));
    }
};

namespace params {

namespace detail {
/// Apply the options addressed to `prefix` to the member of SolverOptions named `canonical`
/// (the caller may choose any prefix, mpc-driver.cpp:54 uses "solver", :61 "dir", ...).
template <class T>
void set_struct(T &p, T pantr::host::SolverOptions::*member, std::string_view canonical,
                std::string_view prefix, std::span<const std::string_view> opts) {
    pantr::host::SolverOptions o;
    o.*member = p;
    const std::string pfx = std::string(prefix) + ".";
    for (std::string_view kv : opts) {
        if (kv.substr(0, pfx.size()) != pfx)
            continue; // addressed to another object
        std::string rewritten = std::string(canonical) + "." + std::string(kv.substr(pfx.size()));
        std::string err;
        if (pantr::host::set_option(o, rewritten, err) != pantr::host::SetResult::Ok)
            throw std::invalid_argument(err);
    }
    p = o.*member;
}
} // namespace detail

/// set_params(struct, "prefix", opts): applies the "prefix.key=value" options to the struct.
inline void set_params(PANTRParams &p, std::string_view prefix, std::span<const std::string_view> opts) {
    detail::set_struct(p, &pantr::host::SolverOptions::solver, "solver", prefix, opts);
}
inline void set_params(NewtonTRDirectionParams &p, std::string_view prefix,
                       std::span<const std::string_view> opts) {
    detail::set_struct(p, &pantr::host::SolverOptions::dir, "dir", prefix, opts);
}
inline void set_params(SteihaugCGParams &p, std::string_view prefix, std::span<const std::string_view> opts) {
    detail::set_struct(p, &pantr::host::SolverOptions::accel, "accel", prefix, opts);
}
inline void set_params(ALMParams &p, std::string_view prefix, std::span<const std::string_view> opts) {
    detail::set_struct(p, &pantr::host::SolverOptions::alm, "alm", prefix, opts);
}
inline void set_params(LBFGSDirectionParams &p, std::string_view prefix, std::span<const std::string_view> opts) {
    detail::set_struct(p, &pantr::host::SolverOptions::ldir, "dir", prefix, opts);
}
inline void set_params(LBFGSParams &p, std::string_view prefix, std::span<const std::string_view> opts) {
    detail::set_struct(p, &pantr::host::SolverOptions::lbfgs, "accel", prefix, opts);
}
inline void set_params(PANOCParams &p, std::string_view prefix, std::span<const std::string_view> opts) {
    pantr::host::SolverOptions o;
    p.split(o.solver, o.panoc);
    const std::string pfx = std::string(prefix) + ".";
    for (std::string_view kv : opts) {
        if (kv.substr(0, pfx.size()) != pfx)
            continue;
        std::string err;
        if (pantr::host::set_option(o, "solver." + std::string(kv.substr(pfx.size())), err) !=
            pantr::host::SetResult::Ok)
            throw std::invalid_argument(err);
    }
    p.join(o.solver, o.panoc);
}

/// Scalars, strings and vectors the driver itself reads ("num_sim=60", "u_dist=-1,1,1", ...):
/// the key is the whole prefix (mpc-driver.cpp:144-157,316,360).
inline std::string_view find_value(std::string_view key, std::span<const std::string_view> opts, bool &found) {
    found = false;
    std::string_view val;
    for (auto kv : opts) {
        auto eq = kv.find('=');
        if (eq != std::string_view::npos && kv.substr(0, eq) == key) {
            val   = kv.substr(eq + 1);
            found = true; // last one wins (dodo.py appends overrides, :101-105)
        }
    }
    return val;
}
inline void set_params(length_t &v, std::string_view key, std::span<const std::string_view> opts) {
    bool found;
    auto s = find_value(key, opts, found);
    if (!found)
        return;
    int32_t i;
    if (!pantr::host::parse_int(s, i))
        throw std::invalid_argument("Invalid value '" + std::string(s) + "' for '" + std::string(key) + "'");
    v = i;
}
inline void set_params(bool &v, std::string_view key, std::span<const std::string_view> opts) {
    bool found;
    auto s = find_value(key, opts, found);
    if (found && !pantr::host::parse_bool(s, v))
        throw std::invalid_argument("Invalid value '" + std::string(s) + "' for '" + std::string(key) + "'");
}
inline void set_params(std::string &v, std::string_view key, std::span<const std::string_view> opts) {
    bool found;
    auto s = find_value(key, opts, found);
    if (found)
        v = std::string(s);
}
inline void set_params(std::string_view &v, std::string_view key, std::span<const std::string_view> opts) {
    bool found;
    auto s = find_value(key, opts, found);
    if (found)
        v = s;
}
/// Comma-separated vector; must keep its size (mpc-driver.cpp:144-146 checks it afterwards).
inline void set_params(vec &v, std::string_view key, std::span<const std::string_view> opts) {
    bool found;
    auto s = find_value(key, opts, found);
    if (!found)
        return;
    vec out;
    while (true) {
        auto comma = s.find(',');
        double d;
        if (!pantr::host::parse_double(s.substr(0, comma), d))
            throw std::invalid_argument("Invalid value in '" + std::string(key) + "'");
        out.push_back(d);
        if (comma == std::string_view::npos)
            break;
        s = s.substr(comma + 1);
    }
    v = out;
}

} // namespace params
} // namespace alpaqa
