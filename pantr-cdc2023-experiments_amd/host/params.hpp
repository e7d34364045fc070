// Solver option structs and the "prefix.key=value" parser.
//
// Mirrors the part of alpaqa's option surface the reference exercises (the "solver", "dir" and
// "accel" prefixes accept the union of the keys of the three methods' parameter structs; a key of
// another method is stored and has no effect):
//   alpaqa::params::set_params(obj, prefix, opts)  drivers/mpc-driver.cpp:54,61,62,113
//   PANTRParams / NewtonTRDirectionParams / SteihaugCGParams / ALMParams
//       values set in new-benchmarks-paper/dodo.py:12-33,98-105 and
//       benchmarks-paper/Makefile:218-224; driver defaults mpc-driver.cpp:51-53,109-111
// Field names and defaults follow upstream alpaqa (develop, mid-2023); the structs are plain
// data so they can cross the C ABI by value.
#pragma once
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <limits>
#include <ostream>
#include <string>
#include <string_view>

namespace pantr::host {

inline constexpr double kEps = std::numeric_limits<double>::epsilon();
inline constexpr double kNaN = std::numeric_limits<double>::quiet_NaN();
inline constexpr double kInf = std::numeric_limits<double>::infinity();

/// alpaqa::PANOCStopCrit
enum class PANOCStopCrit : int32_t {
    ApproxKKT = 0, ApproxKKT2, ProjGradNorm, ProjGradNorm2, ProjGradUnitNorm, ProjGradUnitNorm2,
    FPRNorm, FPRNorm2, Ipopt, LBFGSBpp,
};
inline const char *enum_name(PANOCStopCrit s) {
    static const char *names[] = {"ApproxKKT", "ApproxKKT2", "ProjGradNorm", "ProjGradNorm2",
                                  "ProjGradUnitNorm", "ProjGradUnitNorm2", "FPRNorm", "FPRNorm2",
                                  "Ipopt", "LBFGSBpp"};
    return names[static_cast<int>(s)];
}

/// alpaqa::SolverStatus (printed at mpc-driver.cpp:218)
enum class SolverStatus : int32_t {
    Busy = 0, Converged, MaxTime, MaxIter, NotFinite, NoProgress, Interrupted, Exception,
};
inline const char *enum_name(SolverStatus s) {
    static const char *names[] = {"Busy", "Converged", "MaxTime", "MaxIter", "NotFinite",
                                  "NoProgress", "Interrupted", "Exception"};
    int i = static_cast<int>(s);
    return i >= 0 && i < 8 ? names[i] : "<unknown>";
}

inline std::ostream &operator<<(std::ostream &os, SolverStatus s) { return os << enum_name(s); }
inline std::ostream &operator<<(std::ostream &os, PANOCStopCrit s) { return os << enum_name(s); }

struct LipschitzEstimateParams {
    double L_0 = 0, epsilon = 1e-6, delta = 1e-12, Lgamma_factor = 0.95;
};

struct PANTRParams {
    LipschitzEstimateParams Lipschitz;
    int32_t max_iter      = 100;
    double max_time       = 300; ///< seconds (alpaqa: 5 min)
    double L_min = 1e-5, L_max = 1e20;
    PANOCStopCrit stop_crit = PANOCStopCrit::ApproxKKT;
    int32_t max_no_progress = 10;
    int32_t print_interval = 0, print_precision = 6;
    double quadratic_upperbound_tolerance_factor = 10 * kEps;
    double TR_tolerance_factor                   = 10 * kEps;
    double ratio_threshold_acceptable = 0.2, ratio_threshold_good = 0.8;
    double radius_factor_rejected = 0.35, radius_factor_acceptable = 0.999, radius_factor_good = 2.5;
    double initial_radius = kNaN, min_radius = 100 * kEps;
    bool compute_ratio_using_new_stepsize               = false;
    bool update_direction_on_prox_step                  = true;
    bool recompute_last_prox_step_after_direction_reset = false;
    bool disable_acceleration                           = false;
    bool ratio_approx_fbe_quadratic_model               = true;
};

struct NewtonTRDirectionParams {
    double hessian_vec_factor   = 0.5;
    bool finite_diff            = false;
    double finite_diff_stepsize = 1.4901161193847656e-08; // sqrt(eps)
};

struct SteihaugCGParams {
    double tol_scale = 1, tol_scale_root = 0.5, tol_max = kInf, max_iter_factor = 1;
};

/// The solver `method=` selects (mpc-driver.cpp:360-384; default there: fbetrust, not available).
enum class Method : int32_t { PANTR = 0, PANOC = 1, StrucPANOC = 2 };
inline const char *enum_name(Method m) {
    static const char *names[] = {"pantr", "panoc", "strucpanoc"};
    return names[static_cast<int>(m)];
}

/// The fields of alpaqa::PANOCParams that PANTRParams does not have ("solver.*" with method=panoc,
/// strucpanoc; the shared ones are read from PANTRParams).
struct PANOCExtraParams {
    double min_linesearch_coefficient   = 1.0 / 256;
    bool force_linesearch               = false;
    double linesearch_strictness_factor = 0.95;
    double linesearch_tolerance_factor  = 10 * kEps;
    bool update_direction_in_candidate                    = false;
    bool recompute_last_prox_step_after_stepsize_change   = false;
};
/// alpaqa::PANOCParams as one struct (what make_inner_solver<PANOCSolver<...>> fills,
/// mpc-driver.cpp:49-54); split() / join() map it onto the shared and the extra part.
struct PANOCParams {
    LipschitzEstimateParams Lipschitz;
    int32_t max_iter = 100;
    double max_time  = 300;
    double min_linesearch_coefficient   = 1.0 / 256;
    bool force_linesearch               = false;
    double linesearch_strictness_factor = 0.95;
    double L_min = 1e-5, L_max = 1e20;
    PANOCStopCrit stop_crit = PANOCStopCrit::ApproxKKT;
    int32_t max_no_progress = 10;
    int32_t print_interval = 0, print_precision = 6;
    double quadratic_upperbound_tolerance_factor = 10 * kEps;
    double linesearch_tolerance_factor           = 10 * kEps;
    bool update_direction_in_candidate                  = false;
    bool recompute_last_prox_step_after_stepsize_change = false;

    void split(PANTRParams &c, PANOCExtraParams &e) const {
        c.Lipschitz = Lipschitz; c.max_iter = max_iter; c.max_time = max_time; c.L_min = L_min; c.L_max = L_max;
        c.stop_crit = stop_crit; c.max_no_progress = max_no_progress; c.print_interval = print_interval;
        c.print_precision = print_precision;
        c.quadratic_upperbound_tolerance_factor = quadratic_upperbound_tolerance_factor;
        e.min_linesearch_coefficient = min_linesearch_coefficient; e.force_linesearch = force_linesearch;
        e.linesearch_strictness_factor = linesearch_strictness_factor;
        e.linesearch_tolerance_factor = linesearch_tolerance_factor;
        e.update_direction_in_candidate = update_direction_in_candidate;
        e.recompute_last_prox_step_after_stepsize_change = recompute_last_prox_step_after_stepsize_change;
    }
    void join(const PANTRParams &c, const PANOCExtraParams &e) {
        Lipschitz = c.Lipschitz; max_iter = c.max_iter; max_time = c.max_time; L_min = c.L_min; L_max = c.L_max;
        stop_crit = c.stop_crit; max_no_progress = c.max_no_progress; print_interval = c.print_interval;
        print_precision = c.print_precision;
        quadratic_upperbound_tolerance_factor = c.quadratic_upperbound_tolerance_factor;
        min_linesearch_coefficient = e.min_linesearch_coefficient; force_linesearch = e.force_linesearch;
        linesearch_strictness_factor = e.linesearch_strictness_factor;
        linesearch_tolerance_factor = e.linesearch_tolerance_factor;
        update_direction_in_candidate = e.update_direction_in_candidate;
        recompute_last_prox_step_after_stepsize_change = e.recompute_last_prox_step_after_stepsize_change;
    }
};
/// alpaqa::LBFGSParams ("accel.*" with method=panoc, strucpanoc; dodo.py:35,38 set memory=50)
struct LBFGSParams {
    int32_t memory     = 10;
    double min_div_fac = 1e-10, min_abs_s = 1e-32;
    double cbfgs_alpha = 1, cbfgs_epsilon = 0;
    bool force_pos_def = true;
};
/// alpaqa::LBFGSDirectionParams / StructuredLBFGSDirectionParams ("dir.*" with method=panoc,
/// strucpanoc; dodo.py:39 sets hessian_vec=0)
struct LBFGSDirectionParams {
    bool rescale_on_step_size_changes = false;
    bool hessian_vec                  = true; ///< structured direction: Hessian-vector term for the active set
};

struct ALMParams {
    double tolerance = 1e-5, dual_tolerance = 1e-5;
    double penalty_update_factor = 10, initial_penalty = 1, initial_penalty_factor = 20;
    double initial_tolerance = 1, tolerance_update_factor = 1e-1;
    double rel_penalty_increase_threshold = 0.1;
    double max_multiplier = 1e9, max_penalty = 1e9, min_penalty = 1e-9;
    int32_t max_iter = 100;
    double max_time  = 300; ///< seconds
    int32_t print_interval = 0, print_precision = 6;
    bool single_penalty_factor = false;
};

/// Everything one solver instance is configured by.
struct SolverOptions {
    Method method = Method::PANTR;
    /// `speculation=auto|on|off` (an option of this implementation, spelled like `method=`), method=pantr:
    /// a second wavefront per instance evaluates the successor of a rejected candidate while the solver
    /// works on the trust-region subproblem (csrc/pantr_kernel.hip.h: SpecBox; csrc/pantr_alm.hip.h for the
    /// quadcopter family).  Same results either way; auto = on for batches small enough to leave the GPU's
    /// SIMDs idle otherwise.  -1 auto, 0 off, 1 on.
    int32_t speculation = -1;
    /// `residency=auto|<k>` (an option of this implementation): at most k workgroups of a solve kernel resident
    /// per compute unit (the launch asks for as much LDS as leaves room for k).  Large batches stream their
    /// second-order tapes; fewer instances in flight keep that working set inside the 256 MiB Infinity Cache.
    /// 0 = auto (what the LDS of the kernel allows).  Same results either way.
    int32_t residency = 0;
    /// `launch_order=natural|previous|auto` (an option of this implementation): in which order the workgroups of a
    /// launch start the instances.  natural (default): by index.  previous: longest first by the time each instance
    /// took in the solve before it on this handle - for batches that are solved AGAIN (replays, parameter studies,
    /// repeated passes): with more workgroups than fit the chip the pass otherwise ends with whatever long instance
    /// happened to start last (the bench batch tiled eight times: 170 -> 130 ms).  auto: previous wherever the launch
    /// has more workgroups than the chip holds at once.  A genuine closed loop has no use for it: what an instance
    /// needed a time step earlier says nothing about what it needs now (4800 controllers, 16 steps: rank correlation
    /// between consecutive steps -0.3 ... 0.3, 1.79 s either way - profiles/r05_closed_loop_order.txt), hence the
    /// default.  pantr_batch_set_launch_order overrides all three.  Results do not depend on it.
    /// -1 = auto, 0 = natural, 1 = previous.
    int32_t launch_order = 0;
    /// `spec_patience=<duration>` (an option of this implementation): how long a solver wavefront waits for a
    /// result of its helper before it evaluates the step itself, tells the helper to stop and posts no further
    /// requests (pantr_stats_t::spec_timeouts counts these).  Same results either way.
    double spec_patience = 1.0;
    PANTRParams solver;
    NewtonTRDirectionParams dir;
    SteihaugCGParams accel;
    PANOCExtraParams panoc;   ///< "solver.*" keys of PANOCParams
    LBFGSDirectionParams ldir; ///< "dir.*" keys of the L-BFGS directions
    LBFGSParams lbfgs;        ///< "accel.*" keys of LBFGSParams
    ALMParams alm;
    /// make_inner_solver / make_solver defaults of the reference driver
    /// (mpc-driver.cpp:51-53 and :109-111), applied before the CLI options.
    static SolverOptions driver_defaults() {
        SolverOptions o;
        o.solver.max_iter       = 50'000;
        o.solver.print_interval = 0;
        o.solver.stop_crit      = PANOCStopCrit::FPRNorm;
        o.alm.max_iter          = 200;
        o.alm.tolerance         = 1e-8;
        o.alm.dual_tolerance    = 1e-8;
        o.alm.print_interval    = 0;
        return o;
    }
};

// ---------------------------------------------------------------------------------------
// value parsers
// ---------------------------------------------------------------------------------------
inline bool parse_double(std::string_view s, double &out) {
    std::string tmp(s);
    if (tmp.empty())
        return false;
    char *end = nullptr;
    errno     = 0;
    double v  = std::strtod(tmp.c_str(), &end);
    if (end == tmp.c_str() || *end != '\0')
        return false;
    out = v;
    return true;
}
inline bool parse_int(std::string_view s, int32_t &out) {
    std::string tmp(s);
    if (tmp.empty())
        return false;
    char *end   = nullptr;
    long long v = std::strtoll(tmp.c_str(), &end, 10);
    if (end == tmp.c_str() || *end != '\0') {
        // accept integral floating notation such as 5e4
        double d;
        if (!parse_double(s, d) || d != std::floor(d))
            return false;
        v = static_cast<long long>(d);
    }
    if (v < std::numeric_limits<int32_t>::min() || v > std::numeric_limits<int32_t>::max())
        return false;
    out = static_cast<int32_t>(v);
    return true;
}
inline bool parse_bool(std::string_view s, bool &out) {
    if (s == "true" || s == "1" || s == "True") {
        out = true;
        return true;
    }
    if (s == "false" || s == "0" || s == "False") {
        out = false;
        return true;
    }
    return false;
}
/// "30s", "500ms", "2min", "1h", "250us", "10ns"; a bare number is seconds.
inline bool parse_duration(std::string_view s, double &seconds) {
    static const struct {
        const char *suffix;
        double scale;
    } units[] = {{"ns", 1e-9}, {"us", 1e-6}, {"µs", 1e-6}, {"ms", 1e-3}, {"min", 60}, {"h", 3600}, {"s", 1}};
    for (const auto &u : units) {
        size_t len = std::strlen(u.suffix);
        if (s.size() > len && s.substr(s.size() - len) == u.suffix) {
            double v;
            if (!parse_double(s.substr(0, s.size() - len), v))
                return false;
            seconds = v * u.scale;
            return true;
        }
    }
    return parse_double(s, seconds);
}
inline bool parse_stop_crit(std::string_view s, PANOCStopCrit &out) {
    for (int i = 0; i <= static_cast<int>(PANOCStopCrit::LBFGSBpp); ++i)
        if (s == enum_name(static_cast<PANOCStopCrit>(i))) {
            out = static_cast<PANOCStopCrit>(i);
            return true;
        }
    return false;
}

// ---------------------------------------------------------------------------------------
// key table
// ---------------------------------------------------------------------------------------
enum class FieldType { Double, Int, Bool, Duration, StopCrit };
struct FieldRef {
    FieldType type;
    void *ptr;
};

/// Resolve "prefix.key" to a field; returns false if the prefix is ours but the key is not.
/// `known_prefix` tells the caller whether the string was addressed to one of our structs.
inline bool find_field(SolverOptions &o, std::string_view full_key, FieldRef &ref, bool &known_prefix) {
    auto dot = full_key.find('.');
    known_prefix = false;
    if (dot == std::string_view::npos)
        return false;
    std::string_view prefix = full_key.substr(0, dot), key = full_key.substr(dot + 1);
#define PANTR_FIELD(name, T, member)                                                               \
    if (key == name) {                                                                             \
        ref = {FieldType::T, &(member)};                                                           \
        return true;                                                                               \
    }
    if (prefix == "solver") {
        known_prefix = true;
        auto &s = o.solver;
        PANTR_FIELD("Lipschitz.L_0", Double, s.Lipschitz.L_0)
        PANTR_FIELD("Lipschitz.ε", Double, s.Lipschitz.epsilon)
        PANTR_FIELD("Lipschitz.epsilon", Double, s.Lipschitz.epsilon)
        PANTR_FIELD("Lipschitz.δ", Double, s.Lipschitz.delta)
        PANTR_FIELD("Lipschitz.delta", Double, s.Lipschitz.delta)
        PANTR_FIELD("Lipschitz.Lγ_factor", Double, s.Lipschitz.Lgamma_factor)
        PANTR_FIELD("Lipschitz.Lgamma_factor", Double, s.Lipschitz.Lgamma_factor)
        PANTR_FIELD("max_iter", Int, s.max_iter)
        PANTR_FIELD("max_time", Duration, s.max_time)
        PANTR_FIELD("L_min", Double, s.L_min)
        PANTR_FIELD("L_max", Double, s.L_max)
        PANTR_FIELD("stop_crit", StopCrit, s.stop_crit)
        PANTR_FIELD("max_no_progress", Int, s.max_no_progress)
        PANTR_FIELD("print_interval", Int, s.print_interval)
        PANTR_FIELD("print_precision", Int, s.print_precision)
        PANTR_FIELD("quadratic_upperbound_tolerance_factor", Double, s.quadratic_upperbound_tolerance_factor)
        PANTR_FIELD("TR_tolerance_factor", Double, s.TR_tolerance_factor)
        PANTR_FIELD("ratio_threshold_acceptable", Double, s.ratio_threshold_acceptable)
        PANTR_FIELD("ratio_threshold_good", Double, s.ratio_threshold_good)
        PANTR_FIELD("radius_factor_rejected", Double, s.radius_factor_rejected)
        PANTR_FIELD("radius_factor_acceptable", Double, s.radius_factor_acceptable)
        PANTR_FIELD("radius_factor_good", Double, s.radius_factor_good)
        PANTR_FIELD("initial_radius", Double, s.initial_radius)
        PANTR_FIELD("min_radius", Double, s.min_radius)
        PANTR_FIELD("compute_ratio_using_new_stepsize", Bool, s.compute_ratio_using_new_stepsize)
        PANTR_FIELD("update_direction_on_prox_step", Bool, s.update_direction_on_prox_step)
        PANTR_FIELD("recompute_last_prox_step_after_direction_reset", Bool,
                    s.recompute_last_prox_step_after_direction_reset)
        PANTR_FIELD("disable_acceleration", Bool, s.disable_acceleration)
        PANTR_FIELD("ratio_approx_fbe_quadratic_model", Bool, s.ratio_approx_fbe_quadratic_model)
        auto &pn = o.panoc; // PANOCParams
        PANTR_FIELD("min_linesearch_coefficient", Double, pn.min_linesearch_coefficient)
        PANTR_FIELD("force_linesearch", Bool, pn.force_linesearch)
        PANTR_FIELD("linesearch_strictness_factor", Double, pn.linesearch_strictness_factor)
        PANTR_FIELD("linesearch_tolerance_factor", Double, pn.linesearch_tolerance_factor)
        PANTR_FIELD("update_direction_in_candidate", Bool, pn.update_direction_in_candidate)
        PANTR_FIELD("recompute_last_prox_step_after_stepsize_change", Bool,
                    pn.recompute_last_prox_step_after_stepsize_change)
        return false;
    }
    if (prefix == "dir") {
        known_prefix = true;
        auto &d = o.dir;
        PANTR_FIELD("hessian_vec_factor", Double, d.hessian_vec_factor)
        PANTR_FIELD("finite_diff", Bool, d.finite_diff)
        PANTR_FIELD("finite_diff_stepsize", Double, d.finite_diff_stepsize)
        auto &ld = o.ldir; // L-BFGS directions
        PANTR_FIELD("rescale_on_step_size_changes", Bool, ld.rescale_on_step_size_changes)
        PANTR_FIELD("hessian_vec", Bool, ld.hessian_vec)
        return false;
    }
    if (prefix == "accel") {
        known_prefix = true;
        auto &c = o.accel;
        PANTR_FIELD("tol_scale", Double, c.tol_scale)
        PANTR_FIELD("tol_scale_root", Double, c.tol_scale_root)
        PANTR_FIELD("tol_max", Double, c.tol_max)
        PANTR_FIELD("max_iter_factor", Double, c.max_iter_factor)
        auto &l = o.lbfgs; // LBFGSParams
        PANTR_FIELD("memory", Int, l.memory)
        PANTR_FIELD("min_div_fac", Double, l.min_div_fac)
        PANTR_FIELD("min_abs_s", Double, l.min_abs_s)
        PANTR_FIELD("cbfgs.α", Double, l.cbfgs_alpha)
        PANTR_FIELD("cbfgs.alpha", Double, l.cbfgs_alpha)
        PANTR_FIELD("cbfgs.ε", Double, l.cbfgs_epsilon)
        PANTR_FIELD("cbfgs.epsilon", Double, l.cbfgs_epsilon)
        PANTR_FIELD("force_pos_def", Bool, l.force_pos_def)
        return false;
    }
    if (prefix == "alm") {
        known_prefix = true;
        auto &m = o.alm;
        PANTR_FIELD("tolerance", Double, m.tolerance)
        PANTR_FIELD("dual_tolerance", Double, m.dual_tolerance)
        PANTR_FIELD("penalty_update_factor", Double, m.penalty_update_factor)
        PANTR_FIELD("initial_penalty", Double, m.initial_penalty)
        PANTR_FIELD("initial_penalty_factor", Double, m.initial_penalty_factor)
        PANTR_FIELD("initial_tolerance", Double, m.initial_tolerance)
        PANTR_FIELD("tolerance_update_factor", Double, m.tolerance_update_factor)
        PANTR_FIELD("rel_penalty_increase_threshold", Double, m.rel_penalty_increase_threshold)
        PANTR_FIELD("max_multiplier", Double, m.max_multiplier)
        PANTR_FIELD("max_penalty", Double, m.max_penalty)
        PANTR_FIELD("min_penalty", Double, m.min_penalty)
        PANTR_FIELD("max_iter", Int, m.max_iter)
        PANTR_FIELD("max_time", Duration, m.max_time)
        PANTR_FIELD("print_interval", Int, m.print_interval)
        PANTR_FIELD("print_precision", Int, m.print_precision)
        PANTR_FIELD("single_penalty_factor", Bool, m.single_penalty_factor)
        return false;
    }
#undef PANTR_FIELD
    return false;
}

enum class SetResult { Ok, Ignored, UnknownKey, BadValue, Malformed };

/// Apply one "prefix.key=value" option.  Options for other objects are ignored, as
/// alpaqa::params::set_params ignores keys whose prefix is not the one it was given.
inline SetResult set_option(SolverOptions &o, std::string_view kv, std::string &err) {
    if (kv.substr(0, 7) == "method=") { // mpc-driver.cpp:360-361
        std::string_view v = kv.substr(7);
        for (Method m : {Method::PANTR, Method::PANOC, Method::StrucPANOC})
            if (v == enum_name(m)) {
                o.method = m;
                return SetResult::Ok;
            }
        err = "Unknown solver '" + std::string(v) + "' (available: pantr, panoc, strucpanoc)";
        return SetResult::BadValue;
    }
    if (kv.substr(0, 12) == "speculation=") {
        std::string_view v = kv.substr(12);
        if (v == "auto" || v == "on" || v == "off" || v == "1" || v == "0") {
            o.speculation = v == "auto" ? -1 : (v == "on" || v == "1") ? 1 : 0;
            return SetResult::Ok;
        }
        err = "Invalid value '" + std::string(v) + "' for 'speculation' (auto, on, off)";
        return SetResult::BadValue;
    }
    if (kv.substr(0, 14) == "spec_patience=") {
        if (parse_duration(kv.substr(14), o.spec_patience) && o.spec_patience >= 0)
            return SetResult::Ok;
        err = "Invalid value '" + std::string(kv.substr(14)) + "' for 'spec_patience' (a duration, e.g. 500ms)";
        return SetResult::BadValue;
    }
    if (kv.substr(0, 10) == "residency=") {
        std::string_view v = kv.substr(10);
        int32_t k = 0;
        if (v == "auto" || (parse_int(v, k) && k >= 0 && k <= 32)) {
            o.residency = v == "auto" ? 0 : k;
            return SetResult::Ok;
        }
        err = "Invalid value '" + std::string(v) + "' for 'residency' (auto, or workgroups per compute unit 1..32)";
        return SetResult::BadValue;
    }
    if (kv.substr(0, 13) == "launch_order=") {
        std::string_view v = kv.substr(13);
        if (v == "auto" || v == "natural" || v == "previous") {
            o.launch_order = v == "auto" ? -1 : v == "previous" ? 1 : 0;
            return SetResult::Ok;
        }
        err = "Invalid value '" + std::string(v) + "' for 'launch_order' (auto, natural, previous)";
        return SetResult::BadValue;
    }
    auto eq = kv.find('=');
    if (eq == std::string_view::npos) {
        auto dot = kv.find('.');
        std::string_view prefix = dot == std::string_view::npos ? kv : kv.substr(0, dot);
        if (prefix == "solver" || prefix == "dir" || prefix == "accel" || prefix == "alm") {
            err = "Option '" + std::string(kv) + "' is missing '=value'";
            return SetResult::Malformed;
        }
        return SetResult::Ignored;
    }
    std::string_view key = kv.substr(0, eq), value = kv.substr(eq + 1);
    FieldRef ref{};
    bool known_prefix = false;
    if (!find_field(o, key, ref, known_prefix)) {
        if (!known_prefix)
            return SetResult::Ignored;
        err = "Invalid key '" + std::string(key) + "'";
        return SetResult::UnknownKey;
    }
    bool ok = false;
    switch (ref.type) {
    case FieldType::Double: ok = parse_double(value, *static_cast<double *>(ref.ptr)); break;
    case FieldType::Int: ok = parse_int(value, *static_cast<int32_t *>(ref.ptr)); break;
    case FieldType::Bool: ok = parse_bool(value, *static_cast<bool *>(ref.ptr)); break;
    case FieldType::Duration: ok = parse_duration(value, *static_cast<double *>(ref.ptr)); break;
    case FieldType::StopCrit: ok = parse_stop_crit(value, *static_cast<PANOCStopCrit *>(ref.ptr)); break;
    }
    if (!ok) {
        err = "Invalid value '" + std::string(value) + "' for '" + std::string(key) + "'";
        return SetResult::BadValue;
    }
    return SetResult::Ok;
}

/// Read a field back as a double (enums/bools/ints converted); false if unknown.
inline bool get_option(const SolverOptions &o, std::string_view key, double &out) {
    if (key == "method") {
        out = static_cast<int>(o.method);
        return true;
    }
    if (key == "speculation") {
        out = o.speculation;
        return true;
    }
    if (key == "residency") {
        out = o.residency;
        return true;
    }
    if (key == "launch_order") {
        out = o.launch_order;
        return true;
    }
    if (key == "spec_patience") {
        out = o.spec_patience;
        return true;
    }
    FieldRef ref{};
    bool known = false;
    if (!find_field(const_cast<SolverOptions &>(o), key, ref, known))
        return false;
    switch (ref.type) {
    case FieldType::Double:
    case FieldType::Duration: out = *static_cast<double *>(ref.ptr); break;
    case FieldType::Int: out = *static_cast<int32_t *>(ref.ptr); break;
    case FieldType::Bool: out = *static_cast<bool *>(ref.ptr) ? 1 : 0; break;
    case FieldType::StopCrit: out = static_cast<int>(*static_cast<PANOCStopCrit *>(ref.ptr)); break;
    }
    return true;
}

/// Every option in the canonical spelling set_option() accepts (one entry per field).
inline constexpr const char *kOptionKeys[] = {
    "solver.Lipschitz.L_0", "solver.Lipschitz.epsilon", "solver.Lipschitz.delta", "solver.Lipschitz.Lgamma_factor",
    "solver.max_iter", "solver.max_time", "solver.L_min", "solver.L_max", "solver.stop_crit",
    "solver.max_no_progress", "solver.print_interval", "solver.print_precision",
    "solver.quadratic_upperbound_tolerance_factor", "solver.TR_tolerance_factor",
    "solver.ratio_threshold_acceptable", "solver.ratio_threshold_good", "solver.radius_factor_rejected",
    "solver.radius_factor_acceptable", "solver.radius_factor_good", "solver.initial_radius", "solver.min_radius",
    "solver.compute_ratio_using_new_stepsize", "solver.update_direction_on_prox_step",
    "solver.recompute_last_prox_step_after_direction_reset", "solver.disable_acceleration",
    "solver.ratio_approx_fbe_quadratic_model", "solver.min_linesearch_coefficient", "solver.force_linesearch",
    "solver.linesearch_strictness_factor", "solver.linesearch_tolerance_factor",
    "solver.update_direction_in_candidate", "solver.recompute_last_prox_step_after_stepsize_change",
    "dir.hessian_vec_factor", "dir.finite_diff", "dir.finite_diff_stepsize", "dir.rescale_on_step_size_changes",
    "dir.hessian_vec", "accel.tol_scale", "accel.tol_scale_root", "accel.tol_max", "accel.max_iter_factor",
    "accel.memory", "accel.min_div_fac", "accel.min_abs_s", "accel.cbfgs.alpha", "accel.cbfgs.epsilon",
    "accel.force_pos_def", "alm.tolerance", "alm.dual_tolerance", "alm.penalty_update_factor",
    "alm.initial_penalty", "alm.initial_penalty_factor", "alm.initial_tolerance", "alm.tolerance_update_factor",
    "alm.rel_penalty_increase_threshold", "alm.max_multiplier", "alm.max_penalty", "alm.min_penalty",
    "alm.max_iter", "alm.max_time", "alm.print_interval", "alm.print_precision", "alm.single_penalty_factor",
};

/// "key=value" for one field, in the grammar set_option() parses back to the same value.
inline std::string format_option(const SolverOptions &o, std::string_view key) {
    FieldRef ref{};
    bool known = false;
    if (!find_field(const_cast<SolverOptions &>(o), key, ref, known))
        return {};
    std::string out = std::string(key) + "=";
    char buf[64];
    switch (ref.type) {
    case FieldType::Double:
    case FieldType::Duration:
        std::snprintf(buf, sizeof buf, "%.17g", *static_cast<double *>(ref.ptr));
        return out + buf;
    case FieldType::Int: return out + std::to_string(*static_cast<int32_t *>(ref.ptr));
    case FieldType::Bool: return out + (*static_cast<bool *>(ref.ptr) ? "true" : "false");
    case FieldType::StopCrit: return out + enum_name(*static_cast<PANOCStopCrit *>(ref.ptr));
    }
    return {};
}

} // namespace pantr::host
