// C ABI (include/pantr_mi355x.h) over the gfx950 kernels.  Host side only: option parsing,
// device buffers, launches.  There is deliberately no CPU solve path in this library: every
// compute entry point needs a HIP device and fails with PANTR_ERR_NO_DEVICE without one.
#include "../../include/pantr_mi355x.h"
#include "../host/params.hpp"
#include "../host/problem_data.hpp"
#include "../host/casadi_dll.hpp"
#include "chain_model.hip.h"
#include "pantr_kernel.hip.h"
#include "pantr_alm.hip.h"
#include "panoc_kernel.hip.h"
#include "quad_model.hip.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <vector>

using namespace pantr;
using pantr::host::SolverOptions;

static_assert(sizeof(pantr_stats_t) == sizeof(DeviceStats), "stats layout mismatch");

namespace {

/// Elements of an n-vector per lane that the quadcopter family's PANTR kernels know at compile time (n <= 256).
constexpr int ALM_NJ = 4;
/// quadcopter: horizons up to this run the kernels that carry the tape product for short horizons (lane = (component, stage)).
constexpr int ALM_SHORT_STAGES = ScalarALProblem<QuadModelT<false>>::SHORT_STAGES;
/// The same for the hanging chain's PANTR kernels: the elements of an n-vector a lane owns, nj = ceil(3 N / 64), as a
/// compile-time constant for every horizon up to N = 128 (nj = 1..6: N <= 21, 42, 64, 85, 106, 128 - the horizon sweep of
/// BASELINE.json configs[3] runs nj = 1, 2, 3, the bench batch 3, the stress batch 6); longer horizons run the generic
/// passes (last template argument 0).
constexpr int CHAIN_NJ_MAX = 6;
template <class F>
auto with_chain_nj(int nj, F &&f) {
#ifdef PANTR_CHAIN_GENERIC_PASSES // development switch (A/B): every horizon through the generic passes
    nj = 0;
#endif
    switch (nj) {
#ifndef PANTR_DEV_FEW_KERNELS // development builds (make DEV=1: two minutes less): only the bench and the stress batch's kernels
    case 1: return f(std::integral_constant<int, 1>{});
    case 2: return f(std::integral_constant<int, 2>{});
    case 4: return f(std::integral_constant<int, 4>{});
    case 5: return f(std::integral_constant<int, 5>{});
#endif
    case 3: return f(std::integral_constant<int, 3>{});
    case 6: return f(std::integral_constant<int, 6>{});
    default: return f(std::integral_constant<int, 0>{});
    }
}

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(PANTR_ERR_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

enum class ProblemId { HangingChain, Quadcopter, RealisticQuadcopter };

} // namespace

struct pantr_params {
    SolverOptions opts = SolverOptions::driver_defaults();
};

struct pantr_batch {
    ProblemId problem = ProblemId::HangingChain;
    std::string name;
    int32_t N = 0, n = 0, m = 0, nx = 0, nu = 0, nc = 0, nc_N = 0;
    int32_t npad = 0, nj = 0, maxB = 0, device = 0;
    size_t ws_per_instance = 0; ///< doubles
    size_t lds_bytes       = 0; ///< dynamic LDS with all vectors in LDS (scalar models; the comparators on the chain)
    size_t lds_bytes_pantr = 0; ///< hanging chain, method=pantr: CG block + staging slot only (iterates in HBM)
    size_t lds_bytes_cmp   = 0; ///< scalar models, method=panoc|strucpanoc
    SolverOptions opts = SolverOptions::driver_defaults();
    // device memory
    double *d_x0 = nullptr, *d_u = nullptr, *d_v = nullptr, *d_o1 = nullptr, *d_o2 = nullptr,
           *d_s = nullptr, *d_ws = nullptr, *d_trace = nullptr;
    double *d_bounds = nullptr; ///< hanging chain: box C as two n_pad-vectors (lower, upper; padding 0), read by the PANTR kernels' element-wise passes
    double *d_lbfgs = nullptr; ///< L-BFGS rings of method=panoc/strucpanoc, allocated at the first such solve
    size_t lbfgs_doubles = 0;
    double *d_y = nullptr, *d_sig = nullptr, *d_om = nullptr, *d_uinit = nullptr; ///< m > 0 problems
    int32_t mpad = 0, mj = 0;
    bool al() const { return problem != ProblemId::HangingChain; }
    DeviceStats *d_stats = nullptr;
    const void *checked_ptrs[4] = {nullptr, nullptr, nullptr, nullptr}; ///< last buffer set pantr_batch_solve_device accepted
    int num_cus = 256; ///< compute units of `device` (hipDeviceAttributeMultiprocessorCount)
    int32_t *d_order     = nullptr;
    bool have_order      = false;
    // launch_order=previous|auto: the order the next launch uses, ranked on the device from the solve before it
    int32_t *d_order_prev = nullptr;
    double *d_cost        = nullptr;
    int32_t prev_B        = 0;     ///< batch size d_order_prev describes (0: nothing ranked yet)
    int32_t order_source  = 0;     ///< what the most recent launch used: 0 natural, 1 pantr_batch_set_launch_order, 2 the solve before it
    int32_t trace_instances = 0, trace_cap = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
    // numeric problem data: the compiled-in model's, or what create_from_files read from the sidecars
    BoxData box{};
    std::vector<double> x_init, u_guess;
    int32_t penalty_alm_split = 0;
    double tick_s = 1e-8;  ///< seconds per wall_clock64() tick of this device
    int32_t eval_reps = 1; ///< development aid: repeat count of the timed op in the eval kernels
    int32_t eval_depth = 1; ///< tests: prefetch depth of the products the chain's evaluation kernel runs (1 | 2)
};

namespace {

/// Problem name -> id ("dl:" prefixes dropped, casadi-dll-wrapper.cpp:74-76).
int parse_problem(const char *problem, ProblemId &id, std::string &name) {
    name = problem;
    while (name.rfind("dl:", 0) == 0)
        name = name.substr(3);
    if (name == "hanging_chain")
        id = ProblemId::HangingChain;
    else if (name == "quadcopter")
        id = ProblemId::Quadcopter;
    else if (name == "realistic_quadcopter")
        id = ProblemId::RealisticQuadcopter;
    else
        return fail(PANTR_ERR_OUT_OF_RANGE, "unknown problem '" + name +
                                                "' (available: hanging_chain, quadcopter, realistic_quadcopter)");
    return PANTR_OK;
}

/// Shape and compiled-in numeric data of a problem (host only).
host::ModelShape model_shape(ProblemId id, int32_t N) {
    host::ModelShape s{};
    if (id == ProblemId::HangingChain) {
        s.nx = ChainModel::nx; s.nu = ChainModel::nu; s.nc = 0; s.nc_N = 0;
    } else {
        s.nx = QuadModelT<false>::nx; s.nu = QuadModelT<false>::nu;
        s.nc = s.nc_N = QuadModelT<false>::nc;
    }
    s.n = s.nu * N;
    s.m = s.nc * N + s.nc_N; // ss.py:39-40
    return s;
}
void model_defaults(ProblemId id, const host::ModelShape &s, BoxData &box, std::vector<double> &x_init,
                    std::vector<double> &u_guess) {
    const bool al = id != ProblemId::HangingChain;
    for (int c = 0; c < 4; ++c) {
        box.u_lb[c] = c < s.nu ? (al ? QuadModelT<false>::u_lb(c) : ChainModel::u_lb(c)) : 0.0;
        box.u_ub[c] = c < s.nu ? (al ? QuadModelT<false>::u_ub(c) : ChainModel::u_ub(c)) : 0.0;
        box.d_lb[c] = c < s.nc ? QuadModelT<false>::d_lb(c) : 0.0;
        box.d_ub[c] = c < s.nc ? QuadModelT<false>::d_ub(c) : 0.0;
    }
    x_init.assign(s.nx, 0.0);
    if (al)
        QuadModelT<false>::initial_state(x_init.data()); // quadcopter.py:80-88
    else
        for (int l = 0; l < s.nx; ++l)
            x_init[l] = ChainModel::initial_state(l); // hanging_chain.cpp:121-124
    u_guess.assign(s.n, 0.0); // hanging_chain.cpp:135 / quadcopter.py:111
    for (int i = 0; i < s.n; ++i)
        u_guess[i] = al ? QuadModelT<false>::initial_guess(i % s.nu) : 0.0;
}

template <class T>
int dev_alloc(T *&p, size_t count) {
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
    return PANTR_OK;
}

/// Dynamic LDS of the chain's evaluation kernel: four vectors and the exchange row of the component gather.
size_t eval_lds_bytes(const pantr_batch &b) {
    return (4 * (size_t)b.npad + WaveProblem<ChainModel>::XS_DOUBLES) * sizeof(double);
}

/// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel, not of a handle: keep a
/// running maximum per kernel and never lower it (a later, smaller handle must not shrink the limit
/// of a live, larger one).
int raise_dynamic_lds(const void *kernel, size_t bytes) {
    static std::mutex mtx;
    static std::map<std::pair<int, const void *>, size_t> current;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mtx);
    size_t &cur = current[{dev, kernel}];
    if (bytes <= cur)
        return PANTR_OK;
    HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    cur = bytes;
    return PANTR_OK;
}

int check_supported(const SolverOptions &o) {
    // Options whose non-default value would select a code path the device solver does not have.
    if (o.method != host::Method::PANTR) {
        if (o.lbfgs.memory < 1 || o.lbfgs.memory > 32 * 1024)
            return fail(PANTR_ERR_OUT_OF_RANGE, "accel.memory must be in [1, 32768]");
        if (o.lbfgs.cbfgs_epsilon != 0)
            return fail(PANTR_ERR_INVALID_ARGUMENT, "accel.cbfgs.ε != 0 is not supported by the device solver");
        if (o.ldir.rescale_on_step_size_changes)
            return fail(PANTR_ERR_INVALID_ARGUMENT,
                        "dir.rescale_on_step_size_changes=true is not supported by the device solver");
        if (o.method == host::Method::StrucPANOC && o.ldir.hessian_vec)
            return fail(PANTR_ERR_INVALID_ARGUMENT,
                        "method=strucpanoc needs dir.hessian_vec=0 (new-benchmarks-paper/dodo.py:39): the "
                        "Hessian-vector term of the structured direction is not supported by the device solver");
        if (o.panoc.update_direction_in_candidate || o.panoc.recompute_last_prox_step_after_stepsize_change)
            return fail(PANTR_ERR_INVALID_ARGUMENT,
                        "solver.update_direction_in_candidate / recompute_last_prox_step_after_stepsize_change "
                        "are not supported by the device solver");
    }
    if (o.dir.finite_diff)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "dir.finite_diff=true is not supported by the device solver");
    if (o.solver.compute_ratio_using_new_stepsize)
        return fail(PANTR_ERR_INVALID_ARGUMENT,
                    "solver.compute_ratio_using_new_stepsize=true is not supported by the device solver");
    if (!o.solver.ratio_approx_fbe_quadratic_model)
        return fail(PANTR_ERR_INVALID_ARGUMENT,
                    "solver.ratio_approx_fbe_quadratic_model=false is not supported by the device solver");
    using host::PANOCStopCrit;
    if (o.solver.stop_crit != PANOCStopCrit::ProjGradUnitNorm && o.solver.stop_crit != PANOCStopCrit::FPRNorm)
        return fail(PANTR_ERR_INVALID_ARGUMENT,
                    std::string("solver.stop_crit=") + host::enum_name(o.solver.stop_crit) +
                        " is not supported by the device solver (ProjGradUnitNorm, FPRNorm)");
    return PANTR_OK;
}

KernelParams to_kernel_params(const SolverOptions &o, double tick_s, const BoxData &box) {
    KernelParams k{};
    k.tick_s = tick_s;
    k.solver_max_time_s = o.solver.max_time;
    k.spec_patience_s   = o.spec_patience;
    const auto &s = o.solver;
    k.L_0 = s.Lipschitz.L_0; k.lip_eps = s.Lipschitz.epsilon; k.lip_delta = s.Lipschitz.delta;
    k.Lgamma_factor = s.Lipschitz.Lgamma_factor;
    // m == 0: ALM performs one inner solve with alm.tolerance and alm.max_time
    k.max_time_s = std::fmin(s.max_time, o.alm.max_time);
    k.L_min = s.L_min; k.L_max = s.L_max;
    k.quadratic_upperbound_tolerance_factor = s.quadratic_upperbound_tolerance_factor;
    k.TR_tolerance_factor = s.TR_tolerance_factor;
    k.ratio_threshold_acceptable = s.ratio_threshold_acceptable;
    k.ratio_threshold_good = s.ratio_threshold_good;
    k.radius_factor_rejected = s.radius_factor_rejected;
    k.radius_factor_acceptable = s.radius_factor_acceptable;
    k.radius_factor_good = s.radius_factor_good;
    k.initial_radius = s.initial_radius; k.min_radius = s.min_radius;
    k.hessian_vec_factor = o.dir.hessian_vec_factor;
    k.tol_scale = o.accel.tol_scale; k.tol_scale_root = o.accel.tol_scale_root;
    k.tol_max = o.accel.tol_max; k.max_iter_factor = o.accel.max_iter_factor;
    k.tolerance = o.alm.tolerance;
    k.max_iter = s.max_iter; k.stop_crit = static_cast<int32_t>(s.stop_crit);
    k.max_no_progress = s.max_no_progress; k.disable_acceleration = s.disable_acceleration ? 1 : 0;
    k.always_overwrite_results = 1;
    k.min_linesearch_coefficient = o.panoc.min_linesearch_coefficient;
    k.linesearch_strictness_factor = o.panoc.linesearch_strictness_factor;
    k.linesearch_tolerance_factor = o.panoc.linesearch_tolerance_factor;
    k.force_linesearch = o.panoc.force_linesearch ? 1 : 0;
    k.lbfgs_min_div_fac = o.lbfgs.min_div_fac; k.lbfgs_min_abs_s = o.lbfgs.min_abs_s;
    k.lbfgs_memory = o.lbfgs.memory; k.lbfgs_force_pos_def = o.lbfgs.force_pos_def ? 1 : 0;
    k.box = box;
    return k;
}

AlmKernelParams to_alm_params(const SolverOptions &o) {
    AlmKernelParams a{};
    const auto &m = o.alm;
    a.tolerance = m.tolerance; a.dual_tolerance = m.dual_tolerance; a.max_time_s = m.max_time;
    a.initial_penalty = m.initial_penalty; a.penalty_update_factor = m.penalty_update_factor;
    a.initial_tolerance = m.initial_tolerance; a.tolerance_update_factor = m.tolerance_update_factor;
    a.rel_penalty_increase_threshold = m.rel_penalty_increase_threshold;
    a.max_multiplier = m.max_multiplier; a.max_penalty = m.max_penalty;
    a.max_iter = m.max_iter; a.single_penalty_factor = m.single_penalty_factor ? 1 : 0;
    return a;
}

/// Runs f(Model{}) for the scalar-model problem of the handle.
template <class F>
int with_quad_model(const pantr_batch *b, F &&f) {
    if (b->problem == ProblemId::RealisticQuadcopter)
        return f(QuadModelT<true>{});
    return f(QuadModelT<false>{});
}

int set_option(SolverOptions &o, const char *kv) {
    if (!kv)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null option string");
    std::string err;
    switch (host::set_option(o, kv, err)) {
    case host::SetResult::Ok:
    case host::SetResult::Ignored: return PANTR_OK;
    default: return fail(PANTR_ERR_INVALID_ARGUMENT, err);
    }
}

/// Hanging chain, method = pantr: two wavefronts per instance (pantr_solve_spec_kernel)?  `speculation=on`
/// wherever the doubled LDS fits a CU; auto: while the batch leaves SIMDs idle - at most three
/// instances per CU (their six wavefronts on its four SIMDs: the helpers sleep most of the time) and
/// all of them resident at once.
/// LDS of a workgroup of the two-wavefront kernel: dynamic (both wavefronts' vectors and exchange rows),
/// and with the static part (four landing buffers, the mailbox).
size_t spec_lds_bytes(const pantr_batch &b) { return spec_lds_doubles<ChainModel>(b.npad) * sizeof(double); }
size_t spec_lds_total(const pantr_batch &b) {
    return spec_lds_bytes(b) + 2 * SpecProblemT<ChainModel>::tape_lds_bytes() + sizeof(SpecBox);
}
/// The same question for the quadcopter family (pantr_alm_spec_kernel): dynamic LDS of the two-wavefront
/// workgroup; auto: every instance resident at once, at most one workgroup per pair of SIMDs.
size_t alm_spec_lds_bytes(const pantr_batch &b) {
    const bool realistic = b.problem == ProblemId::RealisticQuadcopter;
    return b.lds_bytes + sizeof(double) * (realistic ? ScalarALProblem<QuadModelT<true>>::helper_lds_doubles(b.N, b.npad)
                                                     : ScalarALProblem<QuadModelT<false>>::helper_lds_doubles(b.N, b.npad));
}
bool use_alm_speculation(const pantr_batch *b, int32_t B) {
    const size_t wg_lds = alm_spec_lds_bytes(*b) + sizeof(SpecBox);
    if (b->opts.method != host::Method::PANTR || b->opts.speculation == 0 || wg_lds > 160 * 1024)
        return false;
    if (b->opts.speculation == 1)
        return true;
    const size_t per_cu = std::min<size_t>(2, (160 * 1024) / wg_lds);
    return (size_t)B <= per_cu * (size_t)b->num_cus;
}
bool use_speculation(const pantr_batch *b, int32_t B) {
    if (b->al())
        return use_alm_speculation(b, B);
    const size_t wg_lds = spec_lds_total(*b);
    if (b->opts.speculation == 0 || wg_lds > 160 * 1024)
        return false;
    if (b->opts.speculation == 1)
        return true;
    const size_t per_cu = std::min<size_t>(3, (160 * 1024) / wg_lds);
    // ... and while the launch does not load the memory system: a product streams its tape stage by stage, the
    // two-wavefront kernel requests one stage ahead and the one-wavefront kernel two, which is what counts once
    // many tapes stream at once - measured at N = 120: 256 instances 303 ms against 356, 512 instances 425 against 393
    // (N = 60, 600 instances: 62 against 73).  The bound: instances x stages, per compute unit.
    const size_t stage_streams = (size_t)B * (size_t)b->N;
    return (size_t)B <= per_cu * (size_t)b->num_cus && stage_streams <= (size_t)176 * (size_t)b->num_cus;
}

/// `residency=k`: the dynamic LDS a launch asks for so that at most k workgroups fit the 160 KiB of a
/// compute unit (never less than the kernel needs; k = 0: exactly what it needs).
size_t residency_lds(const pantr_batch *b, size_t dyn_bytes, size_t static_bytes) {
    const int k = b->opts.residency;
    if (k <= 0)
        return dyn_bytes;
    const size_t cu = 160 * 1024, gran = 1280; // allocation granularity of the LDS (bytes), rounded up generously
    const size_t per_wg = cu / (size_t)(k + 1) + gran; // k + 1 of these no longer fit
    const size_t want = per_wg > static_bytes ? per_wg - static_bytes : 0;
    return std::min(std::max(dyn_bytes, want), cu - static_bytes);
}

/// Box C as device vectors (the element-wise passes read their bounds like any other operand); after every
/// change of b->box.
int upload_bounds(pantr_batch *b) {
    std::vector<double> h(2 * (size_t)b->npad, 0.0);
    for (int i = 0; i < b->n; ++i) {
        h[i]           = b->box.u_lb[i % b->nu];
        h[b->npad + i] = b->box.u_ub[i % b->nu];
    }
    HIP_TRY(hipMemcpy(b->d_bounds, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    return PANTR_OK;
}

// ---- launch order from the solve before (launch_order=previous|auto) --------------------------------------
// A launch with more workgroups than the chip holds at once ends with whatever instance started last; when that is
// a long one, the pass is its start time plus its whole solve.  Instances of an MPC batch need about the work they
// needed a time step earlier, so the time each took in the solve before this one on the handle is a usable forecast:
// two small kernels behind every solve kernel rank the instances by it, longest first, for the next launch (in the
// stream; nothing comes to the host).  The bench batch tiled eight times into one launch: 170 -> 130 ms.
__global__ void launch_cost_kernel(int B, const DeviceStats *stats, double *cost) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) {
        const double t = stats[i].elapsed_s;
        cost[i] = t == t ? t : 0.0;
    }
}
/// order[rank of i] = i, rank = how many instances cost more (ties: the lower index first): a permutation whatever
/// the costs are.
__global__ void launch_rank_kernel(int B, const double *cost, int32_t *order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B)
        return;
    const double c = cost[i];
    int rank = 0;
    for (int j = 0; j < B; ++j) {
        const double cj = cost[j];
        rank += (cj > c || (cj == c && j < i)) ? 1 : 0;
    }
    order[rank] = i;
}
/// Workgroups of `kernel` the chip holds at once (INT32_MAX if the runtime cannot say).
template <class K>
int32_t resident_workgroups(const pantr_batch *b, K kernel, int block, size_t lds) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds) != hipSuccess || per_cu <= 0) {
        (void)hipGetLastError();
        return INT32_MAX;
    }
    return per_cu * b->num_cus;
}
/// Does this launch take (and leave behind) an order ranked from the solve before it?
bool ranks_launches(const pantr_batch *b, int32_t B, int32_t resident) {
    if (b->have_order || b->opts.launch_order == 0)
        return false;
    return b->opts.launch_order == 1 || B > resident;
}
/// The order array of a launch of B instances (nullptr: by index); records where it came from.
const int32_t *launch_order_for(pantr_batch *b, int32_t B, int32_t resident) {
    if (b->have_order) {
        b->order_source = 1;
        return b->d_order;
    }
    if (ranks_launches(b, B, resident) && b->prev_B == B) {
        b->order_source = 2;
        return b->d_order_prev;
    }
    b->order_source = 0;
    return nullptr;
}
/// Behind a solve kernel: rank its instances for the next launch of the same size.
int rank_for_next_launch(pantr_batch *b, int32_t B, int32_t resident, const DeviceStats *d_stats, hipStream_t stream) {
    if (!ranks_launches(b, B, resident)) {
        b->prev_B = 0;
        return PANTR_OK;
    }
    const int block = 256, grid = (B + block - 1) / block;
    hipLaunchKernelGGL(launch_cost_kernel, dim3(grid), dim3(block), 0, stream, (int)B, d_stats, b->d_cost);
    hipLaunchKernelGGL(launch_rank_kernel, dim3(grid), dim3(block), 0, stream, (int)B, (const double *)b->d_cost, b->d_order_prev);
    HIP_TRY(hipGetLastError());
    b->prev_B = B;
    return PANTR_OK;
}

int launch_solve(pantr_batch *b, int32_t B, const double *d_x0, double *d_u, double *d_y,
                 DeviceStats *d_stats, hipStream_t stream) {
    if (int rc = check_supported(b->opts))
        return rc;
    const int method = static_cast<int>(b->opts.method);
    double *d_lbfgs  = nullptr;
    if (method != 0) { // L-BFGS rings in HBM; rho and alpha in the nine spare LDS vector slots 10..18
        const size_t mem = b->opts.lbfgs.memory;
        if (2 * mem > (size_t)9 * b->npad)
            return fail(PANTR_ERR_OUT_OF_RANGE, "accel.memory too large for the LDS of this horizon (2 memory <= 9 n_pad)");
        const size_t need = (size_t)b->maxB * 2 * mem * b->npad;
        if (need > b->lbfgs_doubles) {
            if (b->d_lbfgs)
                HIP_TRY(hipFree(b->d_lbfgs));
            b->d_lbfgs       = nullptr;
            b->lbfgs_doubles = 0;
            if (int rc = dev_alloc(b->d_lbfgs, need))
                return rc;
            b->lbfgs_doubles = need;
        }
        d_lbfgs = b->d_lbfgs;
    }
    if (b->al()) {
        if (b->opts.alm.initial_penalty <= 0)
            return fail(PANTR_ERR_INVALID_ARGUMENT,
                        "alm.initial_penalty must be > 0 (automatic penalty initialisation is not supported)");
        if (method != 0 && b->lds_bytes_cmp > 160 * 1024)
            return fail(PANTR_ERR_OUT_OF_RANGE, "horizon too long for method=panoc|strucpanoc: their vectors do not fit the 160 KiB LDS");
        AlmSolveArgs a{};
        a.B = B; a.N = b->N; a.n = b->n; a.npad = b->npad; a.nj = b->nj; a.m = b->m; a.mpad = b->mpad; a.mj = b->mj;
        a.trace_cap = b->trace_cap; a.trace_instances = b->trace_instances;
        a.x0 = d_x0; a.u = d_u; a.y = d_y; a.ws = b->d_ws; a.lbfgs = d_lbfgs; a.stats = d_stats;
        a.trace = b->trace_instances > 0 ? b->d_trace : nullptr;
        a.prm   = to_kernel_params(b->opts, b->tick_s, b->box);
        a.alm   = to_alm_params(b->opts);
        HIP_TRY(hipEventRecord(b->ev_start, stream));
        int32_t resident = INT32_MAX;
        auto go = [&](auto kernel, int block, size_t lds) {
            resident = B > b->num_cus ? resident_workgroups(b, kernel, block, lds) : INT32_MAX;
            a.order  = launch_order_for(b, B, resident);
            hipLaunchKernelGGL(kernel, dim3(B), dim3(block), lds, stream, a);
        };
        with_quad_model(b, [&](auto model) {
            using M = decltype(model);
            const bool fixed_nj = b->nj == ALM_NJ;
            // quadcopter up to 16 stages: the kernels with the product for short horizons (hess_prod_short)
            constexpr bool HAS_SHORT = M::angles_are_integrated_inputs;
            const bool short_horizon = HAS_SHORT && b->N <= ALM_SHORT_STAGES;
            if (method == 0 && use_alm_speculation(b, B)) {
                if constexpr (HAS_SHORT)
                    if (short_horizon) {
                        go(&pantr_alm_spec_kernel<M, 1, 1>, 2 * WAVE, alm_spec_lds_bytes(*b));
                        return 0;
                    }
                if (fixed_nj)
                    go(&pantr_alm_spec_kernel<M, ALM_NJ, 0>, 2 * WAVE, alm_spec_lds_bytes(*b));
                else
                    go(&pantr_alm_spec_kernel<M, 0, 0>, 2 * WAVE, alm_spec_lds_bytes(*b));
            } else if (method == 0) {
                if constexpr (HAS_SHORT)
                    if (short_horizon) {
                        go(&pantr_alm_kernel<M, 0, 1, 1>, WAVE, b->lds_bytes);
                        return 0;
                    }
                if (fixed_nj)
                    go(&pantr_alm_kernel<M, 0, ALM_NJ, 0>, WAVE, b->lds_bytes);
                else
                    go(&pantr_alm_kernel<M, 0, 0, 0>, WAVE, b->lds_bytes);
            }
            else if (method == 1)
                go(&pantr_alm_kernel<M, 1, 0, 0>, WAVE, b->lds_bytes_cmp);
            else
                go(&pantr_alm_kernel<M, 2, 0, 0>, WAVE, b->lds_bytes_cmp);
            return 0;
        });
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(b->ev_stop, stream));
        b->timed = true;
        return rank_for_next_launch(b, B, resident, d_stats, stream);
    }
    SolveArgs a{};
    a.B = B; a.N = b->N; a.n = b->n; a.npad = b->npad; a.nj = b->nj;
    a.trace_cap = b->trace_cap; a.trace_instances = b->trace_instances;
    a.x0 = d_x0; a.u = d_u; a.ws = b->d_ws; a.lbfgs = d_lbfgs; a.stats = d_stats;
    a.trace = b->trace_instances > 0 ? b->d_trace : nullptr;
    a.prm   = to_kernel_params(b->opts, b->tick_s, b->box);
    a.lbv = b->d_bounds; a.ubv = b->d_bounds + b->npad;
    if (method != 0 && b->lds_bytes + WaveProblem<ChainModel>::tape_lds_bytes() > 160 * 1024)
        return fail(PANTR_ERR_OUT_OF_RANGE, "horizon too long for method=panoc|strucpanoc: their vectors do not fit the 160 KiB LDS");
    HIP_TRY(hipEventRecord(b->ev_start, stream));
    int32_t resident = INT32_MAX;
    auto go = [&](auto kernel, int block, size_t lds) {
        resident = B > b->num_cus ? resident_workgroups(b, kernel, block, lds) : INT32_MAX;
        a.order  = launch_order_for(b, B, resident);
        hipLaunchKernelGGL(kernel, dim3(B), dim3(block), lds, stream, a);
    };
    if (method == 0 && use_speculation(b, B)) {
        const size_t lds = residency_lds(b, spec_lds_bytes(*b), 2 * SpecProblemT<ChainModel>::tape_lds_bytes() + sizeof(SpecBox));
        if (int rc = with_chain_nj(b->nj, [&](auto nj) {
                constexpr int NJ = decltype(nj)::value;
                if (b->opts.residency > 0)
                    if (int rc = raise_dynamic_lds(reinterpret_cast<const void *>(&pantr_solve_spec_kernel<ChainModel, NJ>), lds))
                        return rc;
                go(&pantr_solve_spec_kernel<ChainModel, NJ>, 2 * WAVE, lds);
                return 0;
            }))
            return rc;
    } else if (method == 0) {
        const size_t lds = residency_lds(b, b->lds_bytes_pantr, SolveProblemT<ChainModel, 0>::tape_lds_bytes());
        if (int rc = with_chain_nj(b->nj, [&](auto nj) {
                constexpr int NJ = decltype(nj)::value;
                if (b->opts.residency > 0)
                    if (int rc = raise_dynamic_lds(reinterpret_cast<const void *>(&pantr_solve_kernel<ChainModel, 0, NJ>), lds))
                        return rc;
                go(&pantr_solve_kernel<ChainModel, 0, NJ>, WAVE, lds);
                return 0;
            }))
            return rc;
    } else if (method == 1)
        go(&pantr_solve_kernel<ChainModel, 1, 0>, WAVE, b->lds_bytes);
    else
        go(&pantr_solve_kernel<ChainModel, 2, 0>, WAVE, b->lds_bytes);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(b->ev_stop, stream));
    b->timed = true;
    return rank_for_next_launch(b, B, resident, d_stats, stream);
}

int check_batch(const pantr_batch *b, int32_t B) {
    if (!b)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null batch handle");
    if (B < 1 || B > b->maxB)
        return fail(PANTR_ERR_OUT_OF_RANGE, "batch size " + std::to_string(B) + " outside [1, " +
                                                std::to_string(b->maxB) + "]");
    return PANTR_OK;
}

int run_eval(pantr_batch *b, int32_t B, int mode, double gamma, const double *x0, const double *u,
             const double *v, double *out_s, double *out_v, double *out_v2) {
    if (int rc = check_batch(b, B))
        return rc;
    if (b->al() && mode != EVAL_PROX)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "this problem has general constraints: use the *_al evaluation entry points");
    HIP_TRY(hipSetDevice(b->device));
    const size_t nb = (size_t)B * b->n * sizeof(double);
    if (b->al()) { // the prox step of a problem with general constraints: same box code, nu of that model
        HIP_TRY(hipMemcpy(b->d_u, u, nb, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(b->d_v, v, nb, hipMemcpyHostToDevice));
        AlmEvalArgs a{};
        a.B = B; a.N = b->N; a.n = b->n; a.npad = b->npad; a.nj = b->nj; a.m = 0; a.mpad = b->mpad; a.mj = 0;
        a.mode = ALM_EVAL_PROX; a.x0 = b->d_x0; a.u = b->d_u; a.v = b->d_v; a.y = b->d_y; a.Sig = b->d_sig;
        a.out_s = b->d_s; a.out_v = b->d_o1; a.out_m = b->d_om; a.ws = b->d_ws; a.reps = 1;
        a.out_v2 = b->d_o2; a.out_s4 = b->d_s; a.gamma = gamma; a.box = b->box;
        with_quad_model(b, [&](auto model) {
            using M = decltype(model);
            hipLaunchKernelGGL(pantr_alm_eval_kernel<M>, dim3(B), dim3(WAVE), b->lds_bytes, 0, a);
            return 0;
        });
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        if (out_s)
            HIP_TRY(hipMemcpy(out_s, b->d_s, (size_t)B * 4 * sizeof(double), hipMemcpyDeviceToHost));
        if (out_v)
            HIP_TRY(hipMemcpy(out_v, b->d_o1, nb, hipMemcpyDeviceToHost));
        if (out_v2)
            HIP_TRY(hipMemcpy(out_v2, b->d_o2, nb, hipMemcpyDeviceToHost));
        return PANTR_OK;
    }
    if (x0)
        HIP_TRY(hipMemcpy(b->d_x0, x0, (size_t)B * b->nx * sizeof(double), hipMemcpyHostToDevice));
    else
        HIP_TRY(hipMemset(b->d_x0, 0, (size_t)B * b->nx * sizeof(double)));
    HIP_TRY(hipMemcpy(b->d_u, u, nb, hipMemcpyHostToDevice));
    if (v)
        HIP_TRY(hipMemcpy(b->d_v, v, nb, hipMemcpyHostToDevice));
    EvalArgs a{};
    a.B = B; a.N = b->N; a.n = b->n; a.npad = b->npad; a.nj = b->nj; a.mode = mode;
    a.reps = b->eval_reps;
    a.x0 = b->d_x0; a.u = b->d_u; a.v = v ? b->d_v : nullptr;
    a.out_s = b->d_s; a.out_v = b->d_o1; a.out_v2 = b->d_o2; a.ws = b->d_ws; a.gamma = gamma;
    a.box = b->box;
    if (b->eval_depth == 3)
        hipLaunchKernelGGL((pantr_eval_kernel<ChainModel, 3>), dim3(B), dim3(WAVE), eval_lds_bytes(*b), 0, a);
    else if (b->eval_depth == 2)
        hipLaunchKernelGGL((pantr_eval_kernel<ChainModel, 2>), dim3(B), dim3(WAVE), eval_lds_bytes(*b), 0, a);
    else
        hipLaunchKernelGGL((pantr_eval_kernel<ChainModel, 1>), dim3(B), dim3(WAVE), eval_lds_bytes(*b), 0, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    std::vector<double> s((size_t)B * 4);
    HIP_TRY(hipMemcpy(s.data(), b->d_s, s.size() * sizeof(double), hipMemcpyDeviceToHost));
    if (out_s)
        std::memcpy(out_s, s.data(), s.size() * sizeof(double));
    if (out_v)
        HIP_TRY(hipMemcpy(out_v, b->d_o1, nb, hipMemcpyDeviceToHost));
    if (out_v2)
        HIP_TRY(hipMemcpy(out_v2, b->d_o2, nb, hipMemcpyDeviceToHost));
    return PANTR_OK;
}

} // namespace

// ---- achievable HBM bandwidth (the yardstick next to the 8 TB/s spec peak) --------
namespace {
using f32x4 = float __attribute__((ext_vector_type(4))); // one global_load_dwordx4 / global_store_dwordx4 per access
/// dst[i] = src[i], 16 bytes per lane per access, UNROLL independent accesses in flight per lane, grid-stride.
/// NT: streaming (non-temporal) loads and stores.
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy16_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < count; i += UNROLL * stride) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k)
            v[k] = NT ? __builtin_nontemporal_load(&src[i + k * stride]) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            if (NT)
                __builtin_nontemporal_store(v[k], &dst[i + k * stride]);
            else
                dst[i + k * stride] = v[k];
        }
    }
    for (; i < count; i += stride)
        dst[i] = src[i];
}
} // namespace

extern "C" {

const char *pantr_last_error(void) { return g_last_error.c_str(); }
const char *pantr_version(void) {
    // any build switch that changes what the library computes or reports is named here
#define PANTR_STR2(x) #x
#define PANTR_STR(x) PANTR_STR2(x)
#if defined(PANTR_PHASE_TIMING) && defined(PANTR_REST_TIMING) && defined(PANTR_REST_SPLIT)
    return "pantr-mi355x 0.5 (gfx950) +phase-timing +rest-timing +rest-split=" PANTR_STR(PANTR_REST_SPLIT);
#elif defined(PANTR_PHASE_TIMING) && defined(PANTR_REST_TIMING)
    return "pantr-mi355x 0.5 (gfx950) +phase-timing +rest-timing";
#elif defined(PANTR_PHASE_TIMING)
    return "pantr-mi355x 0.5 (gfx950) +phase-timing";
#elif defined(PANTR_DEV_FEW_KERNELS) || defined(PANTR_VMCNT_CONSERVATIVE) || defined(PANTR_CHAIN_GENERIC_PASSES)
    return "pantr-mi355x 0.5 (gfx950) +development-switches"; // (same results; other kernels or waits than the product build)
#else
    return "pantr-mi355x 0.5 (gfx950)";
#endif
}
const char *pantr_source_hash(void) {
#ifdef PANTR_SOURCE_HASH
    return PANTR_SOURCE_HASH;
#else
    return "";
#endif
}
size_t pantr_stats_size(void) { return sizeof(pantr_stats_t); }
int pantr_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess)
        return 0;
    return c;
}

int pantr_device_copy_bandwidth(int32_t device, int64_t bytes, int32_t reps, double *gb_per_s) {
    if (!gb_per_s || bytes < (1 << 20) || reps < 1)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "need an output pointer, bytes >= 1 MiB and reps >= 1");
    HIP_TRY(hipSetDevice(device));
    const size_t count = (size_t)bytes / sizeof(f32x4);
    f32x4 *src = nullptr, *dst = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&] {
        if (src) (void)hipFree(src);
        if (dst) (void)hipFree(dst);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&src), count * sizeof(f32x4));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dst), count * sizeof(f32x4));
    if (e == hipSuccess) e = hipMemset(src, 1, count * sizeof(f32x4));
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    // the best of a few launch shapes (which one wins moves with the device's clocks and memory state):
    // one access per thread on a grid that covers the buffer; grid-stride with 4 accesses in flight on 8 and
    // 16 workgroups per compute unit, default and non-temporal policy
    float best_ms = 0;
    for (int shape = 0; e == hipSuccess && shape < 5; ++shape) {
        for (int r = 0; e == hipSuccess && r < reps + 1; ++r) { // first pass untimed (page faults, clocks)
            (void)hipEventRecord(e0, nullptr);
            const unsigned full = (unsigned)((count + 255) / 256);
            switch (shape) {
            case 0: hipLaunchKernelGGL((copy16_kernel<1, false>), dim3(full), dim3(256), 0, nullptr, src, dst, count); break;
            case 1: hipLaunchKernelGGL((copy16_kernel<4, false>), dim3(cus * 8), dim3(256), 0, nullptr, src, dst, count); break;
            case 2: hipLaunchKernelGGL((copy16_kernel<4, true>), dim3(cus * 8), dim3(256), 0, nullptr, src, dst, count); break;
            case 3: hipLaunchKernelGGL((copy16_kernel<4, false>), dim3(cus * 16), dim3(256), 0, nullptr, src, dst, count); break;
            default: hipLaunchKernelGGL((copy16_kernel<1, true>), dim3(full), dim3(256), 0, nullptr, src, dst, count); break;
            }
            (void)hipEventRecord(e1, nullptr);
            e = hipEventSynchronize(e1);
            float ms = 0;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (r > 0 && (best_ms == 0 || ms < best_ms))
                best_ms = ms;
        }
    }
    cleanup();
    if (e != hipSuccess)
        return fail(PANTR_ERR_NO_DEVICE, std::string("copy bandwidth measurement: ") + hipGetErrorString(e));
    *gb_per_s = 2.0 * (double)(count * sizeof(f32x4)) / (best_ms * 1e-3) / 1e9; // read + write
    return PANTR_OK;
}

// ---- options ---------------------------------------------------------------------------
int pantr_params_create(pantr_params_t **out) {
    if (!out)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null output pointer");
    *out = new pantr_params;
    return PANTR_OK;
}
void pantr_params_destroy(pantr_params_t *p) { delete p; }
int pantr_params_set(pantr_params_t *p, const char *kv) {
    if (!p)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null params handle");
    return set_option(p->opts, kv);
}
int pantr_params_get(const pantr_params_t *p, const char *key, double *value) {
    if (!p || !key || !value)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    if (!host::get_option(p->opts, key, *value))
        return fail(PANTR_ERR_OUT_OF_RANGE, std::string("Invalid key '") + key + "'");
    return PANTR_OK;
}

// ---- problem ---------------------------------------------------------------------------
int pantr_batch_create(const char *problem, int32_t horizon, int32_t max_instances, int32_t device,
                       pantr_batch_t **out) {
    if (!problem || !out)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    std::string name;
    ProblemId id;
    if (int rc = parse_problem(problem, id, name))
        return rc;
    if (horizon < 1)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "horizon must be >= 1");
    if (max_instances < 1)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "max_instances must be >= 1");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1)
        return fail(PANTR_ERR_NO_DEVICE, "no HIP device available: the PANTR solver has no CPU path");
    if (device < 0 || device >= count)
        return fail(PANTR_ERR_OUT_OF_RANGE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));

    // one cleanup path: every early return below frees what was allocated so far
    std::unique_ptr<pantr_batch, void (*)(pantr_batch_t *)> b(new pantr_batch, pantr_batch_destroy);
    {
        int khz = 0; // wall_clock64() rate
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0)
            b->tick_s = 1.0 / (1e3 * (double)khz);
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
            b->num_cus = cus;
    }
    b->problem = id; b->name = name; b->device = device; b->N = horizon;
    {
        const host::ModelShape sh = model_shape(id, horizon);
        b->nx = sh.nx; b->nu = sh.nu; b->nc = sh.nc; b->nc_N = sh.nc_N; b->n = sh.n; b->m = sh.m;
        model_defaults(id, sh, b->box, b->x_init, b->u_guess);
    }
    b->nj   = (b->n + WAVE - 1) / WAVE;
    // the quadcopter family up to n = 256: every lane owns ALM_NJ elements of every vector (padding included), a
    // compile-time constant of the PANTR kernels - their element-wise passes are straight-line code (Owned<NJ>)
    // (quadcopter up to ALM_SHORT_STAGES stages, n <= 64: one element per lane, the kernels for short horizons)
    const bool short_quad = id == ProblemId::Quadcopter && horizon <= ALM_SHORT_STAGES;
    if (id != ProblemId::HangingChain && b->nj <= ALM_NJ && !short_quad)
        b->nj = ALM_NJ;
    b->npad = b->nj * WAVE;
    b->mj   = (b->m + WAVE - 1) / WAVE;
    b->mpad = b->mj * WAVE;
    if (b->nj > 32)
        return fail(PANTR_ERR_OUT_OF_RANGE, "horizon too long: n = nu*N must be <= 2048");
    b->maxB = max_instances;
    size_t static_lds = 0;
    if (b->al()) {
        // trajectories and multipliers live in LDS; the second-order tape streams from HBM
        b->ws_per_instance = 2 * ScalarALProblem<QuadModelT<false>>::workspace_doubles(b->N); // two tape buffers (speculation)
        auto lds_of = [&](bool dedicated_scratch) {
            return (id == ProblemId::RealisticQuadcopter
                        ? ScalarALProblem<QuadModelT<true>>::lds_doubles(b->N, b->npad, b->mpad, dedicated_scratch)
                        : ScalarALProblem<QuadModelT<false>>::lds_doubles(b->N, b->npad, b->mpad, dedicated_scratch)) *
                   sizeof(double);
        };
        if ((size_t)5 * b->npad < ScalarALProblem<QuadModelT<false>>::scratch_doubles(b->N))
            return fail(PANTR_ERR_OUT_OF_RANGE, "internal: the sweeps' scratch does not fit the five vector slots lent to it");
        b->lds_bytes     = lds_of(false); // method=pantr and the evaluation kernel
        b->lds_bytes_cmp = lds_of(true);  // the comparators: the cascade sweeps' scratch gets LDS of its own
    } else {
        b->ws_per_instance = WaveProblem<ChainModel>::workspace_doubles(b->N);
        const size_t xs_bytes = WaveProblem<ChainModel>::XS_DOUBLES * sizeof(double);
        b->lds_bytes       = (size_t)NUM_VEC * b->npad * sizeof(double) + xs_bytes; // dynamic part: all vectors (comparators)
        b->lds_bytes_pantr = (size_t)NUM_CG_VEC * b->npad * sizeof(double) + xs_bytes;
        static_lds         = SolveProblemT<ChainModel, 0>::tape_lds_bytes();
    }
    if ((b->al() ? b->lds_bytes : b->lds_bytes_pantr) + static_lds > 160 * 1024)
        return fail(PANTR_ERR_OUT_OF_RANGE, "decision vectors do not fit the 160 KiB LDS");
    if (b->al()) {
        for (const void *k : {(const void *)&pantr_alm_kernel<QuadModelT<false>, 0, 0, 0>,
                              (const void *)&pantr_alm_kernel<QuadModelT<true>, 0, 0, 0>,
                              (const void *)&pantr_alm_kernel<QuadModelT<false>, 0, ALM_NJ, 0>,
                              (const void *)&pantr_alm_kernel<QuadModelT<true>, 0, ALM_NJ, 0>,
                              (const void *)&pantr_alm_kernel<QuadModelT<false>, 0, 1, 1>,
                              (const void *)&pantr_alm_eval_kernel<QuadModelT<false>>,
                              (const void *)&pantr_alm_eval_kernel<QuadModelT<true>>})
            if (int rc = raise_dynamic_lds(k, b->lds_bytes))
                return rc;
        if (alm_spec_lds_bytes(*b) + sizeof(SpecBox) <= 160 * 1024)
            for (const void *k : {(const void *)&pantr_alm_spec_kernel<QuadModelT<false>, 0, 0>,
                                  (const void *)&pantr_alm_spec_kernel<QuadModelT<true>, 0, 0>,
                                  (const void *)&pantr_alm_spec_kernel<QuadModelT<false>, ALM_NJ, 0>,
                                  (const void *)&pantr_alm_spec_kernel<QuadModelT<true>, ALM_NJ, 0>,
                                  (const void *)&pantr_alm_spec_kernel<QuadModelT<false>, 1, 1>})
                if (int rc = raise_dynamic_lds(k, alm_spec_lds_bytes(*b)))
                    return rc;
        if (b->lds_bytes_cmp <= 160 * 1024) // else the comparators are refused at launch
            for (const void *k : {(const void *)&pantr_alm_kernel<QuadModelT<false>, 1>,
                                  (const void *)&pantr_alm_kernel<QuadModelT<false>, 2>,
                                  (const void *)&pantr_alm_kernel<QuadModelT<true>, 1>,
                                  (const void *)&pantr_alm_kernel<QuadModelT<true>, 2>})
                if (int rc = raise_dynamic_lds(k, b->lds_bytes_cmp))
                    return rc;
    } else {
        if (int rc = with_chain_nj(b->nj, [&](auto nj) {
                constexpr int NJ = decltype(nj)::value;
                if (int rc = raise_dynamic_lds((const void *)&pantr_solve_kernel<ChainModel, 0, NJ>, b->lds_bytes_pantr))
                    return rc;
                if (spec_lds_total(*b) <= 160 * 1024)
                    if (int rc = raise_dynamic_lds((const void *)&pantr_solve_spec_kernel<ChainModel, NJ>, spec_lds_bytes(*b)))
                        return rc;
                return 0;
            }))
            return rc;
        if (b->lds_bytes + static_lds <= 160 * 1024) // else the comparators are refused at launch
            for (const void *k : {(const void *)&pantr_solve_kernel<ChainModel, 1, 0>,
                                  (const void *)&pantr_solve_kernel<ChainModel, 2, 0>})
                if (int rc = raise_dynamic_lds(k, b->lds_bytes))
                    return rc;
        if (int rc = raise_dynamic_lds((const void *)&pantr_eval_kernel<ChainModel, 2>, eval_lds_bytes(*b)))
            return rc;
        if (int rc = raise_dynamic_lds((const void *)&pantr_eval_kernel<ChainModel, 1>, eval_lds_bytes(*b)))
            return rc;
    }
    const size_t B  = b->maxB;
    const size_t nv = std::max(b->n, b->nx); // the vector buffers also carry nx-sized rows
    const size_t mv = std::max<size_t>(b->m, 1);
    int rc;
    if ((rc = dev_alloc(b->d_x0, B * b->nx)) || (rc = dev_alloc(b->d_u, B * nv)) ||
        (rc = dev_alloc(b->d_v, B * nv)) || (rc = dev_alloc(b->d_o1, B * nv)) ||
        (rc = dev_alloc(b->d_o2, B * nv)) || (rc = dev_alloc(b->d_s, B * 4)) ||
        (rc = dev_alloc(b->d_ws, std::max<size_t>(B * b->ws_per_instance, 1))) || (rc = dev_alloc(b->d_stats, B)) ||
        (rc = dev_alloc(b->d_order, B)) || (rc = dev_alloc(b->d_order_prev, B)) || (rc = dev_alloc(b->d_cost, B)) || (rc = dev_alloc(b->d_y, B * mv)) || (rc = dev_alloc(b->d_sig, B * mv)) ||
        (rc = dev_alloc(b->d_om, B * mv)) || (rc = dev_alloc(b->d_uinit, nv)) ||
        (rc = dev_alloc(b->d_bounds, 2 * (size_t)b->npad)))
        return rc;
    if ((rc = upload_bounds(b.get())))
        return rc;
    HIP_TRY(hipMemcpy(b->d_uinit, b->u_guess.data(), (size_t)b->n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(b->d_ws, 0, std::max<size_t>(B * b->ws_per_instance, 1) * sizeof(double)));
    HIP_TRY(hipEventCreate(&b->ev_start));
    HIP_TRY(hipEventCreate(&b->ev_stop));
    *out = b.release();
    return PANTR_OK;
}

void pantr_batch_destroy(pantr_batch_t *b) {
    if (!b)
        return;
    (void)hipSetDevice(b->device);
    for (void *p : {(void *)b->d_x0, (void *)b->d_u, (void *)b->d_v, (void *)b->d_o1, (void *)b->d_o2,
                    (void *)b->d_s, (void *)b->d_ws, (void *)b->d_trace, (void *)b->d_stats,
                    (void *)b->d_order, (void *)b->d_order_prev, (void *)b->d_cost, (void *)b->d_y, (void *)b->d_sig, (void *)b->d_om, (void *)b->d_uinit,
                    (void *)b->d_lbfgs, (void *)b->d_bounds})
        if (p)
            (void)hipFree(p);
    if (b->ev_start)
        (void)hipEventDestroy(b->ev_start);
    if (b->ev_stop)
        (void)hipEventDestroy(b->ev_stop);
    delete b;
}

int pantr_batch_dims(const pantr_batch_t *b, int32_t *n, int32_t *m, int32_t *nx, int32_t *nu,
                     int32_t *nc, int32_t *nc_N) {
    if (!b)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null batch handle");
    if (n) *n = b->n;
    if (m) *m = b->m;
    if (nx) *nx = b->nx;
    if (nu) *nu = b->nu;
    if (nc) *nc = b->nc;
    if (nc_N) *nc_N = b->nc_N;
    return PANTR_OK;
}

int pantr_batch_get_box_C(const pantr_batch_t *b, double *lb, double *ub) {
    if (!b || !lb || !ub)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    for (int i = 0; i < b->n; ++i) { // formulation.hpp:220-227: U tiled N times
        lb[i] = b->box.u_lb[i % b->nu];
        ub[i] = b->box.u_ub[i % b->nu];
    }
    return PANTR_OK;
}

int pantr_batch_get_box_D(const pantr_batch_t *b, double *lb, double *ub) {
    if (!b || (b->m > 0 && (!lb || !ub)))
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    for (int j = 0; j < b->m; ++j) { // formulation.hpp:229-238: D tiled N times, then D_N
        lb[j] = b->box.d_lb[j % b->nc];
        ub[j] = b->box.d_ub[j % b->nc];
    }
    return PANTR_OK;
}

int pantr_batch_get_initial_guess(const pantr_batch_t *b, double *u) {
    if (!b || !u)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    std::copy(b->u_guess.begin(), b->u_guess.end(), u);
    return PANTR_OK;
}

int pantr_batch_get_initial_state(const pantr_batch_t *b, double *x0) {
    if (!b || !x0)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    std::copy(b->x_init.begin(), b->x_init.end(), x0);
    return PANTR_OK;
}

int pantr_batch_get_penalty_alm_split(const pantr_batch_t *b, int32_t *split) {
    if (!b || !split)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    *split = b->penalty_alm_split;
    return PANTR_OK;
}

// ---- compiled-in problem data and the sidecar files (host only: no device needed) ----------------
int pantr_problem_dims(const char *problem, int32_t horizon, int32_t *n, int32_t *m, int32_t *nx, int32_t *nu,
                       int32_t *nc, int32_t *nc_N) {
    if (!problem)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    std::string name;
    ProblemId id;
    if (int rc = parse_problem(problem, id, name))
        return rc;
    if (horizon < 1)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "horizon must be >= 1");
    const host::ModelShape s = model_shape(id, horizon);
    if (n) *n = s.n;
    if (m) *m = s.m;
    if (nx) *nx = s.nx;
    if (nu) *nu = s.nu;
    if (nc) *nc = s.nc;
    if (nc_N) *nc_N = s.nc_N;
    return PANTR_OK;
}

int pantr_problem_defaults(const char *problem, int32_t horizon, double *C_lb, double *C_ub, double *D_lb,
                           double *D_ub, double *x_init, double *u_guess) {
    if (!problem)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    std::string name;
    ProblemId id;
    if (int rc = parse_problem(problem, id, name))
        return rc;
    if (horizon < 1)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "horizon must be >= 1");
    const host::ModelShape s = model_shape(id, horizon);
    BoxData box{};
    std::vector<double> xi, ug;
    model_defaults(id, s, box, xi, ug);
    for (int i = 0; i < s.n; ++i) {
        if (C_lb) C_lb[i] = box.u_lb[i % s.nu];
        if (C_ub) C_ub[i] = box.u_ub[i % s.nu];
    }
    for (int j = 0; j < s.m; ++j) {
        if (D_lb) D_lb[j] = box.d_lb[j % s.nc];
        if (D_ub) D_ub[j] = box.d_ub[j % s.nc];
    }
    if (x_init) std::copy(xi.begin(), xi.end(), x_init);
    if (u_guess) std::copy(ug.begin(), ug.end(), u_guess);
    return PANTR_OK;
}

struct pantr_problem_data {
    host::ProblemData d;
    host::ModelShape shape{};
    ProblemId id = ProblemId::HangingChain;
};

int pantr_problem_data_load(const char *path, const char *formulation, int32_t horizon,
                            pantr_problem_data_t **out) {
    if (!path || !formulation || !out)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    const std::string form = formulation;
    if (form != "ss" && form != "ss2" && form != "ss2p") // casadi-dll-wrapper.cpp:29-32
        return fail(PANTR_ERR_INVALID_ARGUMENT,
                    "formulation '" + form + "' is not available (single shooting only: ss, ss2, ss2p)");
    if (horizon < 1)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "horizon must be >= 1");
    auto pd = std::make_unique<pantr_problem_data>();
    std::string err;
    switch (host::load_problem_data(path, form, horizon, pd->d, err)) {
    case host::LoadStatus::Ok: break;
    default: return fail(PANTR_ERR_RUNTIME, err); // std::runtime_error in the reference (:58-59)
    }
    std::string name;
    if (int rc = parse_problem(pd->d.name.c_str(), pd->id, name))
        return rc;
    pd->shape = model_shape(pd->id, horizon);
    if (!host::check_problem_data(pd->d, pd->shape, err))
        return fail(PANTR_ERR_INVALID_ARGUMENT, err); // std::logic_error-like: wrong dimensions
    *out = pd.release();
    return PANTR_OK;
}

void pantr_problem_data_destroy(pantr_problem_data_t *d) { delete d; }

int pantr_problem_data_get(const pantr_problem_data_t *pd, int32_t *has_csv, double *C_lb, double *C_ub,
                           double *D_lb, double *D_ub, double *param, double *guess, int32_t *penalty_alm_split) {
    if (!pd)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    // rows the files do not give fall back to the compiled-in model's values
    BoxData box{};
    std::vector<double> xi, ug;
    model_defaults(pd->id, pd->shape, box, xi, ug);
    const auto &d = pd->d;
    const auto &s = pd->shape;
    if (has_csv) *has_csv = d.have_csv ? 1 : 0;
    for (int i = 0; i < s.n; ++i) {
        if (C_lb) C_lb[i] = d.C_lb.empty() ? box.u_lb[i % s.nu] : d.C_lb[i];
        if (C_ub) C_ub[i] = d.C_ub.empty() ? box.u_ub[i % s.nu] : d.C_ub[i];
        if (guess) guess[i] = d.guess[i];
    }
    for (int j = 0; j < s.m; ++j) {
        if (D_lb) D_lb[j] = d.D_lb.empty() ? box.d_lb[j % s.nc] : d.D_lb[j];
        if (D_ub) D_ub[j] = d.D_ub.empty() ? box.d_ub[j % s.nc] : d.D_ub[j];
    }
    for (int l = 0; l < s.nx; ++l)
        if (param) param[l] = d.param.empty() ? xi[l] : d.param[l];
    if (penalty_alm_split) *penalty_alm_split = d.penalty_alm_split;
    return PANTR_OK;
}

int pantr_batch_create_from_files(const char *path, const char *formulation, int32_t horizon,
                                  int32_t max_instances, int32_t device, pantr_batch_t **out) {
    if (!out)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    pantr_problem_data_t *raw = nullptr;
    if (int rc = pantr_problem_data_load(path, formulation, horizon, &raw))
        return rc;
    std::unique_ptr<pantr_problem_data> pd(raw);
    pantr_batch_t *h = nullptr;
    if (int rc = pantr_batch_create(pd->d.name.c_str(), horizon, max_instances, device, &h))
        return rc;
    std::unique_ptr<pantr_batch, void (*)(pantr_batch_t *)> b(h, pantr_batch_destroy);
    const auto &d = pd->d;
    for (int c = 0; c < b->nu && !d.C_lb.empty(); ++c) {
        b->box.u_lb[c] = d.C_lb[c];
        b->box.u_ub[c] = d.C_ub[c];
    }
    for (int c = 0; c < b->nc && !d.D_lb.empty(); ++c) {
        b->box.d_lb[c] = d.D_lb[c];
        b->box.d_ub[c] = d.D_ub[c];
    }
    if (!d.param.empty())
        b->x_init = d.param; // CasADiProblem::param, the initial state (casadi-dll-wrapper.cpp:64)
    b->u_guess           = d.guess;
    b->penalty_alm_split = d.penalty_alm_split;
    HIP_TRY(hipMemcpy(b->d_uinit, b->u_guess.data(), (size_t)b->n * sizeof(double), hipMemcpyHostToDevice));
    if (int rc = upload_bounds(b.get()))
        return rc;
    *out = b.release();
    return PANTR_OK;
}

int pantr_batch_set_param(pantr_batch_t *b, const char *kv) {
    if (!b)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null batch handle");
    return set_option(b->opts, kv);
}
int pantr_batch_get_param(const pantr_batch_t *b, const char *key, double *value) {
    if (!b || !key || !value)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    if (!host::get_option(b->opts, key, *value))
        return fail(PANTR_ERR_OUT_OF_RANGE, std::string("Invalid key '") + key + "'");
    return PANTR_OK;
}

int pantr_batch_footprint(const pantr_batch_t *b, int64_t *lds_bytes, int64_t *ws_bytes_per_instance) {
    return pantr_batch_footprint_for(b, b ? b->maxB : 0, lds_bytes, ws_bytes_per_instance);
}
int pantr_batch_footprint_for(const pantr_batch_t *b, int32_t B, int64_t *lds_bytes, int64_t *ws_bytes_per_instance) {
    if (!b)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null batch handle");
    if (B < 1 || B > b->maxB)
        return fail(PANTR_ERR_OUT_OF_RANGE, "batch size outside [1, max_instances]");
    if (lds_bytes) { // of the solve kernel the handle's current method and options select for a batch of B (pantr_batch_kernel_name)
        const bool pantr_chain = !b->al() && b->opts.method == host::Method::PANTR;
        const bool cmp_scalar = b->al() && b->opts.method != host::Method::PANTR;
        if (pantr_chain && use_speculation(b, B))
            *lds_bytes = (int64_t)spec_lds_total(*b);
        else if (b->al() && use_alm_speculation(b, B))
            *lds_bytes = (int64_t)(alm_spec_lds_bytes(*b) + sizeof(SpecBox));
        else
            *lds_bytes = (int64_t)((pantr_chain ? b->lds_bytes_pantr : cmp_scalar ? b->lds_bytes_cmp : b->lds_bytes) +
                                   (b->al() ? 0 : pantr_chain ? SolveProblemT<ChainModel, 0>::tape_lds_bytes()
                                                              : SolveProblemT<ChainModel, 1>::tape_lds_bytes()));
    }
    if (ws_bytes_per_instance) *ws_bytes_per_instance = (int64_t)(b->ws_per_instance * sizeof(double));
    return PANTR_OK;
}

const char *pantr_batch_kernel_name(const pantr_batch_t *b, int32_t B) {
    if (!b)
        return "";
    const int method = static_cast<int>(b->opts.method);
    if (b->al()) {
        const bool realistic = b->problem == ProblemId::RealisticQuadcopter;
        // (template arguments behind the model and the method: the compile-time elements per lane - ALM_NJ, or 0 = generic
        // passes -, then 1 for the kernels with the product for short horizons, quadcopter up to ALM_SHORT_STAGES stages)
        static const char *names[2][3] = {
            {"pantr_alm_kernel<QuadModelT<false>, 0, 0, 0>", "pantr_alm_kernel<QuadModelT<false>, 1, 0, 0>", "pantr_alm_kernel<QuadModelT<false>, 2, 0, 0>"},
            {"pantr_alm_kernel<QuadModelT<true>, 0, 0, 0>", "pantr_alm_kernel<QuadModelT<true>, 1, 0, 0>", "pantr_alm_kernel<QuadModelT<true>, 2, 0, 0>"}};
        const bool fixed_nj = b->nj == ALM_NJ, short_horizon = !realistic && b->N <= ALM_SHORT_STAGES;
        if (use_alm_speculation(b, B)) {
            if (short_horizon)
                return "pantr_alm_spec_kernel<QuadModelT<false>, 1, 1>";
            return realistic ? (fixed_nj ? "pantr_alm_spec_kernel<QuadModelT<true>, 4, 0>" : "pantr_alm_spec_kernel<QuadModelT<true>, 0, 0>")
                             : (fixed_nj ? "pantr_alm_spec_kernel<QuadModelT<false>, 4, 0>" : "pantr_alm_spec_kernel<QuadModelT<false>, 0, 0>");
        }
        if (method == 0 && short_horizon)
            return "pantr_alm_kernel<QuadModelT<false>, 0, 1, 1>";
        if (method == 0 && fixed_nj)
            return realistic ? "pantr_alm_kernel<QuadModelT<true>, 0, 4, 0>" : "pantr_alm_kernel<QuadModelT<false>, 0, 4, 0>";
        return names[realistic ? 1 : 0][method];
    }
    // (the last template argument: the compile-time elements per lane, 1..CHAIN_NJ_MAX, or 0 = generic passes)
    if (method == 0) {
        static const char *spec[CHAIN_NJ_MAX + 1] = {
            "pantr_solve_spec_kernel<ChainModel, 0>", "pantr_solve_spec_kernel<ChainModel, 1>", "pantr_solve_spec_kernel<ChainModel, 2>",
            "pantr_solve_spec_kernel<ChainModel, 3>", "pantr_solve_spec_kernel<ChainModel, 4>", "pantr_solve_spec_kernel<ChainModel, 5>",
            "pantr_solve_spec_kernel<ChainModel, 6>"};
        static const char *solo[CHAIN_NJ_MAX + 1] = {
            "pantr_solve_kernel<ChainModel, 0, 0>", "pantr_solve_kernel<ChainModel, 0, 1>", "pantr_solve_kernel<ChainModel, 0, 2>",
            "pantr_solve_kernel<ChainModel, 0, 3>", "pantr_solve_kernel<ChainModel, 0, 4>", "pantr_solve_kernel<ChainModel, 0, 5>",
            "pantr_solve_kernel<ChainModel, 0, 6>"};
        int v = b->nj >= 1 && b->nj <= CHAIN_NJ_MAX ? b->nj : 0;
#ifdef PANTR_CHAIN_GENERIC_PASSES
        v = 0;
#endif
#ifdef PANTR_DEV_FEW_KERNELS
        v = (v == 3 || v == 6) ? v : 0;
#endif
        return use_speculation(b, B) ? spec[v] : solo[v];
    }
    return method == 1 ? "pantr_solve_kernel<ChainModel, 1, 0>" : "pantr_solve_kernel<ChainModel, 2, 0>";
}

// ---- the hot path ------------------------------------------------------------------------
int pantr_batch_solve(pantr_batch_t *b, int32_t B, const double *x0, double *u, double *y,
                      pantr_stats_t *stats) {
    if (int rc = check_batch(b, B))
        return rc;
    if (!x0 || !u || !stats)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipMemcpy(b->d_x0, x0, (size_t)B * b->nx * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_u, u, (size_t)B * b->n * sizeof(double), hipMemcpyHostToDevice));
    if (b->al()) {
        if (!y)
            return fail(PANTR_ERR_INVALID_ARGUMENT, "this problem has general constraints: y[B][m] is required");
        HIP_TRY(hipMemcpy(b->d_y, y, (size_t)B * b->m * sizeof(double), hipMemcpyHostToDevice));
    }
    if (int rc = launch_solve(b, B, b->d_x0, b->d_u, b->d_y, b->d_stats, nullptr))
        return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (b->al())
        HIP_TRY(hipMemcpy(y, b->d_y, (size_t)B * b->m * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(u, b->d_u, (size_t)B * b->n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(stats, b->d_stats, (size_t)B * sizeof(DeviceStats), hipMemcpyDeviceToHost));
    return PANTR_OK;
}

int pantr_batch_solve_device(pantr_batch_t *b, int32_t B, const double *d_x0, double *d_u, double *d_y,
                             pantr_stats_t *d_stats, void *stream) {
    if (int rc = check_batch(b, B))
        return rc;
    if (!d_x0 || !d_u || !d_stats || (b->al() && !d_y))
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(b->device));
    // The buffers come from another allocator (e.g. PyTorch).  If that allocator lives in a different
    // HIP runtime instance than the one this library is bound to (a process that loaded this
    // library before `import torch`), the runtime here does not know the pointers: fail loudly
    // instead of launching a kernel on addresses it cannot see.
    // Checked once per set of base pointers (a caller passes the same buffers launch after launch; the
    // attribute queries stay off the launch path).  Managed memory is device-accessible and accepted.
    const void *ptrs[4] = {d_x0, d_u, d_stats, d_y};
    if (std::memcmp(ptrs, b->checked_ptrs, sizeof ptrs) != 0) {
        for (const void *p : ptrs) {
            if (!p)
                continue;
            hipPointerAttribute_t attr{};
            const bool known = hipPointerGetAttributes(&attr, p) == hipSuccess;
            const bool ok = known && ((attr.type == hipMemoryTypeDevice && attr.device == b->device) ||
                                      attr.type == hipMemoryTypeManaged);
            if (!ok) {
                (void)hipGetLastError();
                return fail(PANTR_ERR_NO_DEVICE,
                            // (the set stays unaccepted: checked_ptrs is only written after all four passed)
                            "pantr_batch_solve_device: a buffer is not device (or managed) memory of this handle's GPU as "
                            "seen by the HIP runtime this library is bound to (in a process that also uses PyTorch, "
                            "import torch and initialise torch.cuda before the first call into this library)");
            }
        }
        std::memcpy(b->checked_ptrs, ptrs, sizeof ptrs);
    }
    const int rc = launch_solve(b, B, d_x0, d_u, d_y, reinterpret_cast<DeviceStats *>(d_stats),
                                static_cast<hipStream_t>(stream));
    if (rc != PANTR_OK) // whatever went wrong: the next call looks at its buffers again
        std::memset(b->checked_ptrs, 0, sizeof b->checked_ptrs);
    return rc;
}

int pantr_batch_forget_buffers(pantr_batch_t *b) {
    if (!b)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null batch handle");
    std::memset(b->checked_ptrs, 0, sizeof b->checked_ptrs);
    return PANTR_OK;
}

int pantr_batch_last_kernel_ms(pantr_batch_t *b, float *ms) {
    if (!b || !ms)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    if (!b->timed)
        return fail(PANTR_ERR_RUNTIME, "no solve has been launched on this handle");
    HIP_TRY(hipEventSynchronize(b->ev_stop));
    HIP_TRY(hipEventElapsedTime(ms, b->ev_start, b->ev_stop));
    return PANTR_OK;
}

int pantr_batch_set_launch_order(pantr_batch_t *b, int32_t B, const int32_t *order) {
    if (!b)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null batch handle");
    if (!order) {
        b->have_order = false;
        return PANTR_OK;
    }
    if (int rc = check_batch(b, B))
        return rc;
    std::vector<char> seen(B, 0);
    for (int i = 0; i < B; ++i) {
        if (order[i] < 0 || order[i] >= B || seen[order[i]])
            return fail(PANTR_ERR_INVALID_ARGUMENT, "launch order is not a permutation of 0..B-1");
        seen[order[i]] = 1;
    }
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipMemcpy(b->d_order, order, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice));
    b->have_order = true;
    return PANTR_OK;
}

int pantr_batch_last_launch_order(pantr_batch_t *b, int32_t B, int32_t *source, int32_t *next_order) {
    if (!b)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null batch handle");
    if (!b->timed)
        return fail(PANTR_ERR_RUNTIME, "no solve has been launched on this handle");
    if (source)
        *source = b->order_source;
    if (next_order) {
        if (int rc = check_batch(b, B))
            return rc;
        if (b->prev_B != B)
            return fail(PANTR_ERR_RUNTIME, "the most recent solve left no ranking for a launch of this size (launch_order, batch size)");
        HIP_TRY(hipSetDevice(b->device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(next_order, b->d_order_prev, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return PANTR_OK;
}

int pantr_batch_set_trace(pantr_batch_t *b, int32_t instances, int32_t cap) {
    if (!b)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null batch handle");
    if (instances < 0 || instances > b->maxB || cap < 0)
        return fail(PANTR_ERR_OUT_OF_RANGE, "trace size out of range");
    HIP_TRY(hipSetDevice(b->device));
    if (b->d_trace) {
        HIP_TRY(hipFree(b->d_trace));
        b->d_trace = nullptr;
    }
    b->trace_instances = instances;
    b->trace_cap       = cap;
    if (instances > 0 && cap > 0) {
        size_t cnt = (size_t)instances * cap * TRACE_COLS;
        if (int rc = dev_alloc(b->d_trace, cnt))
            return rc;
        HIP_TRY(hipMemset(b->d_trace, 0, cnt * sizeof(double)));
    } else {
        b->trace_instances = b->trace_cap = 0;
    }
    return PANTR_OK;
}

int pantr_batch_get_trace(pantr_batch_t *b, double *out) {
    if (!b || !out)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    if (!b->d_trace)
        return fail(PANTR_ERR_RUNTIME, "tracing is not enabled on this handle");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipMemcpy(out, b->d_trace, (size_t)b->trace_instances * b->trace_cap * TRACE_COLS * sizeof(double),
                      hipMemcpyDeviceToHost));
    return PANTR_OK;
}

// ---- problem functions -------------------------------------------------------------------
int pantr_batch_eval_psi(pantr_batch_t *b, int32_t B, const double *x0, const double *u, double *psi) {
    if (!x0 || !u || !psi)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    std::vector<double> s((size_t)(B > 0 ? B : 0) * 4);
    if (int rc = run_eval(b, B, EVAL_PSI, 0, x0, u, nullptr, s.data(), nullptr, nullptr))
        return rc;
    for (int i = 0; i < B; ++i)
        psi[i] = s[(size_t)i * 4];
    return PANTR_OK;
}

int pantr_batch_eval_psi_grad_psi(pantr_batch_t *b, int32_t B, const double *x0, const double *u,
                                  double *psi, double *grad) {
    if (!x0 || !u || !psi || !grad)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    std::vector<double> s((size_t)(B > 0 ? B : 0) * 4);
    if (int rc = run_eval(b, B, EVAL_PSI_GRAD, 0, x0, u, nullptr, s.data(), grad, nullptr))
        return rc;
    for (int i = 0; i < B; ++i)
        psi[i] = s[(size_t)i * 4];
    return PANTR_OK;
}

int pantr_batch_eval_hess_psi_prod(pantr_batch_t *b, int32_t B, const double *x0, const double *u,
                                   const double *v, double *Hv) {
    if (!x0 || !u || !v || !Hv)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    // the solver's own path: record the tape at u, then replay it for v
    return run_eval(b, B, EVAL_HESS_PROD_TAPE, 0, x0, u, v, nullptr, Hv, nullptr);
}

/// Development aid: same as above, also returning {psi, shader cycles, 100 MHz ticks, 0} per instance.
extern "C" int pantr_debug_hess_prod_timed(pantr_batch_t *b, int32_t B, const double *x0, const double *u,
                                           const double *v, double *Hv, double *scal) {
    return run_eval(b, B, EVAL_HESS_PROD_TAPE, 0, x0, u, v, scal, Hv, nullptr);
}

/// Development aid (profiling build): (psi, grad psi) returning {psi, cycles of `reps` forward sweeps, of
/// `reps` adjoint sweeps, of `reps` recording adjoint sweeps} per instance.
extern "C" int pantr_debug_psi_grad_timed(pantr_batch_t *b, int32_t B, const double *x0, const double *u,
                                          double *grad, double *scal) {
    return run_eval(b, B, EVAL_PSI_GRAD, 0, x0, u, nullptr, scal, grad, nullptr);
}

/// Development aid / tests: `count` doubles of instance `inst`'s HBM workspace from `offset` (layout:
/// WaveProblem::bind_workspace - the trajectory rows of the last stored forward sweep come first).
extern "C" int pantr_debug_read_workspace(pantr_batch_t *b, int32_t inst, int64_t offset, int64_t count, double *out) {
    if (!b || !out || inst < 0 || inst >= b->maxB || offset < 0 || count < 0 ||
        (size_t)(offset + count) > b->ws_per_instance)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "pantr_debug_read_workspace: bad range");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipMemcpy(out, b->d_ws + (size_t)inst * b->ws_per_instance + offset, (size_t)count * sizeof(double),
                      hipMemcpyDeviceToHost));
    return PANTR_OK;
}

/// Tests / development: the prefetch depth (1 | 2) of the products the chain's evaluation kernel runs.
extern "C" void pantr_debug_set_eval_prefetch_depth(pantr_batch_t *b, int depth) {
    if (b)
        b->eval_depth = depth == 2 || depth == 3 ? depth : 1;
}

extern "C" void pantr_debug_set_eval_reps(pantr_batch_t *b, int r) {
    if (b)
        b->eval_reps = r > 0 ? r : 1;
}

int pantr_batch_eval_prox_grad_step(pantr_batch_t *b, int32_t B, double gamma, const double *x,
                                    const double *grad, double *xhat, double *p, double *scalars) {
    if (!x || !grad || !xhat || !p || !scalars)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    return run_eval(b, B, EVAL_PROX, gamma, nullptr, x, grad, scalars, xhat, p);
}

int pantr_batch_simulate_dynamics(pantr_batch_t *b, int32_t B, const double *x, const double *u,
                                  double *x_next) {
    if (int rc = check_batch(b, B))
        return rc;
    if (!x || !u || !x_next)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipMemcpy(b->d_x0, x, (size_t)B * b->nx * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_v, u, (size_t)B * b->nu * sizeof(double), hipMemcpyHostToDevice));
    if (b->al())
        with_quad_model(b, [&](auto model) {
            using M = decltype(model);
            hipLaunchKernelGGL(pantr_scalar_simulate_kernel<M>, dim3((B + 63) / 64), dim3(64), 0, 0, B, b->d_x0, b->d_v,
                               b->nu, b->d_o1);
            return 0;
        });
    else
        hipLaunchKernelGGL(pantr_simulate_kernel<ChainModel>, dim3(B), dim3(WAVE), 0, 0, B, b->d_x0, b->d_v,
                           b->nu, b->d_o1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(x_next, b->d_o1, (size_t)B * b->nx * sizeof(double), hipMemcpyDeviceToHost));
    return PANTR_OK;
}

// ---- problem functions with multipliers (m > 0) --------------------------------------------------
static int run_al_eval(pantr_batch *b, int32_t B, int mode, const double *x0, const double *u, const double *y,
                       const double *Sigma, const double *v, double *psi, double *out_v, double *yhat) {
    if (int rc = check_batch(b, B))
        return rc;
    if (!b->al())
        return fail(PANTR_ERR_INVALID_ARGUMENT, "this problem has no general constraints: use the plain evaluation entry points");
    if (!x0 || !u || !y || !Sigma || !out_v)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(b->device));
    const size_t nb = (size_t)B * b->n * sizeof(double), mb = (size_t)B * b->m * sizeof(double);
    HIP_TRY(hipMemcpy(b->d_x0, x0, (size_t)B * b->nx * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_u, u, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_y, y, mb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_sig, Sigma, mb, hipMemcpyHostToDevice));
    if (v)
        HIP_TRY(hipMemcpy(b->d_v, v, nb, hipMemcpyHostToDevice));
    AlmEvalArgs a{};
    a.B = B; a.N = b->N; a.n = b->n; a.npad = b->npad; a.nj = b->nj; a.m = b->m; a.mpad = b->mpad; a.mj = b->mj;
    a.mode = mode; a.x0 = b->d_x0; a.u = b->d_u; a.v = v ? b->d_v : nullptr; a.y = b->d_y; a.Sig = b->d_sig;
    a.out_s = b->d_s; a.out_v = b->d_o1; a.out_m = b->d_om; a.ws = b->d_ws; a.reps = b->eval_reps;
    a.box = b->box;
    with_quad_model(b, [&](auto model) {
        using M = decltype(model);
        hipLaunchKernelGGL(pantr_alm_eval_kernel<M>, dim3(B), dim3(WAVE), b->lds_bytes, 0, a);
        return 0;
    });
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (psi)
        HIP_TRY(hipMemcpy(psi, b->d_s, (size_t)B * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_v, b->d_o1, nb, hipMemcpyDeviceToHost));
    if (yhat)
        HIP_TRY(hipMemcpy(yhat, b->d_om, mb, hipMemcpyDeviceToHost));
    return PANTR_OK;
}

int pantr_batch_eval_psi_grad_psi_al(pantr_batch_t *b, int32_t B, const double *x0, const double *u,
                                     const double *y, const double *Sigma, double *psi, double *grad,
                                     double *yhat) {
    return run_al_eval(b, B, ALM_EVAL_PSI_GRAD, x0, u, y, Sigma, nullptr, psi, grad, yhat);
}

int pantr_batch_eval_hess_psi_prod_al(pantr_batch_t *b, int32_t B, const double *x0, const double *u,
                                      const double *y, const double *Sigma, const double *v, double *Hv) {
    if (!v)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    return run_al_eval(b, B, ALM_EVAL_HESS_PROD, x0, u, y, Sigma, v, nullptr, Hv, nullptr);
}

int pantr_batch_eval_grad_L(pantr_batch_t *b, int32_t B, const double *x0, const double *u, const double *y,
                            double *grad) {
    if (int rc = check_batch(b, B))
        return rc;
    if (!b->al()) { // m = 0: L = f = psi
        if (!x0 || !u || !grad)
            return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
        return run_eval(b, B, EVAL_PSI_GRAD, 0, x0, u, nullptr, nullptr, grad, nullptr);
    }
    if (!y)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "this problem has general constraints: y[B][m] is required");
    const std::vector<double> ones((size_t)B * b->m, 1.0); // the penalties do not enter grad_L
    return run_al_eval(b, B, ALM_EVAL_GRAD_L, x0, u, y, ones.data(), nullptr, nullptr, grad, nullptr);
}

int pantr_batch_eval_g(pantr_batch_t *b, int32_t B, const double *x0, const double *u, double *g) {
    if (int rc = check_batch(b, B))
        return rc;
    if (!b->al()) // m = 0: nothing to write
        return x0 && u ? PANTR_OK : fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    if (!g)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    // g(u) out of the psi sweep: with y = 0, Sigma = 1 and the box D collapsed to {0},
    // zeta = g, zhat = 0 and yhat = Sigma (zeta - zhat) = g exactly
    const std::vector<double> zeros((size_t)B * b->m, 0.0), ones((size_t)B * b->m, 1.0);
    std::vector<double> grad((size_t)B * b->n);
    const BoxData saved = b->box;
    for (int c = 0; c < 4; ++c)
        b->box.d_lb[c] = b->box.d_ub[c] = 0.0;
    int rc = run_al_eval(b, B, ALM_EVAL_PSI_GRAD, x0, u, zeros.data(), ones.data(), nullptr, nullptr, grad.data(), g);
    b->box = saved;
    return rc;
}

// ---- closed loops on the device -------------------------------------------------------------
int pantr_batch_mpc(pantr_batch_t *b, int32_t B, int32_t num_sim, int32_t warm, int32_t num_dist,
                    const double *x_init, const double *u_dist, double *states, double *inputs,
                    pantr_stats_t *stats) {
    return pantr_batch_mpc_record(b, B, num_sim, warm, num_dist, x_init, u_dist, states, inputs, stats, nullptr,
                                  nullptr, nullptr);
}

int pantr_batch_mpc_record(pantr_batch_t *b, int32_t B, int32_t num_sim, int32_t warm, int32_t num_dist,
                           const double *x_init, const double *u_dist, double *states, double *inputs,
                           pantr_stats_t *stats, double *guesses_u, double *guesses_y, double *multipliers) {
    if (int rc = check_batch(b, B))
        return rc;
    if (!u_dist || !states || !inputs || !stats)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    if (num_sim < 1 || num_dist < 0)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "num_sim must be >= 1 and num_dist >= 0");
    HIP_TRY(hipSetDevice(b->device));
    const size_t xb = (size_t)B * b->nx * sizeof(double), ub = (size_t)B * b->n * sizeof(double);
    // x0: the problem's initial state (or the caller's), u: the initial guess
    std::vector<double> x0((size_t)B * b->nx), xi(b->nx);
    pantr_batch_get_initial_state(b, xi.data());
    for (int i = 0; i < B; ++i)
        for (int l = 0; l < b->nx; ++l)
            x0[(size_t)i * b->nx + l] = x_init ? x_init[(size_t)i * b->nx + l] : xi[l];
    HIP_TRY(hipMemcpy(b->d_x0, x0.data(), xb, hipMemcpyHostToDevice));
    for (int i = 0; i < B; ++i) // u = initial guess (hanging_chain.cpp:135 / quadcopter.py:111)
        HIP_TRY(hipMemcpy(b->d_u + (size_t)i * b->n, b->d_uinit, (size_t)b->n * sizeof(double), hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemset(b->d_y, 0, std::max<size_t>((size_t)B * b->m, 1) * sizeof(double)));
    HIP_TRY(hipMemcpy(b->d_v, u_dist, (size_t)B * b->nu * sizeof(double), hipMemcpyHostToDevice));
    auto simulate = [&](const double *u, int stride) {
        if (b->al())
            with_quad_model(b, [&](auto model) {
                using M = decltype(model);
                hipLaunchKernelGGL(pantr_scalar_simulate_kernel<M>, dim3((B + 63) / 64), dim3(64), 0, 0, B, b->d_x0, u,
                                   stride, b->d_x0);
                return 0;
            });
        else
            hipLaunchKernelGGL(pantr_simulate_kernel<ChainModel>, dim3(B), dim3(WAVE), 0, 0, B, b->d_x0, u, stride,
                               b->d_x0);
    };
    for (int i = 0; i < num_dist; ++i) // initial disturbance (mpc-driver.cpp:153-154)
        simulate(b->d_v, b->nu);
    if (int rc = launch_solve(b, B, b->d_x0, b->d_u, b->d_y, b->d_stats, nullptr)) // untimed first solve (:158)
        return rc;
    simulate(b->d_u, b->n); // :160-161
    for (int i = 0; i < num_sim; ++i) {
        // warm: shift u by nu and y by nc (:168-175) | cold: initial guess, y = 0 (:192-197)
        hipLaunchKernelGGL(pantr_shift_uy_kernel, dim3(B), dim3(256), 0, 0, B, b->n, b->nu, b->m, b->nc, warm,
                           b->d_uinit, b->d_u, b->d_y);
        // record the state this step is solved for, then solve, record, advance the plant
        HIP_TRY(hipMemcpyAsync(states + (size_t)i * B * b->nx, b->d_x0, xb, hipMemcpyDeviceToHost, nullptr));
        if (guesses_u)
            HIP_TRY(hipMemcpyAsync(guesses_u + (size_t)i * B * b->n, b->d_u, ub, hipMemcpyDeviceToHost, nullptr));
        if (guesses_y && b->m > 0)
            HIP_TRY(hipMemcpyAsync(guesses_y + (size_t)i * B * b->m, b->d_y, (size_t)B * b->m * sizeof(double),
                                   hipMemcpyDeviceToHost, nullptr));
        if (int rc = launch_solve(b, B, b->d_x0, b->d_u, b->d_y, b->d_stats, nullptr))
            return rc;
        HIP_TRY(hipMemcpyAsync(inputs + (size_t)i * B * b->n, b->d_u, ub, hipMemcpyDeviceToHost, nullptr));
        if (multipliers && b->m > 0)
            HIP_TRY(hipMemcpyAsync(multipliers + (size_t)i * B * b->m, b->d_y, (size_t)B * b->m * sizeof(double),
                                   hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipMemcpyAsync(stats + (size_t)i * B, b->d_stats, (size_t)B * sizeof(DeviceStats),
                               hipMemcpyDeviceToHost, nullptr));
        simulate(b->d_u, b->n); // :205
    }
    HIP_TRY(hipMemcpyAsync(states + (size_t)num_sim * B * b->nx, b->d_x0, xb, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return PANTR_OK;
}

} // extern "C"

// ---- the reference's generated problem library as a cross-checker (host/casadi_dll.hpp) ----------
struct pantr_casadi_dll {
    host::CasadiDll dll;
    explicit pantr_casadi_dll(const char *path) : dll(path) {}
};

namespace {
/// C++ exceptions of the loader -> the error codes of this ABI (the reference lets them propagate).
template <class F>
int dll_guard(F &&f) {
    try {
        return f();
    } catch (const std::out_of_range &e) {
        return fail(PANTR_ERR_OUT_OF_RANGE, e.what());
    } catch (const std::invalid_argument &e) {
        return fail(PANTR_ERR_INVALID_ARGUMENT, e.what());
    } catch (const std::exception &e) {
        return fail(PANTR_ERR_RUNTIME, e.what());
    }
}
} // namespace

extern "C" {

int pantr_casadi_dll_open(const char *so_path, pantr_casadi_dll_t **out) {
    if (!so_path || !out)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    return dll_guard([&] {
        *out = new pantr_casadi_dll(so_path);
        return PANTR_OK;
    });
}

void pantr_casadi_dll_close(pantr_casadi_dll_t *d) { delete d; }

int pantr_casadi_dll_function_info(pantr_casadi_dll_t *d, const char *name, int32_t *n_in, int32_t *n_out,
                                   int64_t *nnz_in, int64_t *nnz_out, int32_t cap, int64_t work[4]) {
    if (!d || !name)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    return dll_guard([&] {
        host::CasadiFunction &f = d->dll.function(name);
        if (n_in)
            *n_in = (int32_t)f.n_in;
        if (n_out)
            *n_out = (int32_t)f.n_out;
        for (int32_t i = 0; nnz_in && i < cap && i < f.n_in; ++i)
            nnz_in[i] = f.nnz_in[i];
        for (int32_t j = 0; nnz_out && j < cap && j < f.n_out; ++j)
            nnz_out[j] = f.nnz_out[j];
        if (work) {
            work[0] = f.sz_arg; work[1] = f.sz_res; work[2] = f.sz_iw; work[3] = f.sz_w;
        }
        return PANTR_OK;
    });
}

int pantr_casadi_dll_call(pantr_casadi_dll_t *d, const char *name, int32_t n_in, const double *const *args,
                          int32_t n_out, double *const *res) {
    if (!d || !name || (n_in > 0 && !args) || (n_out > 0 && !res) || n_in < 0 || n_out < 0)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    return dll_guard([&] {
        d->dll.function(name).call(std::vector<const double *>(args, args + n_in),
                                   std::vector<double *>(res, res + n_out));
        return PANTR_OK;
    });
}

int pantr_batch_check_casadi_dll(pantr_batch_t *b, const char *so_path, int32_t samples, double *max_rel_err) {
    if (!b || !so_path || !max_rel_err)
        return fail(PANTR_ERR_INVALID_ARGUMENT, "null argument");
    if (samples < 1)
        return fail(PANTR_ERR_OUT_OF_RANGE, "samples must be >= 1");
    return dll_guard([&]() -> int {
        const int n = b->n, m = b->m, nx = b->nx;
        host::CasadiProblemFunctions dll(so_path, n, m, nx); // refuses a library of other sizes
        // ---- the sample points (fixed seed: the same points in every run) ----
        std::vector<double> Clb(n), Cub(n), Dlb(std::max(m, 1)), Dub(std::max(m, 1));
        pantr_batch_get_box_C(b, Clb.data(), Cub.data());
        if (m > 0)
            pantr_batch_get_box_D(b, Dlb.data(), Dub.data());
        std::mt19937_64 rng(12345);
        std::uniform_real_distribution<double> U(-1.0, 1.0);
        const size_t S = (size_t)samples;
        std::vector<double> X0(S * nx), Uv(S * n), V(S * n), Y(S * std::max(m, 1)), Sig(S * std::max(m, 1));
        for (size_t s = 0; s < S; ++s) {
            for (int l = 0; l < nx; ++l) // the initial state, perturbed (chain: well inside the spring lengths)
                X0[s * nx + l] = b->x_init[l] + 0.02 * U(rng);
            for (int i = 0; i < n; ++i) { // around the initial guess, inside box C
                const double lo = std::isfinite(Clb[i]) ? Clb[i] : b->u_guess[i] - 1.0,
                             hi = std::isfinite(Cub[i]) ? Cub[i] : b->u_guess[i] + 1.0;
                const double x = b->u_guess[i] + 0.3 * (hi - lo) * U(rng);
                Uv[s * n + i]  = std::fmin(std::fmax(x, lo), hi);
                V[s * n + i]   = U(rng);
            }
            for (int j = 0; j < m; ++j) {
                Y[s * m + j]   = U(rng);
                Sig[s * m + j] = 1.0 + 49.5 * (1.0 + U(rng)); // 1 .. 100
            }
        }
        // ---- the device side, in chunks of the handle's batch size ----
        std::vector<double> psi_d(S), grad_d(S * n), Hv_d(S * n), g_d(S * std::max(m, 1)), yhat_d(S * std::max(m, 1));
        for (size_t s0 = 0; s0 < S; s0 += (size_t)b->maxB) {
            const int32_t B = (int32_t)std::min<size_t>(b->maxB, S - s0);
            const double *x0 = &X0[s0 * nx], *u = &Uv[s0 * n], *v = &V[s0 * n];
            int rc;
            if (m == 0) {
                if ((rc = pantr_batch_eval_psi_grad_psi(b, B, x0, u, &psi_d[s0], &grad_d[s0 * n])) ||
                    (rc = pantr_batch_eval_hess_psi_prod(b, B, x0, u, v, &Hv_d[s0 * n])))
                    return rc;
            } else {
                const double *y = &Y[s0 * m], *sg = &Sig[s0 * m];
                if ((rc = pantr_batch_eval_psi_grad_psi_al(b, B, x0, u, y, sg, &psi_d[s0], &grad_d[s0 * n], &yhat_d[s0 * m])) ||
                    (rc = pantr_batch_eval_hess_psi_prod_al(b, B, x0, u, y, sg, v, &Hv_d[s0 * n])) ||
                    (rc = pantr_batch_eval_g(b, B, x0, u, &g_d[s0 * m])))
                    return rc;
            }
        }
        // ---- the generated library, point by point ----
        auto rel = [](const double *ref, const double *got, int len) {
            double num = 0, den = 0;
            for (int i = 0; i < len; ++i) {
                const double d = std::fabs(ref[i] - got[i]);
                num = d > num || d != d ? d : num; // (a NaN on either side is an error of NaN)
                den = std::fmax(den, std::fabs(ref[i]));
            }
            return num / std::fmax(den, 1e-300);
        };
        double err[PANTR_DLL_CHECK_FIELDS] = {0, 0, dll.has_hess_psi_prod() ? 0.0 : -1.0,
                                              (m > 0 && dll.has_g()) ? 0.0 : -1.0, m > 0 ? 0.0 : -1.0};
        auto worst = [](double &acc, double e) { acc = (e > acc || e != e) ? e : acc; };
        std::vector<double> grad(n), Hv(n), g(std::max(m, 1)), yhat(std::max(m, 1));
        for (size_t s = 0; s < S; ++s) {
            const double *x = &Uv[s * n], *p = &X0[s * nx], *y = m ? &Y[s * m] : nullptr, *sg = m ? &Sig[s * m] : nullptr,
                         *zl = m ? Dlb.data() : nullptr, *zu = m ? Dub.data() : nullptr;
            const double psi = dll.eval_psi_grad_psi(x, p, y, sg, zl, zu, grad.data());
            worst(err[0], rel(&psi, &psi_d[s], 1));
            worst(err[1], rel(grad.data(), &grad_d[s * n], n));
            if (dll.has_hess_psi_prod()) {
                dll.eval_hess_psi_prod(x, p, y, sg, 1.0, zl, zu, &V[s * n], Hv.data());
                worst(err[2], rel(Hv.data(), &Hv_d[s * n], n));
            }
            if (m > 0) {
                if (dll.has_g()) {
                    dll.eval_g(x, p, g.data());
                    worst(err[3], rel(g.data(), &g_d[s * m], m));
                }
            }
            const double psi2 = dll.eval_psi(x, p, y, sg, zl, zu, m ? yhat.data() : nullptr);
            worst(err[0], rel(&psi2, &psi_d[s], 1));
            if (m > 0)
                worst(err[4], rel(yhat.data(), &yhat_d[s * m], m));
        }
        std::copy(err, err + PANTR_DLL_CHECK_FIELDS, max_rel_err);
        return PANTR_OK;
    });
}

} // extern "C"
