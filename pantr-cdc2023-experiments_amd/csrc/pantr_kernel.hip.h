// Batched PANTR inner solver for gfx950: one 64-wide wavefront (= one workgroup) solves one
// MPC instance from start to finish inside a single launch.
//
// What it replaces (the reference only *names* these; their source is the absent alpaqa
// submodule, .gitmodules:1-4):
//   alpaqa::ALMSolver<PANTRSolver<NewtonTRDirection>>::operator()   drivers/mpc-driver.cpp:104-115,217,364-370
//   PANTRParams / NewtonTRDirectionParams / SteihaugCGParams          new-benchmarks-paper/dodo.py:12-33
//   problem functions psi, psi_grad_psi, hess_psi_prod, grad_L        problems/problems/formulation.hpp:79-99
//   box prox step / inactive set                                      problems/problems/formulation.hpp:220-227
//
// Data placement (DESIGN.md "Layout"):
//   LDS     : the decision-space vectors of the trust-region subproblem (q, Hv output, r_J, and the CG
//             vectors z, r, d, step; the step slot doubles as the in/out vector of the solve), n_pad
//             doubles each; the three iterates x {x, xhat, grad, p} too, except for PANTR on the
//             hanging chain (WaveProblem<M, true>), which keeps them in HBM
//   HBM/L2  : the state trajectory X[(N+1)][64] and its tangent Xt written by the forward
//             sweeps and re-read (one stage ahead) by the reverse sweeps, one 512-byte row
//             per stage per wavefront
//   VGPR    : the stage state (one component per lane), adjoints, RK4 intermediates
//   SGPR    : every scalar of the algorithm (psi, gamma, L, radius, rho, ...), made uniform
//             with readfirstlane after each wavefront all-reduce
#pragma once
#include "wave_math.hip.h"
#include <stdint.h>
#include <type_traits>

namespace pantr {

enum SolverStatus : int32_t {
    Busy = 0, Converged = 1, MaxTime = 2, MaxIter = 3, NotFinite = 4, NoProgress = 5,
    Interrupted = 6, Exception = 7,
};
enum StopCrit : int32_t {
    ApproxKKT = 0, ApproxKKT2, ProjGradNorm, ProjGradNorm2, ProjGradUnitNorm, ProjGradUnitNorm2,
    FPRNorm, FPRNorm2, IpoptCrit, LBFGSBpp,
};

/// Box C = U tiled N times and box D = D tiled N times + D_N (formulation.hpp:220-238), one period
/// each: numeric problem data (the reference keeps them in the problem's .csv sidecar,
/// generate_problem.py:62-69), so they reach the kernels as arguments, not as compiled constants.
struct BoxData {
    double u_lb[4], u_ub[4], d_lb[4], d_ub[4];
};
__device__ __forceinline__ double box_pick(const double (&a)[4], int c) {
    return c == 0 ? a[0] : c == 1 ? a[1] : c == 2 ? a[2] : a[3];
}

/// Solver options as the kernel consumes them (filled from PANTRParams & co. on the host).
struct KernelParams {
    double L_0, lip_eps, lip_delta, Lgamma_factor, max_time_s, L_min, L_max;
    double quadratic_upperbound_tolerance_factor, TR_tolerance_factor;
    double ratio_threshold_acceptable, ratio_threshold_good;
    double radius_factor_rejected, radius_factor_acceptable, radius_factor_good;
    double initial_radius, min_radius;
    double hessian_vec_factor, tol_scale, tol_scale_root, tol_max, max_iter_factor;
    double tolerance;
    double spec_patience_s;   ///< speculation: how long the solver wavefront waits for a result (`spec_patience=`)
    double solver_max_time_s; ///< solver.max_time alone (max_time_s above = min(solver.max_time, alm.max_time))
    double tick_s;            ///< seconds per wall_clock64() tick (hipDeviceAttributeWallClockRate)
    // PANOC line search and L-BFGS (method = panoc, strucpanoc; panoc_kernel.hip.h)
    double min_linesearch_coefficient, linesearch_strictness_factor, linesearch_tolerance_factor;
    double lbfgs_min_div_fac, lbfgs_min_abs_s;
    int32_t max_iter, stop_crit, max_no_progress, disable_acceleration;
    int32_t always_overwrite_results, force_linesearch, lbfgs_memory, lbfgs_force_pos_def;
    BoxData box;
};

/// Per-instance result record (mirrors pantr_stats_t in include/pantr_mi355x.h).
struct DeviceStats {
    double eps, final_gamma, final_psi, final_fbe, final_radius, elapsed_s;
    int32_t status, iterations, cg_iterations, accepted, rejected, qub_backtracks;
    int32_t n_psi, n_grad_psi, n_psi_grad_psi, n_grad_L, n_hess_psi_prod, n_prox_grad_step;
    int32_t outer_iterations, inner_convergence_failures; ///< ALM
    int32_t n_hess_psi_prod_calls; ///< n_hess_psi_prod + the zero products the direction skips (alpaqa's count)
    int32_t spec_timeouts;         ///< speculation: results the solver wavefront stopped waiting for
    double delta, penalty_norm;                           ///< ||Sigma^{-1}(yhat - y)||_inf, ||Sigma|| / sqrt(m)
    int32_t linesearch_failures, linesearch_backtracks, lbfgs_failures, lbfgs_rejected, tau_1_accepted,
        count_tau; ///< PANOC
    double sum_tau;
    /// Seconds per phase; filled only by the profiling build (-DPANTR_PHASE_TIMING, `make phase`):
    /// [0] direction (hess_prod | L-BFGS apply), [1] psi_grad(candidate), [2] psi(xhat),
    /// [3] grad_L(xhat) + tape (PANTR) | L-BFGS update (PANOC).  Zero in the product build.
    double phase_s[4];
};
enum Phase : int { PH_DIRECTION = 0, PH_PSI_GRAD = 1, PH_PSI_XHAT = 2, PH_GRAD_XHAT = 3 };
/// Field by field (a struct copy would make the compiler keep the whole record in scratch memory).
__device__ __forceinline__ void write_stats(DeviceStats *o, const DeviceStats &s) {
    o->eps = s.eps; o->final_gamma = s.final_gamma; o->final_psi = s.final_psi; o->final_fbe = s.final_fbe;
    o->final_radius = s.final_radius; o->elapsed_s = s.elapsed_s;
    o->status = s.status; o->iterations = s.iterations; o->cg_iterations = s.cg_iterations;
    o->accepted = s.accepted; o->rejected = s.rejected; o->qub_backtracks = s.qub_backtracks;
    o->n_psi = s.n_psi; o->n_grad_psi = s.n_grad_psi; o->n_psi_grad_psi = s.n_psi_grad_psi;
    o->n_grad_L = s.n_grad_L; o->n_hess_psi_prod = s.n_hess_psi_prod; o->n_prox_grad_step = s.n_prox_grad_step;
    o->outer_iterations = s.outer_iterations; o->inner_convergence_failures = s.inner_convergence_failures;
    o->n_hess_psi_prod_calls = s.n_hess_psi_prod_calls; o->spec_timeouts = s.spec_timeouts;
    o->delta = s.delta; o->penalty_norm = s.penalty_norm;
    o->linesearch_failures = s.linesearch_failures; o->linesearch_backtracks = s.linesearch_backtracks;
    o->lbfgs_failures = s.lbfgs_failures; o->lbfgs_rejected = s.lbfgs_rejected;
    o->tau_1_accepted = s.tau_1_accepted; o->count_tau = s.count_tau; o->sum_tau = s.sum_tau;
    o->phase_s[0] = s.phase_s[0]; o->phase_s[1] = s.phase_s[1]; o->phase_s[2] = s.phase_s[2]; o->phase_s[3] = s.phase_s[3];
}

constexpr int TRACE_COLS = 11;
constexpr int NUM_VEC    = 20; ///< working vectors of the inner solvers (slots S_* inside them)
constexpr int NUM_ITER_VEC = 12; ///< three iterates x {x, xhat, grad, p}
/// PANTR on the hanging chain keeps only the vectors of the trust-region subproblem in LDS (q, Hv,
/// r_J, z, r, d, step; the decision vector a sweep reads stage by stage is staged in z's slot, which
/// is dead outside the CG loop); the three iterates live in HBM/L2 (WaveProblem<M, true>): they are
/// touched by element-wise passes only, every lane reading and writing its own elements as coalesced
/// 512-byte rows.  7 slots x 192 doubles + the landing buffers of the products (10 KiB one stage ahead,
/// 15.75 KiB two stages ahead: the one-wavefront kernel) = 20.5 / 26.5 KiB at N = 60 (six instances of the
/// one-wavefront kernel per CU) and 31 / 37 KiB at N = 120 (four per CU); with everything in LDS it was 40 KiB
/// and 70 KiB.
constexpr int NUM_CG_VEC = 7;
/// The in/out vector of an inner solve shares the slot of the direction's scratch vector (PANTR:
/// the CG step; PANOC: unused): it is read before the first and written after the last use of
/// that scratch.  Offset inside the CG block (slot PB::CG_BASE + C_XIO).
constexpr int C_XIO      = 7;
constexpr int S_XIO      = NUM_ITER_VEC + C_XIO; ///< with all 20 vectors in LDS (PANOC, scalar models)

struct SolveArgs {
    int32_t B, N, n, npad, nj;
    int32_t trace_cap, trace_instances, pad0;
    const double *x0;      ///< [B][nx] initial states (the problem parameter)
    double *u;             ///< [B][n]  in: initial guess, out: solution
    double *ws;            ///< [B][WaveProblem::workspace_doubles(N)] trajectories + tape
    double *lbfgs;         ///< [B][2 * lbfgs_memory * npad] L-BFGS rings (method != pantr) or nullptr
    DeviceStats *stats;    ///< [B]
    double *trace;         ///< [trace_instances][trace_cap][TRACE_COLS] or nullptr
    const int32_t *order;  ///< optional launch order (instance index per workgroup) or nullptr
    const double *lbv, *ubv; ///< [npad] box C tiled over the horizon, padding 0 (the passes of the NJ > 0 kernels read them)
    KernelParams prm;
};

// ---------------------------------------------------------------------------------------
// Element-wise passes with a compile-time number of elements per lane (PB::NJ > 0)
// ---------------------------------------------------------------------------------------
// A lone wavefront pays for every instruction and for every dependent LDS round trip.  The generic passes
// (`for (j < nj) { i = 64 j + lane; if (i < n) ... }`) run one trip per element under an exec mask; with NJ
// known at compile time and the padding of every vector kept at zero a pass is straight-line code: all of a
// lane's elements are loaded first (one wait), the arithmetic of the NJ elements interleaves, the stores
// follow.  Element by element and sum by sum the operations - and their order - are those of the generic
// passes: a padding element contributes +0 to a sum (or is masked out), which changes no bit.
template <int NJ>
struct Owned {
    double v[NJ];
    // (the lane index passes through an empty asm: the address arithmetic of a pass stays at the pass instead of
    // being hoisted out of the solver's loops into registers that then live - and spill - across the sweeps)
    __device__ __forceinline__ static Owned load(const double *a, int lane) {
        asm volatile("" : "+v"(lane));
        Owned r;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            r.v[j] = a[j * WAVE + lane];
        return r;
    }
    __device__ __forceinline__ void store(double *a, int lane) const {
        asm volatile("" : "+v"(lane));
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            a[j * WAVE + lane] = v[j];
    }
    __device__ __forceinline__ double &operator[](int j) { return v[j]; }
    __device__ __forceinline__ const double &operator[](int j) const { return v[j]; }
};
/// Two / three wave_sum()s level by level side by side (the same sums: each chain is wave_sum's).
__device__ __forceinline__ void wave_sum2(double &x, double &y) {
    double a, b, c, d;
    row_partner<true>(x, a, b);
    row_partner<true>(y, c, d);
    x = a + b;
    y = c + d;
    row_partner<false>(x, a, b);
    row_partner<false>(y, c, d);
    x = a + b;
    y = c + d;
    double px = dpp_partner<0x128>(x), py = dpp_partner<0x128>(y);
    x = x + px;
    y = y + py;
    px = xor4_partner(x), py = xor4_partner(y);
    x = x + px;
    y = y + py;
    px = dpp_partner<0x4E>(x), py = dpp_partner<0x4E>(y);
    x = x + px;
    y = y + py;
    px = dpp_partner<0xB1>(x), py = dpp_partner<0xB1>(y);
    x = x + px;
    y = y + py;
}
__device__ __forceinline__ void wave_sum3(double &x, double &y, double &z) {
    double a, b, c, d, e, f;
    row_partner<true>(x, a, b);
    row_partner<true>(y, c, d);
    row_partner<true>(z, e, f);
    x = a + b;
    y = c + d;
    z = e + f;
    row_partner<false>(x, a, b);
    row_partner<false>(y, c, d);
    row_partner<false>(z, e, f);
    x = a + b;
    y = c + d;
    z = e + f;
    double px = dpp_partner<0x128>(x), py = dpp_partner<0x128>(y), pz = dpp_partner<0x128>(z);
    x = x + px;
    y = y + py;
    z = z + pz;
    px = xor4_partner(x), py = xor4_partner(y), pz = xor4_partner(z);
    x = x + px;
    y = y + py;
    z = z + pz;
    px = dpp_partner<0x4E>(x), py = dpp_partner<0x4E>(y), pz = dpp_partner<0x4E>(z);
    x = x + px;
    y = y + py;
    z = z + pz;
    px = dpp_partner<0xB1>(x), py = dpp_partner<0xB1>(y), pz = dpp_partner<0xB1>(z);
    x = x + px;
    y = y + py;
    z = z + pz;
}

// ---------------------------------------------------------------------------------------
// Speculation on a rejected candidate (two-wavefront workgroups; hanging chain, method = pantr)
// ---------------------------------------------------------------------------------------
// Six iterations in ten of the benchmark reject their trust-region candidate and continue from the
// prox iterate xbar - whose next steps (psi(xhat(xbar)), the quadratic-upper-bound test, grad_L(xhat)
// and the tape there) depend on nothing the trust-region subproblem computes.  With a second
// wavefront in the workgroup, those steps run WHILE the main wavefront solves the subproblem: the
// main wavefront posts the request before its CG loop, and on a rejection takes the result (the
// value, the gradient, the recorded tape) instead of computing it; on an acceptance the result is
// dropped.  Both wavefronts run the same device functions on the same inputs, so a taken result has
// the bits the main wavefront would have produced.  Mailbox in LDS; sequence numbers written with
// release, read with acquire semantics at workgroup scope.
struct alignas(16) SpecBox { // a multiple of 16 bytes: the dynamic LDS behind it stays 16-byte aligned
    int req_seq, done_seq; ///< main: requests posted; helper: requests finished
    int abort;             ///< main: it stopped waiting for a result - the helper ends at its next check
    int stopped;           ///< helper: it has ended (set on every way out of its loop): nothing of the instance is written by it any more
    int cmd;               ///< SPEC_EVAL | SPEC_EXIT
    int xh_slot;           ///< iterate vector (pb.ivec index) holding the point to evaluate: xhat of the prox iterate
    int yh_slot;           ///< problems with general constraints: the m-vector that receives yhat there
    int tape_sel;          ///< tape buffer (0 | 1) the helper records into
    int ok;                ///< response: the quadratic upper bound held (no backtracking: the result is usable)
    double psi_x, gradTp, pTp, L, qub_factor, L_max; ///< request: what the quadratic-upper-bound test needs
    double psi_xh;         ///< response: psi(xhat)
    // SPEC_MODEL: the model value m(p) = p'r_J + 1/2 p'H_JJ p of the finished CG's step - the one product
    // of an iteration nothing waits for until the ratio test: the main wavefront evaluates the
    // candidate meanwhile
    int p_off, rj_off;     ///< request: offsets (doubles, from the workgroup's dynamic LDS) of p and r_J
    unsigned mask[WAVE];   ///< request: the inactive set, one bit per owned element and lane
    double mval;           ///< response
};
enum : int { SPEC_EVAL = 1, SPEC_EXIT = 2, SPEC_MODEL = 3 };

// ---------------------------------------------------------------------------------------
// Wave-cooperative problem functions of the single-shooting problem
// ---------------------------------------------------------------------------------------
/// DEPTH: how many stages ahead a Hessian-vector product requests its tape rows (1 | 2; hess_prod_tape).
/// NJ_: 0, or nj = npad / 64 known at compile time - the element-wise passes are then straight-line code (Owned<NJ>).
/// RECOMP (DEPTH == 2 only): the products keep the tangent state of every stage instead of the tangent spring vectors
/// and run the stage's forward tangent again inside the adjoint sweep (chain_model.hip.h, s_store): 7 instead of 9.25 KiB
/// per stage for 86 more instructions - what the one-wavefront kernel of large, memory-bound batches wants.
template <class M, bool ITER_GLOBAL = false, bool SPEC = false, int DEPTH = 1, int NJ_ = 0, bool RECOMP = false>
struct WaveProblem {
    static_assert(!RECOMP || DEPTH == 2, "the recomputing products are a form of the two-stages-ahead sweeps");
    static_assert(DEPTH == 1 || DEPTH == 2, "one or two stages ahead");
    static constexpr int PREFETCH_DEPTH = DEPTH;
    using Lane  = typename M::Lane;
    using State = typename M::State;
    const Lane &L;
    const BoxData &box; ///< bounds of the decision variables (kernel argument)
    double *lds; ///< base of this wavefront's LDS vectors
    int N, n, npad, nj;
    State x0;       ///< this lane's share of the initial state
    double *X;      ///< global: primal trajectory rows, M::TRAJ_DOUBLES per stage
    double *Xt;     ///< global: tangent-state rows of the current Hv, M::XT_DOUBLES per stage
    double4 *tape;  ///< global: recorded second-order rows, M::TAPE_D4_PER_STAGE per stage
    double4 *st;    ///< global: tangent spring vectors of the current Hv, M::ST_D4_PER_STAGE per stage
    char *tbuf0, *tbuf1; ///< LDS: stage-sized landing buffers (distinct objects) for the LDS-DMA (a third: tbuf2 below)
    /// The first stage's K rows of the current tape are already in flight to (or in) tbuf0: every
    /// product leaves them there for the next one, so only the first product after a recording
    /// waits for them.
    mutable bool primed = false;
    char *tbuf2 = nullptr; ///< DEPTH == 2: the third landing buffer (stage k's rows land in buffer k mod 3)
    double *giter = nullptr; ///< ITER_GLOBAL: this instance's NUM_ITER_VEC iterate vectors in the HBM workspace
    // speculation (see SpecBox): nullptr / unused in one-wavefront workgroups
    SpecBox *spec = nullptr;   ///< LDS mailbox
    // (the two tape buffers and the helper's rows are addressed from the instance's workspace base when
    // they are needed - tape_buf(), helper_*() below - rather than held in registers through the solve)
    double *wg_lds = nullptr;               ///< base of the workgroup's dynamic LDS (both wavefronts' vectors)
    double *ws0    = nullptr;               ///< base of this instance's HBM workspace
    mutable bool spec_stashed = false, spec_stash_ok = false; ///< result of a finished SPEC_EVAL kept while the mailbox
    mutable double spec_stash_psi = 0.0;                      ///< carries a SPEC_MODEL request
    mutable int tape_cur = 0;               ///< which of the two `tape` is
    mutable bool spec_pending = false;      ///< a request is out whose result has been neither taken nor dropped
    mutable bool spec_dead = false;         ///< the helper missed a deadline: no further requests
    mutable int spec_timeouts = 0;          ///< how often that happened (stats.spec_timeouts)
    double spec_tick_s = 1e-8, spec_patience_s = 1.0; ///< clock period; how long the main wavefront waits for a result
#ifdef PANTR_PHASE_TIMING
    mutable long long fwd_cycles = 0; ///< profiling build: shader cycles spent in the forward half of the products
#endif

    /// LDS vector `slot`; the CG block starts at CG_BASE.
    static constexpr int CG_BASE   = ITER_GLOBAL ? 0 : NUM_ITER_VEC;
    static constexpr int LDS_SLOTS = ITER_GLOBAL ? NUM_CG_VEC : NUM_VEC;
    /// slot of the in/out vector of an inner solve = the direction's step vector (see C_XIO)
    static constexpr int XIO_SLOT  = CG_BASE + (ITER_GLOBAL ? NUM_CG_VEC - 1 : C_XIO);
    /// ITER_GLOBAL: staging copy of a sweep's decision vector; shares the slot of the CG iterate z, which
    /// lives only inside the CG loop (no sweep runs there)
    static constexpr int S_STAGE   = CG_BASE + 3;
    /// ITER_GLOBAL: the adjoint sweep collects the gradient here and copies it to its HBM vector once at
    /// the end (a store to HBM inside the stage loop is waited for - with everything else outstanding -
    /// by the next stage's read of its prefetched state).  The slot of the CG residual r: dead outside
    /// the CG loop like z.
    static constexpr int S_GSTAGE  = CG_BASE + 4;
    /// LDS vector slots of a helper wavefront (SpecBox): slots 0..S_GSTAGE, of which it uses the product
    /// output (CG_BASE + 1) and the two staging slots
    static constexpr int HELPER_SLOTS = S_GSTAGE + 1;
    __device__ __forceinline__ double *vec(int slot) const { return lds + slot * npad; }
    /// Exchange row of the component gather: XS_DOUBLES doubles of dynamic LDS behind the kernel's vectors.
    static constexpr int XS_DOUBLES = M::PAIR_XS_DOUBLES; // 32 doubles per half of the wavefront (comps_pair)
    double *xrow = nullptr;
    const double *lbv = nullptr, *ubv = nullptr; ///< HBM, shared by the batch: box C as n_pad-vectors (SolveArgs)
    __device__ __forceinline__ double *xs() const { return xrow; }
    /// Iterate vector `slot` (0..11): HBM (ITER_GLOBAL) or the first twelve LDS slots.
    __device__ __forceinline__ double *ivec(int slot) const {
        if constexpr (ITER_GLOBAL)
            return giter + (size_t)slot * npad;
        else
            return lds + slot * npad;
    }
    /// The decision vector of a sweep where the stage loops can read it at LDS latency.
    __device__ __forceinline__ const double *staged(const double *u) const {
        if constexpr (!ITER_GLOBAL)
            return u;
        double *s = vec(S_STAGE);
        if constexpr (NJ_ > 0) {
            Owned<NJ_>::load(u, L.l).store(s, L.l); // (the loads of a lane's elements in flight together)
        } else {
            for (int j = 0; j < nj; ++j) { // each lane copies the elements it owns (and wrote)
                int i = j * WAVE + L.l;
                if (i < n)
                    s[i] = u[i];
            }
        }
        return s;
    }

    /// Static LDS of a kernel using this problem: the DEPTH + 1 landing buffers.
    /// bytes of one landing buffer (DEPTH == 2: with the tangent-state row behind the tangent tape)
    // (products that recompute the forward tangent have no tangent-state row to land: 3 x 256 bytes less, which is what
    // lets a sixth workgroup of the N = 60 one-wavefront kernel into a compute unit's 160 KiB)
    static constexpr int LANDING_BYTES = DEPTH == 2 && !RECOMP ? M::LDS_STAGE_BYTES_XT : M::LDS_STAGE_BYTES;
    __host__ __device__ static size_t tape_lds_bytes() { return (size_t)(DEPTH + 1) * LANDING_BYTES; }
    /// Doubles of HBM workspace one instance needs: X and Xt rows, the tape, the tangent tape, and
    /// (used by the ITER_GLOBAL layout) the iterate vectors.
    __host__ __device__ static size_t workspace_doubles(int N) {
        const size_t npad_ = (size_t)((M::nu * N + WAVE - 1) / WAVE) * WAVE;
        return (size_t)(N + 1) * (M::TRAJ_DOUBLES + M::XT_DOUBLES) +
               (size_t)N * (M::TAPE_D4_PER_STAGE + M::ST_D4_PER_STAGE) * 4 + NUM_ITER_VEC * npad_ +
               // speculation: second tape, the helper's trajectory rows, gradient, tangent rows and tangent tape
               (size_t)N * M::TAPE_D4_PER_STAGE * 4 + (size_t)(N + 1) * M::TRAJ_DOUBLES + npad_ +
               (size_t)(N + 1) * M::XT_DOUBLES + (size_t)N * M::ST_D4_PER_STAGE * 4;
    }
    __device__ __forceinline__ void bind_workspace(double *ws, int inst) {
        X     = ws + (size_t)inst * workspace_doubles(N);
        ws0   = X;
        Xt    = X + (size_t)(N + 1) * M::TRAJ_DOUBLES;
        tape  = reinterpret_cast<double4 *>(Xt + (size_t)(N + 1) * M::XT_DOUBLES);
        st    = tape + (size_t)N * M::TAPE_D4_PER_STAGE;
        giter = reinterpret_cast<double *>(st + (size_t)N * M::ST_D4_PER_STAGE);
    }
    /// Tape buffer 0 | 1 (`tape` is one of them).
    __device__ __forceinline__ double4 *tape_buf(int which) const {
        double *t0 = ws0 + (size_t)(N + 1) * (M::TRAJ_DOUBLES + M::XT_DOUBLES);
        double *t1 = t0 + (size_t)N * (M::TAPE_D4_PER_STAGE + M::ST_D4_PER_STAGE) * 4 + (size_t)NUM_ITER_VEC * npad;
        return reinterpret_cast<double4 *>(which ? t1 : t0);
    }
    /// The helper's trajectory rows, gradient, tangent rows and tangent tape (behind tape buffer 1).
    __device__ __forceinline__ double *helper_X() const {
        return reinterpret_cast<double *>(tape_buf(1) + (size_t)N * M::TAPE_D4_PER_STAGE);
    }
    __device__ __forceinline__ double *helper_G() const { return helper_X() + (size_t)(N + 1) * M::TRAJ_DOUBLES; }
    __device__ __forceinline__ double *helper_Xt() const { return helper_G() + npad; }
    __device__ __forceinline__ double4 *helper_st() const {
        return reinterpret_cast<double4 *>(helper_Xt() + (size_t)(N + 1) * M::XT_DOUBLES);
    }

    // ---- speculation: the main wavefront's side (no-ops without a mailbox) ---------------------
    static constexpr bool HAS_SPEC = SPEC; ///< only the two-wavefront kernel instantiates the hooks
    static constexpr int NJ = NJ_;         ///< elements per lane known at compile time (Owned<NJ>), or 0
    static_assert(NJ_ == 0 || ITER_GLOBAL, "the compile-time passes are those of method = pantr");
    static_assert(!SPEC || ITER_GLOBAL, "speculation needs the iterates in HBM");
    /// Post "evaluate psi, the bound test and grad_L + tape at ivec(xh_slot)"; the helper starts at once.
    __device__ __forceinline__ void spec_request(int xh_slot, double psi_x, double gradTp, double pTp, double Lip,
                                                 const KernelParams &P, int /*yh_slot*/) const {
        if (!spec || spec_dead)
            return;
        wave_sync(); // the iterate vectors the helper will read are written (and, in HBM, acknowledged)
        if (L.l == 0) {
            spec->cmd = SPEC_EVAL; spec->xh_slot = xh_slot; spec->tape_sel = 1 - tape_cur;
            spec->psi_x = psi_x; spec->gradTp = gradTp; spec->pTp = pTp; spec->L = Lip;
            spec->qub_factor = P.quadratic_upperbound_tolerance_factor; spec->L_max = P.L_max;
            __hip_atomic_store(&spec->req_seq, spec->req_seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        spec_pending = true;
    }
    /// Wait until the helper has finished the posted request (bounded work: it always does).  false: it
    /// has not within the time after which the helper itself gives up (spec_serve) - nothing of its
    /// result is used then.
    __device__ __forceinline__ bool spec_wait() const {
        const int want = __hip_atomic_load(&spec->req_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const long long t0 = wall_clock64();
        // (spec_patience = 0: the first result asked for counts as late whether it is there or not - how the tests
        // reach this path; the helper is normally done long before the solver asks)
        while (!(spec_patience_s > 0) ||
               __hip_atomic_load(&spec->done_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != want) {
            __builtin_amdgcn_s_sleep(2);
            if ((double)(wall_clock64() - t0) * spec_tick_s > spec_patience_s) {
                spec_dead = true; // no further requests; the helper ends at its next look at the flag
                ++spec_timeouts;
                if (L.l == 0)
                    __hip_atomic_store(&spec->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return false;
            }
        }
        return true;
    }
    /// The candidate was accepted (or nothing depends on the result any more): let the helper finish and forget it.
    __device__ __forceinline__ void spec_drop() const {
        if (spec && spec_pending)
            spec_wait();
        spec_pending = spec_stashed = false;
    }
    /// The candidate was rejected: the value of psi(xhat) if the helper's result is usable.
    __device__ __forceinline__ bool spec_take(double &psi_xh) const {
        if (spec_stashed) { // taken out of the mailbox when the SPEC_MODEL request went in
            spec_stashed = false;
            psi_xh       = spec_stash_psi;
            return spec_stash_ok;
        }
        if (!spec || !spec_pending)
            return false;
        const bool done = spec_wait();
        spec_pending = false;
        const bool ok = done && uniform(spec->ok) != 0;
        if (ok)
            psi_xh = uniform(spec->psi_xh);
        return ok;
    }
    /// Ask the helper for the model value of p (an LDS vector of this wavefront) with r_J and the
    /// inactive set `mask`.  false: no helper, or it is still busy with the SPEC_EVAL request (a short
    /// CG loop) - the caller evaluates the product itself.
    __device__ __forceinline__ bool spec_post_model(const double *p, const double *rJ, unsigned mask) const {
        if (!spec || spec_dead)
            return false;
        if (spec_pending) {
            const int want = __hip_atomic_load(&spec->req_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__hip_atomic_load(&spec->done_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != want)
                return false;
            spec_stash_ok  = uniform(spec->ok) != 0;
            spec_stash_psi = uniform(spec->psi_xh);
            spec_stashed   = true;
            spec_pending   = false;
        }
        spec->mask[L.l] = mask;
        wave_sync(); // p, r_J and the mask are written
        if (L.l == 0) {
            spec->cmd = SPEC_MODEL; spec->tape_sel = tape_cur;
            spec->p_off = (int)(p - wg_lds); spec->rj_off = (int)(rJ - wg_lds);
            __hip_atomic_store(&spec->req_seq, spec->req_seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return true;
    }
    /// The model value asked for; false if the helper did not deliver (the caller evaluates it itself).
    __device__ __forceinline__ bool spec_take_model(double &mval) const {
        if (!spec_wait())
            return false;
        mval = uniform(spec->mval);
        return true;
    }
    /// The gradient and the tape that go with a taken value: grad <- the helper's, tape <- the one it recorded.
    __device__ __forceinline__ void spec_adopt(double *grad) const {
        wave_sync();
        const double *Gh = helper_G();
        if constexpr (NJ_ > 0) {
            Owned<NJ_>::load(Gh, L.l).store(grad, L.l); // (the helper's gradient vector: padding never written, zero)
        } else {
            for (int j = 0; j < nj; ++j) {
                int i = j * WAVE + L.l;
                if (i < n)
                    grad[i] = Gh[i];
            }
        }
        tape_cur = 1 - tape_cur;
        const_cast<WaveProblem *>(this)->tape = tape_buf(tape_cur);
        primed = false;
    }
    __device__ __forceinline__ void spec_exit() const {
        if (!spec)
            return;
        spec_drop();
        if (L.l == 0) {
            spec->cmd = SPEC_EXIT;
            __hip_atomic_store(&spec->req_seq, spec->req_seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    /// The helper wavefront: serves requests until SPEC_EXIT (or until the instance's time is up: a
    /// helper never outlives max_time by more than one request).  `this` is the helper's own problem
    /// object: its own LDS vectors and exchange row, X = helper_X(), the same iterate vectors and tapes.
    __device__ __forceinline__ void spec_serve(const KernelParams &P, long long t_start) {
        int seen = 0;
        auto aborted = [&]() {
            return uniform(__hip_atomic_load(&spec->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0;
        };
        while (true) {
            int r;
            while ((r = __hip_atomic_load(&spec->req_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == seen) {
                __builtin_amdgcn_s_sleep(8);
                if (aborted() || (double)(wall_clock64() - t_start) * P.tick_s > 2.0 * P.max_time_s + 1.0)
                    return;
            }
            seen = r;
            const int cmd = uniform(spec->cmd);
            if (cmd == SPEC_EXIT || aborted())
                return;
            if (cmd == SPEC_MODEL) {
                const double *p = wg_lds + uniform(spec->p_off), *rJ = wg_lds + uniform(spec->rj_off);
                const unsigned mask = spec->mask[L.l];
                double *Hp = vec(CG_BASE + 1);
                tape   = tape_buf(uniform(spec->tape_sel));
                primed = false;
                hess_prod_tape(p, Hp);
                wait_vmem(); // the rows the product left in flight for a successor (there is none here)
                const double mval = dot(p, rJ) + 0.5 * dot_masked(p, Hp, mask);
                if (L.l == 0) {
                    spec->mval = mval;
                    __hip_atomic_store(&spec->done_seq, seen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                continue;
            }
            const double *xh = ivec(uniform(spec->xh_slot));
            tape             = tape_buf(uniform(spec->tape_sel));
            const double psi_xh = psi_forward<true>(xh);
            const double psi_x = uniform(spec->psi_x), gradTp = uniform(spec->gradTp), pTp = uniform(spec->pTp),
                         Lip = uniform(spec->L);
            const double margin = (1.0 + fabs(psi_x)) * uniform(spec->qub_factor);
            const bool violated = psi_xh > ((psi_x + gradTp) + (0.5 * Lip) * pTp) + margin;
            const bool ok       = !(Lip < uniform(spec->L_max) && violated);
            if (aborted()) // the solver stopped waiting (and evaluates this itself): nothing of the result is used
                return;
            if (ok)
                grad_reverse<true>(xh, helper_G());
            wave_sync(); // the gradient and the tape rows are written (and acknowledged) before the result is announced
            if (L.l == 0) {
                spec->psi_xh = psi_xh;
                spec->ok     = ok ? 1 : 0;
                __hip_atomic_store(&spec->done_seq, seen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }

    // ---- the interface pantr_inner() drives (m = 0: no multipliers) ----------------------------
    __device__ __forceinline__ double eval_psi_grad(const double *u, double *grad) const { return psi_grad<false>(u, grad); }
    __device__ __forceinline__ double eval_psi_xhat(const double *u, int) const { return psi_forward<true>(u); }
    __device__ __forceinline__ void eval_grad_xhat(const double *u, double *grad, int) const {
        grad_reverse<true>(u, grad);
        primed = false; // new tape
    }
    __device__ __forceinline__ void hess_prod(const double *, const double *v, double *Hv) const { hess_prod_tape(v, Hv); }
    __device__ __forceinline__ void update_multipliers(int) const {}

    /// psi(u) = sum_k l(x_k,u_k) + l_N(x_N); optionally stores the trajectory.
    template <bool STORE>
    __device__ __forceinline__ double psi_forward(const double *u_in) const {
        const double *u = staged(u_in);
        wave_sync(); // u was written by its owner lanes
        State s = x0, acc{0.0, 0.0};
        for (int k = 0; k < N; ++k) {
            double uc = M::load_input(L, u + M::nu * k);
            if (STORE)
                M::traj_store(L, X + (size_t)k * M::TRAJ_DOUBLES, s);
            M::cost_accumulate(L, s, uc, true, acc);
            s = M::rk4(L, s, uc, xs());
        }
        if (STORE)
            M::traj_store(L, X + (size_t)N * M::TRAJ_DOUBLES, s);
        M::cost_accumulate(L, s, 0.0, false, acc);
        return M::cost_total(L, acc, xs());
    }

    /// psi and grad psi (forward sweep, then adjoint sweep over the stored trajectory).
    /// RECORD: the adjoint sweep also writes the tape the Hessian-vector products replay.
    template <bool RECORD = false>
    __device__ __forceinline__ double psi_grad(const double *u, double *grad) const {
        double psi = psi_forward<true>(u);
        grad_reverse<RECORD>(u, grad);
        return psi;
    }
    /// The adjoint sweep alone, over the trajectory the last psi_forward<true>(u) stored (same u).
    template <bool RECORD = false>
    __device__ __forceinline__ void grad_reverse(const double *u_in, double *grad_out) const {
        const double *u = staged(u_in);
        double *grad = ITER_GLOBAL ? vec(S_GSTAGE) : grad_out;
        wave_sync(); // the sweep below re-reads rows this wavefront has just written
        State lam{0.0, 0.0};
        M::cost_grad_x_add(L, M::traj_load(L, X + (size_t)N * M::TRAJ_DOUBLES), lam);
        State s = M::traj_load(L, X + (size_t)(N - 1) * M::TRAJ_DOUBLES);
        for (int k = N - 1; k >= 0; --k) {
            State s_prev = s;
            if (k > 0) // one stage ahead
                s_prev = M::traj_load(L, X + (size_t)(k - 1) * M::TRAJ_DOUBLES);
            double uc = M::load_input(L, u + M::nu * k);
            State sbar;
            double ubar;
            M::rk4_adj(L, s, uc, lam, sbar, ubar, RECORD ? tape + (size_t)k * M::TAPE_D4_PER_STAGE : nullptr, xs());
            M::store_input_grad(L, grad + M::nu * k, ubar + M::c2g * uc);
            lam = sbar;
            M::cost_grad_x_add(L, s, lam);
            s = s_prev;
        }
        wave_sync(); // grad was written by the owner of the input adjoint
        if constexpr (ITER_GLOBAL) {
            for (int j = 0; j < nj; ++j) {
                int i = j * WAVE + L.l;
                if (i < n)
                    grad_out[i] = grad[i];
            }
        }
    }

    // ---- Hessian-vector products: hess_prod_tape(v, Hv) = d/du (grad psi(u)^T v) at the point the tape was
    // recorded at (psi_grad<true>): tangent forward sweep F_t = K d_t, second-order adjoint sweep ld_t = K lF_t +
    // M d_t, the stages' tape rows streaming from HBM into LDS by LDS-DMA one (DEPTH == 1) or two (DEPTH == 2)
    // stages ahead of the stage that computes.  Same arithmetic, same bits. ----
    // ---- DEPTH == 2 -------------------------------------------------------------------------------
    // Two stages ahead.  A wavefront of the one-stage-ahead form keeps 2-5 KiB of tape rows in flight; with a
    // thousand such wavefronts on the chip the memory system runs at bytes-in-flight / latency, far below what
    // it can deliver, and a product takes twice as long as on an empty chip (profiles/r04_product_under_load.txt).
    // Here stage k's rows land in buffer k mod 3 and are requested while stage k -+ 2 computes.  s_waitcnt
    // vmcnt(n) leaves the n youngest vector-memory operations of the wavefront in flight (they complete in
    // order on this target), so each stage start names what may stay behind: the next stage's rows and the
    // stores the previous stage issued after requesting them - FWD_ROW_TRANSFERS, FWD_STAGE_STORES and
    // REV_ROW_TRANSFERS below, each defined next to the code that issues the operations it counts;
    // scripts/check_vmem_counts.py verifies the counts in the compiled code on every path (the Makefile runs it on
    // every library it links and fails the build otherwise).  A smaller n is always safe; -DPANTR_VMCNT_CONSERVATIVE
    // turns every one of these waits into vmcnt(0).
    template <int N_>
    __device__ static __forceinline__ void wait_vmem_but() {
        // expcnt(6) never waits (no exports in a compute kernel) and the compiler never writes it: it MARKS the waits
        // of these sweeps in the machine code, which is how scripts/check_vmem_counts.py tells them from the compiler's
#ifdef PANTR_VMCNT_CONSERVATIVE // build switch: every explicit wait of the sweeps below as a full one (always safe)
        asm volatile("s_waitcnt vmcnt(0) expcnt(6)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(%0) expcnt(6)" ::"n"(N_) : "memory");
#endif
    }
    /// The lanes the tangent sweeps run on (chain_model.hip.h: both substage chains side by side, as in every sweep).
    using TanLane = Lane;
    static constexpr int TAN_ST_STORES = M::PAIR_ST_STORES_PER_STAGE; ///< d_t stores of a forward stage: one per substage pair
    // What the waits of the two-stages-ahead sweeps may leave in flight, from ONE place: the vector-memory operations
    // a stage issues behind its request (the model names what its functions issue).
    static constexpr int FWD_STAGE_STORES  = RECOMP ? 1 : TAN_ST_STORES + 1; ///< + the tangent-state row (xt_store); RECOMP: the state row alone
    static constexpr int FWD_ROW_TRANSFERS = M::FWD_ROW_TRANSFERS;  ///< prefetch_fwd<true>: the K rows of a stage
    static constexpr int REV_ROW_TRANSFERS = M::REV_ROW_TRANSFERS;  ///< prefetch_rev_xt: K, M, d_t rows and the tangent-state row
    /// The lanes of a product, built where the product starts (the lane index passes through an empty asm: the lane
    /// constants of the tangent sweeps - byte offsets into the landing buffers, masks as doubles - are then recomputed per
    /// product, a few dozen integer operations, instead of living in registers, and spilling, through the whole solve).
    __device__ __forceinline__ TanLane tan_lane() const {
        int lane = L.l;
        asm volatile("" : "+v"(lane));
        return TanLane(lane);
    }
    __device__ __forceinline__ const Lane &address_lane(const TanLane &T) const {
        if constexpr (NJ_ >= 4)
            return T;
        else
            return L;
    }
    /// RECOMP: the state row of stage k (the tangent tape's region of the workspace holds these rows instead)
    __device__ __forceinline__ double *srow(int k) const {
        return reinterpret_cast<double *>(st) + (size_t)k * M::S_ROW_DOUBLES;
    }
    __device__ __forceinline__ void hess_prod_tape_d2(const double *v, double *Hv) const {
        wave_sync();
#ifdef PANTR_PHASE_TIMING
        const long long hv_t0_ = clock64();
#endif
        const TanLane T = tan_lane();
        // The lanes the transfers' addresses are formed from: the kernels with four or more elements per lane are
        // short of registers - with the problem's own lanes (L) the compiler keeps `base + 16 lane` of every tape in
        // registers through the whole solve and reloads them from scratch memory inside the stage loops (a wait for
        // every vector-memory operation in flight each time); formed from the product's lanes they live a product long.
        const Lane &A = address_lane(T);
        // Control flow of both sweeps (round 5): a stage body issues the same number of vector-memory operations behind
        // its wait whatever k is - where a forward stage has nothing to request it requests something useful to the
        // adjoint sweep instead, and the adjoint stages that request nothing (1 and 0) are bodies of their own.  The
        // marked waits are then right on every path the machine code's control-flow graph has, not only on the paths the
        // values of k allow: scripts/check_vmem_counts.py takes the minimum over all of them.
        // ---- forward tangent sweep ----
        // The adjoint sweep below starts where this one ends, and its first two stages find their rows in LDS:
        // K from this sweep, M requested a product ahead (prime), the tangent spring vectors and the tangent-state
        // row of stage N - 1 written by the last stage straight into its landing buffer.
        State s_t{0.0, 0.0};
        State s_last{0.0, 0.0}; // RECOMP: the tangent state at the start of the last stage
        // `kind`: 0 = a stage that is not the last, 1 = the last stage of N > 1 stages, 2 = the only stage (N = 1)
        auto stage_fwd = [&](int k, char *cur, char *nxt2, char *prev, auto kind) {
            constexpr bool LAST = decltype(kind)::value != 0, ONLY = decltype(kind)::value == 2;
            // stage k's K rows have landed in `cur`; in flight behind them may be: stage k+1's rows (FWD_ROW_TRANSFERS
            // operations, issued at the start of stage k-1) and the FWD_STAGE_STORES stores stage k-1 issued after them
            if constexpr (!LAST) {
                if (k == 0)
                    wait_vmem_but<FWD_ROW_TRANSFERS>(); // stage 1's rows (a stage that is not the last: N > 1)
                else
                    wait_vmem_but<FWD_ROW_TRANSFERS + FWD_STAGE_STORES>();
            } else if constexpr (ONLY) {
                wait_vmem();
            } else {
                wait_vmem_but<FWD_STAGE_STORES>(); // nothing was requested after its rows but stage k-1's stores
            }
            if constexpr (!LAST) {
                // FWD_ROW_TRANSFERS operations either way: the K rows of stage k + 2, or - stage N - 2, which has none
                // to ask for - the tangent tape and tangent-state row of stage N - 3 (k - 1) for the adjoint sweep, to
                // their place in that stage's buffer, which is `nxt2` (the adjoint sweep asks for them again with the
                // rest of the stage's rows; N = 2: of stage 0, not yet written - nobody reads that buffer)
                static_assert(FWD_ROW_TRANSFERS == 2, "prefetch_st_xt issues two operations, like prefetch_fwd");
                if (k + 2 < N) {
                    M::template prefetch_fwd<true>(A, tape + (size_t)(k + 2) * M::TAPE_D4_PER_STAGE, nxt2);
                } else {
                    const int kk = k > 0 ? k - 1 : 0;
                    if constexpr (RECOMP)
                        M::prefetch_s(A, srow(kk), nxt2);
                    else
                        M::prefetch_st_xt(A, st + (size_t)kk * M::ST_D4_PER_STAGE, Xt + (size_t)kk * M::XT_DOUBLES, nxt2);
                }
            } else {
                // all rows of stage N - 2 for the adjoint sweep's second stage (its K and M rows are in LDS already:
                // they come again, so that the adjoint sweep's first stage, too, has REV_ROW_TRANSFERS operations
                // behind it; N = 1: of stage 0 itself, into a buffer nobody reads)
                const int kk = k > 0 ? k - 1 : 0;
                if constexpr (RECOMP)
                    M::prefetch_rev_s(A, tape + (size_t)kk * M::TAPE_D4_PER_STAGE, srow(kk), prev);
                else
                    M::prefetch_rev_xt(A, tape + (size_t)kk * M::TAPE_D4_PER_STAGE, st + (size_t)kk * M::ST_D4_PER_STAGE,
                                       Xt + (size_t)kk * M::XT_DOUBLES, prev);
            }
            const double uc_t = M::load_input(T, v + M::nu * k);
            double4 *st_k;
            if constexpr (RECOMP) {
                // the state at the stage's start goes to HBM (the last stage's stays in registers: s_last); the d_t of
                // this sweep go nowhere - the landing buffer's tangent-tape area, which the adjoint sweep fills anew
                if constexpr (!LAST)
                    M::s_store(T, srow(k), s_t);
                else
                    s_last = s_t;
                st_k = reinterpret_cast<double4 *>(cur + M::TAPE_STAGE_BYTES);
            } else if constexpr (!LAST) {
                M::xt_store(T, Xt + (size_t)k * M::XT_DOUBLES, s_t);
                st_k = st + (size_t)k * M::ST_D4_PER_STAGE;
            } else {
                M::xt_store(T, reinterpret_cast<double *>(cur + M::LDS_STAGE_BYTES), s_t);
                st_k = reinterpret_cast<double4 *>(cur + M::TAPE_STAGE_BYTES);
            }
            s_t = M::rk4_tan(T, cur, s_t, uc_t, st_k, xs());
        };
        if (!primed)
            prime(A);
        {
            constexpr std::integral_constant<int, 0> F{};
            constexpr std::integral_constant<int, 1> LASTT{};
            const int NL = N - 1; // the last stage by itself: its landing buffer is a constant in each of the three cases
            if (NL == 0) {
                stage_fwd(0, tbuf0, tbuf2, tbuf2, std::integral_constant<int, 2>{});
            } else {
            // three stages per trip, straight through (a lone wavefront pays some twenty cycles for every taken branch:
            // one stage per trip with a three-way branch on k mod 3 cost the memory-bound launches 6-9 %)
            int k = 0;
            do { // (NL >= 1: at least one stage runs before the last one - also in the compiled control flow)
                stage_fwd(k, tbuf0, tbuf2, tbuf2, F);
                if (k + 1 < NL)
                    stage_fwd(k + 1, tbuf1, tbuf0, tbuf0, F);
                if (k + 2 < NL)
                    stage_fwd(k + 2, tbuf2, tbuf1, tbuf1, F);
                k += 3;
            } while (k < NL);
            if (NL % 3 == 0)
                stage_fwd(NL, tbuf0, tbuf2, tbuf2, LASTT);
            else if (NL % 3 == 1)
                stage_fwd(NL, tbuf1, tbuf0, tbuf0, LASTT);
            else
                stage_fwd(NL, tbuf2, tbuf1, tbuf1, LASTT);
            }
        }
        // (nothing to wait for here: a wavefront's LDS accesses execute in order, and so do its vector-memory
        // operations - the rows the adjoint sweep requests were stored by this wavefront at least a stage earlier)
        wave_sync_self();
#ifdef PANTR_PHASE_TIMING
        fwd_cycles += clock64() - hv_t0_;
#endif
        // ---- second-order adjoint sweep: stage k's rows (K, M, the tangent tape and the tangent-state row) in
        // buffer k mod 3; nothing here loads into registers from HBM, so every wait of the sweep is one of the
        // explicit ones below ----
        State lam_t{0.0, 0.0};
        M::cost_grad_x_add_tan(T, T.end ? s_t.x : s_t.v, lam_t);
        bool first_rev = true; // RECOMP: stage N - 1 takes its state from registers
        // `tail`: 0 = a stage k >= 2 (it asks for the rows of stage k - 2), 1 = stage 1, 2 = stage 0 (nothing left to ask
        // for; stage 0 is followed by nothing: a full wait) - stages 1 and 0 always sit in tbuf1 and tbuf0
        auto stage_rev = [&](int k, char *cur, char *nxt2, auto tail) {
            constexpr int TAIL = decltype(tail)::value;
            // stage k's rows have landed; behind them may be stage k-1's: the REV_ROW_TRANSFERS operations the stage
            // before this one issued, the youngest (stage N - 1: those the forward sweep's last stage issued - it needs
            // nothing from them, the wait passes)
            if constexpr (TAIL == 2)
                wait_vmem();
            else
                wait_vmem_but<REV_ROW_TRANSFERS>();
            double xt, uc_t = M::load_input(T, v + M::nu * k);
            State s_k{0.0, 0.0};
            if constexpr (RECOMP) { // the state at the stage's start: registers (stage N - 1) or the row that landed
                const State s_row = M::s_lds(T, cur);
                s_k = {first_rev ? s_last.x : s_row.x, first_rev ? s_last.v : s_row.v};
                xt  = T.end ? s_k.x : s_k.v;
            } else {
                xt = M::xt_lds(T, cur);
            }
            if constexpr (TAIL == 0) { // two stages ahead
                if constexpr (RECOMP)
                    M::prefetch_rev_s(A, tape + (size_t)(k - 2) * M::TAPE_D4_PER_STAGE, srow(k - 2), nxt2);
                else
                    M::prefetch_rev_xt(A, tape + (size_t)(k - 2) * M::TAPE_D4_PER_STAGE, st + (size_t)(k - 2) * M::ST_D4_PER_STAGE,
                                       Xt + (size_t)(k - 2) * M::XT_DOUBLES, nxt2);
            }
            if constexpr (RECOMP) { // the stage's forward tangent once more: its d_t, where the tape would have landed them
                M::rk4_tan(T, cur, s_k, uc_t, reinterpret_cast<double4 *>(cur + M::TAPE_STAGE_BYTES), xs());
                first_rev = false;
            }
            State sbar_t;
            double ubar_t;
            M::rk4_adj_tan(T, cur, lam_t, sbar_t, ubar_t, xs());
            M::store_input_grad(T, Hv + M::nu * k, __builtin_fma(M::c2g, uc_t, ubar_t));
            lam_t = sbar_t;
            M::cost_grad_x_add_tan(T, xt, lam_t);
        };
        {
            constexpr std::integral_constant<int, 0> MID{};
            int k = N - 1;
            // down to k = 2 (mod 3) by hand, then three stages per trip while all three ask for rows (k - 2 >= 2),
            // then stage 2 and the two stages that ask for nothing
            if (k >= 2 && k % 3 == 1) {
                stage_rev(k, tbuf1, tbuf2, MID);
                --k;
            }
            if (k >= 2 && k % 3 == 0) {
                stage_rev(k, tbuf0, tbuf1, MID);
                --k;
            }
            for (; k >= 5; k -= 3) {
                stage_rev(k, tbuf2, tbuf0, MID);
                stage_rev(k - 1, tbuf1, tbuf2, MID);
                stage_rev(k - 2, tbuf0, tbuf1, MID);
            }
            if (k >= 2) // k == 2
                stage_rev(2, tbuf2, tbuf0, MID);
            if (N >= 2)
                stage_rev(1, tbuf1, tbuf2, std::integral_constant<int, 1>{});
            stage_rev(0, tbuf0, tbuf1, std::integral_constant<int, 2>{});
        }
        wave_sync();
        prime(A); // for the next product (the CG algebra runs meanwhile)
        primed = true;
    }
    /// What a product two stages ahead wants in flight before it starts: the M rows of the last two stages (where the
    /// adjoint sweep starts; the oldest requests), then the K rows of stages 0 and 1 (the youngest two: stage 0's
    /// wait leaves them in flight).
    __device__ __forceinline__ void prime(const Lane &A) const {
        // (a branch per buffer, not a selected address: every transfer names its landing buffer as a constant - with
        // a selected one the compiler lost track of what the stage loops' LDS accesses can alias: 960 bytes of scratch)
        auto m_rows = [&](int k) {
            const double4 *rows = tape + (size_t)k * M::TAPE_D4_PER_STAGE;
            if (k % 3 == 0)
                M::template prefetch_m<true>(A, rows, tbuf0);
            else if (k % 3 == 1)
                M::template prefetch_m<true>(A, rows, tbuf1);
            else
                M::template prefetch_m<true>(A, rows, tbuf2);
        };
        m_rows(N - 1);
        if (N > 1)
            m_rows(N - 2);
        asm volatile("" ::: "memory");
        M::template prefetch_fwd<true>(A, tape, tbuf0);
        asm volatile("" ::: "memory");
        if (N > 1)
            M::template prefetch_fwd<true>(A, tape + (size_t)M::TAPE_D4_PER_STAGE, tbuf1);
    }
    // ---- DEPTH == 1 -------------------------------------------------------------------------------
    // One stage ahead: a full wait at every stage start.  The two landing buffers are distinct __shared__ objects
    // and the stage loops are unrolled by two, so every LDS read names a buffer no transfer is writing (otherwise
    // the compiler drains the transfers in flight with vmcnt(0) before each read and the prefetch hides nothing).
    /// What a product one stage ahead wants in flight before it starts (requested by the product before it, so that
    /// the CG algebra in between covers the trip, or at the start of the first product on a new tape): the K rows of
    /// stage 0 and the M rows of stage N - 1 - the adjoint sweep starts at that stage, in the landing buffer the
    /// forward sweep's last stage used: K is there already, M arrives meanwhile (that part of both buffers is idle
    /// during the forward sweep), the tangent spring vectors are written there directly and the tangent-state
    /// component stays in a register.  The first adjoint stage then reads nothing from HBM (it used to wait for
    /// 5 KiB requested only after the forward sweep: two exposed trips per product).
    __device__ __forceinline__ void prime_d1() const {
        M::prefetch_fwd(L, tape, tbuf0);
        if ((N - 1) & 1) // (a branch per buffer, not a selected address: every transfer names its landing buffer)
            M::prefetch_m(L, tape + (size_t)(N - 1) * M::TAPE_D4_PER_STAGE, tbuf1);
        else
            M::prefetch_m(L, tape + (size_t)(N - 1) * M::TAPE_D4_PER_STAGE, tbuf0);
    }
    __device__ __forceinline__ void hess_prod_tape_d1(const double *v, double *Hv) const {
        wave_sync();
#ifdef PANTR_PHASE_TIMING
        const long long hv_t0_ = clock64();
#endif
        const TanLane T = tan_lane();
        // ---- forward tangent sweep ----
        State s_t{0.0, 0.0};
        double xt_last = 0.0; // what xt_store() would write for stage N - 1 (and this lane read back)
        auto stage_fwd = [&](int k, char *cur, char *nxt, auto last) {
            constexpr bool LAST = decltype(last)::value;
            wait_vmem(); // stage k's rows have landed in `cur`
            // the stage input is read BEFORE the next stage's transfers are requested: an LDS read behind an LDS-DMA
            // request gets a full s_waitcnt vmcnt(0) from the compiler whenever it cannot tell the vector from the
            // landing buffers - the request just made would be waited for, stage after stage (+13 % per product when
            // it happened; scripts/check_vmem_counts.py looks for that pattern in the compiled kernels)
            const double uc_t = M::load_input(T, v + M::nu * k);
            if constexpr (!LAST) {
                M::prefetch_fwd(L, tape + (size_t)(k + 1) * M::TAPE_D4_PER_STAGE, nxt);
                M::xt_store(T, Xt + (size_t)k * M::XT_DOUBLES, s_t);
            } else {
                xt_last = T.end ? s_t.x : s_t.v;
            }
            // the last stage writes its tangent spring vectors where the adjoint sweep's first stage reads them
            double4 *st_k = LAST ? reinterpret_cast<double4 *>(cur + M::TAPE_STAGE_BYTES) : st + (size_t)k * M::ST_D4_PER_STAGE;
            s_t = M::rk4_tan(T, cur, s_t, uc_t, st_k, xs());
        };
        if (!primed)
            prime_d1();
        {
            constexpr std::false_type F{};
            constexpr std::true_type T{};
            int k = 0;
            for (; k + 2 < N; k += 2) {
                stage_fwd(k, tbuf0, tbuf1, F);
                stage_fwd(k + 1, tbuf1, tbuf0, F);
            }
            if (k + 2 == N) {
                stage_fwd(k, tbuf0, tbuf1, F);
                stage_fwd(k + 1, tbuf1, tbuf0, T);
            } else {
                stage_fwd(k, tbuf0, tbuf1, T);
            }
        }
        // (LDS accesses of a wavefront execute in order, and its vector-memory operations do: the tangent tape rows
        // the adjoint sweep requests were stored at least a stage earlier by this wavefront - no wait for them here)
        wave_sync_self();
#ifdef PANTR_PHASE_TIMING
        fwd_cycles += clock64() - hv_t0_;
#endif
        // ---- second-order adjoint sweep (even stages land in tbuf0, odd ones in tbuf1) ----
        State lam_t{0.0, 0.0};
        M::cost_grad_x_add_tan(T, T.end ? s_t.x : s_t.v, lam_t);
        double xt = 0.0, xt_n = xt_last;
        bool first = true; // stage N - 1 finds its rows in LDS
        auto stage_rev = [&](int k, char *cur, char *nxt) {
            if (!first)
                wait_vmem(); // stage k's rows and Xt row have landed
            first = false;
            xt = xt_n;
            double uc_t = M::load_input(T, v + M::nu * k);
            if (k > 0) { // one stage ahead
                M::prefetch_rev(L, tape + (size_t)(k - 1) * M::TAPE_D4_PER_STAGE,
                                st + (size_t)(k - 1) * M::ST_D4_PER_STAGE, nxt);
                xt_n = M::xt_load(T, Xt + (size_t)(k - 1) * M::XT_DOUBLES);
            }
            State sbar_t;
            double ubar_t;
            M::rk4_adj_tan(T, cur, lam_t, sbar_t, ubar_t, xs());
            M::store_input_grad(T, Hv + M::nu * k, __builtin_fma(M::c2g, uc_t, ubar_t));
            lam_t = sbar_t;
            M::cost_grad_x_add_tan(T, xt, lam_t);
        };
        int k = N - 1;
        if (!(k & 1)) {
            stage_rev(k, tbuf0, tbuf1);
            --k;
        }
        for (; k >= 1; k -= 2) {
            stage_rev(k, tbuf1, tbuf0);
            stage_rev(k - 1, tbuf0, tbuf1);
        }
        wave_sync();
        prime_d1(); // for the next product (the CG algebra runs meanwhile)
        primed = true;
    }
    __device__ __forceinline__ void hess_prod_tape(const double *v, double *Hv) const {
        if constexpr (DEPTH == 2)
            hess_prod_tape_d2(v, Hv);
        else
            hess_prod_tape_d1(v, Hv);
    }

    // ---- vector helpers over the LDS vectors: lane l owns i = l, l+64, ... ------------------
    using OV = Owned<(NJ_ > 0 ? NJ_ : 1)>;
    __device__ __forceinline__ double dot(const double *a, const double *b) const {
        double acc = 0.0;
        if constexpr (NJ_ > 0) {
            const OV av = OV::load(a, L.l), bv = OV::load(b, L.l);
#pragma unroll
            for (int j = 0; j < NJ_; ++j)
                acc = acc + av[j] * bv[j];
        } else {
            for (int j = 0; j < nj; ++j) {
                int i = j * WAVE + L.l;
                if (i < n)
                    acc = acc + a[i] * b[i];
            }
        }
        return uniform(wave_sum(acc));
    }
    /// dot(a, mask ? b : 0)
    __device__ __forceinline__ double dot_masked(const double *a, const double *b, uint32_t mask) const {
        double acc = 0.0;
        if constexpr (NJ_ > 0) {
            const OV av = OV::load(a, L.l), bv = OV::load(b, L.l);
#pragma unroll
            for (int j = 0; j < NJ_; ++j)
                acc = acc + av[j] * ((mask >> j) & 1u ? bv[j] : 0.0);
        } else {
            for (int j = 0; j < nj; ++j) {
                int i = j * WAVE + L.l;
                if (i < n)
                    acc = acc + a[i] * ((mask >> j) & 1u ? b[i] : 0.0);
            }
        }
        return uniform(wave_sum(acc));
    }
    __device__ __forceinline__ double norm_inf(const double *a) const {
        double r = 0.0;
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + L.l;
            if (i < n)
                r = fmax(r, fabs(a[i]));
        }
        return uniform(wave_max(r));
    }

    /// Box prox-gradient step: p = min(max(-gamma grad, lb - x), ub - x), xhat = x + p.
    /// Returns p'p and grad'p.
    __device__ __forceinline__ void prox_grad_step(double gamma, const double *x, const double *grad,
                                                   double *xh, double *p, double &pTp,
                                                   double &gradTp) const {
        double a1 = 0.0, a2 = 0.0;
        if constexpr (NJ_ > 0) {
            const OV xv = OV::load(x, L.l), gv = OV::load(grad, L.l), lo = OV::load(lbv, L.l), hi = OV::load(ubv, L.l);
            OV pv, xhv;
#pragma unroll
            for (int j = 0; j < NJ_; ++j) { // (padding: x = grad = lb = ub = 0, so p = 0 and xhat = 0)
                double t = (-gamma) * gv[j];
                pv[j]    = fmin(fmax(t, lo[j] - xv[j]), hi[j] - xv[j]);
                xhv[j]   = xv[j] + pv[j];
                a1       = a1 + pv[j] * pv[j];
                a2       = a2 + pv[j] * gv[j];
            }
            pv.store(p, L.l);
            xhv.store(xh, L.l);
            wave_sum2(a1, a2);
            pTp    = uniform(a1);
            gradTp = uniform(a2);
            return;
        }
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + L.l;
            if (i < n) {
                int c     = i % M::nu;
                double xi = x[i], gi = grad[i];
                double t  = (-gamma) * gi;
                double pi = fmin(fmax(t, box_pick(box.u_lb, c) - xi), box_pick(box.u_ub, c) - xi);
                p[i]      = pi;
                xh[i]     = xi + pi;
                a1        = a1 + pi * pi;
                a2        = a2 + pi * gi;
            }
        }
        pTp    = uniform(wave_sum(a1));
        gradTp = uniform(wave_sum(a2));
    }

    /// || Pi_C(x - grad) - x ||_inf  (ProjGradUnitNorm)
    __device__ __forceinline__ double proj_grad_unit_norm(const double *x, const double *grad) const {
        double r = 0.0;
        if constexpr (NJ_ > 0) {
            const OV xv = OV::load(x, L.l), gv = OV::load(grad, L.l), lo = OV::load(lbv, L.l), hi = OV::load(ubv, L.l);
#pragma unroll
            for (int j = 0; j < NJ_; ++j) {
                double t  = (-1.0) * gv[j];
                double pi = fmin(fmax(t, lo[j] - xv[j]), hi[j] - xv[j]);
                r         = fmax(r, fabs(pi));
            }
            return uniform(wave_max(r));
        }
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + L.l;
            if (i < n) {
                int c     = i % M::nu;
                double xi = x[i];
                double t  = (-1.0) * grad[i];
                double pi = fmin(fmax(t, box_pick(box.u_lb, c) - xi), box_pick(box.u_ub, c) - xi);
                r         = fmax(r, fabs(pi));
            }
        }
        return uniform(wave_max(r));
    }

    /// Inactive set J = { i : lb < x_i - gamma grad_i < ub } as one bit per owned element.
    __device__ __forceinline__ uint32_t inactive_mask(double gamma, const double *x, const double *grad,
                                                      int &nJ) const {
        uint32_t mask = 0;
        int cnt       = 0;
        if constexpr (NJ_ > 0) {
            const OV xv = OV::load(x, L.l), gv = OV::load(grad, L.l), lo = OV::load(lbv, L.l), hi = OV::load(ubv, L.l);
#pragma unroll
            for (int j = 0; j < NJ_; ++j) { // (padding: 0 < 0 is false)
                double xfw = xv[j] - gamma * gv[j];
                bool in    = lo[j] < xfw && xfw < hi[j];
                mask |= (in ? 1u : 0u) << j;
                cnt += __builtin_popcountll(__ballot(in ? 1 : 0));
            }
            nJ = cnt;
            return mask;
        }
        for (int j = 0; j < nj; ++j) {
            int i   = j * WAVE + L.l;
            bool in = false;
            if (i < n) {
                int c      = i % M::nu;
                double xfw = x[i] - gamma * grad[i];
                in = box_pick(box.u_lb, c) < xfw && xfw < box_pick(box.u_ub, c);
            }
            mask |= (in ? 1u : 0u) << j;
            cnt += __builtin_popcountll(__ballot(in ? 1 : 0)); // wave-uniform count, no reduction needed
        }
        nJ = cnt;
        return mask;
    }
};

struct IterateRef {
    int base; ///< first of 4 consecutive vector slots: x, xhat, grad, p
    int yh;   ///< index of this iterate's yhat m-vector (problems with general constraints)
    double psi_x = __builtin_nan(""), psi_xh = __builtin_nan(""), gamma = __builtin_nan(""),
           L = __builtin_nan(""), pTp = __builtin_nan(""), gradTp = __builtin_nan("");
    __device__ __forceinline__ double fbe() const { return (psi_x + pTp / (2.0 * gamma)) + gradTp; }
};

__device__ __forceinline__ void swap_iter(IterateRef &a, IterateRef &b) {
    IterateRef t = a;
    a            = b;
    b            = t;
}

__device__ __forceinline__ bool is_finite(double v) { return fabs(v) < __builtin_huge_val(); }

// Profiling build (-DPANTR_PHASE_TIMING; `make phase` builds it as a separate library, never over
// the product): stats.phase_s[] receives the seconds spent per phase (scripts/phase_timing.sh);
// all documented stats fields keep their meaning and pantr_version() reports the build.
#if defined(PANTR_REST_SPLIT)
// Third profiling variant (`make phase EXTRA_HIPFLAGS="-DPANTR_REST_TIMING -DPANTR_REST_SPLIT=<base>"`,
// scripts/rest_timing.py): the time outside the evaluations in sixteen pieces, four per build - phase_s[i]
// receives piece <base> + i.  Pieces: 0 iterate swap + the wait for the helper's result, 1 bound test, 2 progress
// check, 3 stop criterion; 4 prox copy (+ adoption of the helper's gradient), 5 prox step, 6 request to the helper,
// 7 inactive set + right-hand side; 8 CG start, 9 first CG pass + d'Bd, 10 second CG pass, 11 the rest of a CG
// iteration (and the model value); 12 full step, 13 candidate, 14 its prox step (+ the deferred model value),
// 15 ratio test and radius.
#define PANTR_FINE(k)                                                                              \
    do {                                                                                           \
        long long t_ = wall_clock64();                                                             \
        const int b_ = rest_cur_ - (PANTR_REST_SPLIT);                                             \
        if (b_ >= 0 && b_ < 4)                                                                     \
            rest_bucket_[b_] += (double)(t_ - rest_last_) * P.tick_s;                              \
        rest_last_ = t_;                                                                           \
        rest_cur_  = (k);                                                                          \
    } while (0)
#define PANTR_REGION(k) do {} while (0)
#define PANTR_PHASE(idx, ...)                                                                      \
    do {                                                                                           \
        const int keep_ = rest_cur_;                                                               \
        PANTR_FINE(-100);                                                                          \
        __VA_ARGS__;                                                                               \
        PANTR_FINE(keep_);                                                                         \
    } while (0)
#elif defined(PANTR_REST_TIMING)
// Second profiling variant (`make phase EXTRA_HIPFLAGS=-DPANTR_REST_TIMING`, scripts/rest_timing.py;
// pantr_version() says "+rest-timing"): where a PANTR iteration's time OUTSIDE the evaluations goes.
// phase_s[] then means: [0] all evaluations, [1] from the end of the CG loop to the stop criterion of the
// next iteration, [2] prox iterate .. start of the CG loop, [3] the CG loop without its products.
#define PANTR_FINE(k) do {} while (0)
#define PANTR_REGION(k)                                                                            \
    do {                                                                                           \
        long long t_ = wall_clock64();                                                             \
        rest_bucket_[rest_cur_] += (double)(t_ - rest_last_) * P.tick_s;                           \
        rest_last_ = t_;                                                                           \
        rest_cur_  = (k);                                                                          \
    } while (0)
#define PANTR_PHASE(idx, ...)                                                                      \
    do {                                                                                           \
        const int keep_ = rest_cur_;                                                               \
        PANTR_REGION(0);                                                                           \
        __VA_ARGS__;                                                                               \
        PANTR_REGION(keep_);                                                                       \
    } while (0)
#elif defined(PANTR_PHASE_TIMING)
#define PANTR_FINE(k) do {} while (0)
#define PANTR_REGION(k) do {} while (0)
#define PANTR_PHASE(idx, ...)                                                                      \
    do {                                                                                           \
        long long t0_ = wall_clock64();                                                            \
        __VA_ARGS__;                                                                               \
        st.phase_s[idx] += (double)(wall_clock64() - t0_) * P.tick_s;                              \
    } while (0)
#else
#define PANTR_FINE(k) do {} while (0)
#define PANTR_REGION(k) do {} while (0)
#define PANTR_PHASE(idx, ...)                                                                    \
    do {                                                                                           \
        __VA_ARGS__;                                                                               \
    } while (0)
#endif

// ---------------------------------------------------------------------------------------
// The PANTR inner solver, generic over the wave-cooperative problem `PB`
// ---------------------------------------------------------------------------------------
struct InnerResult {
    int status, iterations;
    double eps, gamma, psi, fbe, radius;
};

/// One inner solve: x_io (an LDS n-vector) holds the initial guess on entry and the solution on
/// exit; multipliers/penalties live in the problem.  deadline_s counts from t_start.
template <class PB>
__device__ __forceinline__ InnerResult pantr_inner(PB &pb, const KernelParams &P, double tolerance,
                                                   double deadline_s, long long t_start, double *x_io,
                                                   DeviceStats &st, double *trace, int trace_cap) {
    const int lane = threadIdx.x & (WAVE - 1); // (two-wavefront workgroups: speculation)
    const int n = pb.n, nj = pb.nj;
    // PB::NJ > 0: every lane owns exactly NJ elements of every vector (padding included, kept at zero) and the
    // passes below are straight-line code on them (Owned<NJ>); otherwise the generic loops over nj
    constexpr int NJ = PB::NJ;
    using OV = Owned<(NJ > 0 ? NJ : 1)>;
    // iterate vectors (pb.ivec: HBM or LDS, the problem decides) and the CG block in LDS
    enum { S_IT0 = 0, S_IT1 = 4, S_IT2 = 8, S_Q = PB::CG_BASE + 0, S_WORK = PB::CG_BASE + 1, S_RJ = PB::CG_BASE + 2,
           S_Z = PB::CG_BASE + 3, S_R = PB::CG_BASE + 4, S_D = PB::CG_BASE + 5, S_STEP = PB::XIO_SLOT };
    IterateRef curr{S_IT0, 0}, prox{S_IT1, 1}, cand{S_IT2, 2};
    auto X_  = [&](const IterateRef &it) { return pb.ivec(it.base + 0); };
    auto XH_ = [&](const IterateRef &it) { return pb.ivec(it.base + 1); };
    auto G_  = [&](const IterateRef &it) { return pb.ivec(it.base + 2); };
    auto P_  = [&](const IterateRef &it) { return pb.ivec(it.base + 3); };

    int status    = Busy;
    double eps_k  = __builtin_nan("");
    double radius = __builtin_nan("");

    // ---- load the initial guess ----
    for (int j = 0; j < nj; ++j) {
        int i = j * WAVE + lane;
        if (i < n)
            X_(curr)[i] = x_io[i];
    }

    // ---- Lipschitz estimate: finite difference of the gradient along the gradient ----
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
        pb.eval_psi_grad(XH_(curr), G_(cand));
        ++st.n_grad_psi;
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + lane;
            if (i < n)
                h[i] = G_(cand)[i] - G_(curr)[i];
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
        radius = P.initial_radius;
        if (!is_finite(radius) || radius == 0)
            radius = 0.1 * __builtin_sqrt(pb.dot(G_(curr), G_(curr)));
        radius = fmax(radius, P.min_radius);
    }

    // `which` names the iterate whose psi(xhat) and quadratic-upper-bound backtracking are
    // pending: 0 = the initial iterate, 1 = the accepted candidate, 2 = the prox iterate.
    int which = 0;
#ifdef PANTR_REST_TIMING
    double rest_bucket_[4] = {0, 0, 0, 0};
    long long rest_last_ = wall_clock64();
    int rest_cur_ = 1;
#endif
    while (status == Busy) {
        PANTR_FINE(0);
        // speculation (PB::HAS_SPEC, two-wavefront workgroups): psi(xhat), the bound test, grad_L(xhat)
        // and the tape of a rejected candidate's successor were evaluated by the helper wavefront
        // during the CG loop; taken here, dropped if the candidate was accepted
        bool have_spec  = false;
        double spec_psi = 0.0;
        if (which == 1) {
            swap_iter(curr, cand);
            if constexpr (PB::HAS_SPEC)
                pb.spec_drop();
        } else if (which == 2) {
            swap_iter(curr, prox);
            if constexpr (PB::HAS_SPEC)
                have_spec = pb.spec_take(spec_psi);
        }
        // ---- psi(xhat) + quadratic upper bound backtracking on the new current iterate ----
        PANTR_FINE(1);
        while (true) {
            // keeps the trajectory (and yhat) for grad_L(xhat)
            if (have_spec)
                curr.psi_xh = spec_psi; // (the helper made the test below with the same numbers: it holds)
            else
                PANTR_PHASE(PH_PSI_XHAT, curr.psi_xh = pb.eval_psi_xhat(XH_(curr), curr.yh));
            ++st.n_psi;
            double margin = (1.0 + fabs(curr.psi_x)) * P.quadratic_upperbound_tolerance_factor;
            bool violated = curr.psi_xh > ((curr.psi_x + curr.gradTp) + (0.5 * curr.L) * curr.pTp) + margin;
            if (!(curr.L < P.L_max && violated))
                break;
            have_spec = false;
            curr.gamma = curr.gamma / 2.0;
            curr.L     = curr.L * 2.0;
            pb.prox_grad_step(curr.gamma, X_(curr), G_(curr), XH_(curr), P_(curr), curr.pTp,
                              curr.gradTp);
            ++st.n_prox_grad_step;
            ++st.qub_backtracks;
        }
        PANTR_FINE(2);
        if (which != 0) {
            // did x move at all?  (the previous iterate now sits in cand or prox)
            const double *xo = pb.ivec(which == 1 ? cand.base : prox.base);
            const double *xn = X_(curr);
            bool same = true;
            if constexpr (NJ > 0) {
                const OV a = OV::load(xn, lane), b = OV::load(xo, lane);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    same = same && a[j] == b[j];
            } else {
                for (int j = 0; j < nj; ++j) {
                    int i = j * WAVE + lane;
                    if (i < n)
                        same = same && xn[i] == xo[i];
                }
            }
            bool all_same = __builtin_amdgcn_readfirstlane(__all(same ? 1 : 0)) != 0;
            no_progress   = all_same ? no_progress + 1 : 0;
            ++k;
        }

        // ---- stop criterion ----
        PANTR_FINE(3);
        if (P.stop_crit == ProjGradUnitNorm) {
            eps_k = pb.proj_grad_unit_norm(X_(curr), G_(curr));
            ++st.n_prox_grad_step;
        } else if (P.stop_crit == FPRNorm) {
            eps_k = pb.norm_inf(P_(curr)) / curr.gamma;
        } else {
            eps_k = __builtin_nan("");
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

        PANTR_REGION(2);
        PANTR_FINE(4);
        // ---- forward-backward step: prox iterate xbar = xhat_k, grad via grad_L(xhat) ----
        if constexpr (NJ > 0) {
            OV::load(XH_(curr), lane).store(X_(prox), lane);
        } else {
            for (int j = 0; j < nj; ++j) {
                int i = j * WAVE + lane;
                if (i < n)
                    X_(prox)[i] = XH_(curr)[i];
            }
        }
        prox.psi_x = curr.psi_xh;
        // grad_L(xhat): xbar = xhat_k was the last point psi was evaluated at (trajectory X is still
        // in the workspace), so only the adjoint sweep runs; it records the tape at xbar
        if (have_spec) {
            if constexpr (PB::HAS_SPEC)
                pb.spec_adopt(G_(prox)); // the helper's gradient, the tape it recorded
        } else {
            PANTR_PHASE(PH_GRAD_XHAT, pb.eval_grad_xhat(X_(prox), G_(prox), curr.yh));
        }
        ++st.n_grad_L;
        prox.gamma = curr.gamma;
        prox.L     = curr.L;
        PANTR_FINE(5);
        pb.prox_grad_step(prox.gamma, X_(prox), G_(prox), XH_(prox), P_(prox), prox.pTp, prox.gradTp);
        ++st.n_prox_grad_step;
        PANTR_FINE(6);
        if constexpr (PB::HAS_SPEC)
            if (!P.disable_acceleration) // should the candidate be rejected: xbar's own next steps, meanwhile
                pb.spec_request(prox.base + 1, prox.psi_x, prox.gradTp, prox.pTp, prox.L, P, prox.yh);

        // ---- Newton trust-region direction at xbar (Steihaug-Toint CG on the inactive set) ----
        PANTR_FINE(7);
        bool accept    = false, model_deferred = false;
        double rho     = __builtin_nan(""), q_model = __builtin_nan("");
        int nJ = 0, cg_its = 0;
        if (!P.disable_acceleration) {
            double *q = pb.vec(S_Q), *work = pb.vec(S_WORK), *rJ = pb.vec(S_RJ), *z = pb.vec(S_Z),
                   *r = pb.vec(S_R), *d = pb.vec(S_D), *step = pb.vec(S_STEP);
            const uint32_t mask = pb.inactive_mask(prox.gamma, X_(prox), G_(prox), nJ);
            const double inv_gamma = -1.0 / prox.gamma;
            bool qK_zero = true;
            if constexpr (NJ > 0) {
                const OV pp = OV::load(P_(prox), lane);
                OV rj, qq;
#pragma unroll
                for (int j = 0; j < NJ; ++j) { // (padding: outside J, p = 0)
                    bool in = (mask >> j) & 1u;
                    rj[j]   = in ? inv_gamma * pp[j] : 0.0;
                    qq[j]   = in ? 0.0 : pp[j];
                    qK_zero = qK_zero && (in || pp[j] == 0.0);
                }
                rj.store(rJ, lane);
                qq.store(q, lane);
            } else {
                for (int j = 0; j < nj; ++j) {
                    int i = j * WAVE + lane;
                    if (i < n) {
                        bool in   = (mask >> j) & 1u;
                        double pi = P_(prox)[i];
                        rJ[i]     = in ? inv_gamma * pi : 0.0;
                        q[i]      = in ? 0.0 : pi;
                        qK_zero   = qK_zero && (in || pi == 0.0);
                    }
                }
            }
            // H q_K is not evaluated when q_K = 0 (active bounds the prox step does not leave - nine
            // iterations in ten of the hanging-chain benchmark): the product is exactly zero
            const bool skip_kterm = __builtin_amdgcn_readfirstlane(__all(qK_zero ? 1 : 0)) != 0;
            enum { KTERM, CGSTART, CG, EVAL_A, EVAL_B, EVAL_STEP };
            // PB::HAS_SPEC: the model value of the CG's step is the one product of an iteration whose
            // result nothing needs before the ratio test - the helper wavefront evaluates it
            // (spec_post_model) while this one goes on to the candidate; `model_deferred`: q_model is
            // still on its way
            auto defer_model = [&]() {
                if constexpr (PB::HAS_SPEC)
                    return pb.spec_post_model(step, rJ, mask);
                return false;
            };
            int state = (P.hessian_vec_factor != 0 && nJ < n && !skip_kterm) ? KTERM : CGSTART;
            if (P.hessian_vec_factor != 0 && nJ < n && skip_kterm)
                ++st.n_hess_psi_prod_calls; // alpaqa would call hess_psi_prod here (and get zero)
            const double *vin = q;
            double r_sq = 0, tolerance = 0, qa = 0;
            int cg_i = 0;
            const int cg_max = (int)lround((double)nJ * P.max_iter_factor);
            PANTR_REGION(3);
            while (true) {
                if (state == CGSTART) {
                    PANTR_FINE(8);
                    // all 64 nj slots: the CG passes below run without bounds checks, the padding of
                    // every vector they touch is zero and stays zero (rJ's and work's are never written)
                    if constexpr (NJ > 0) {
                        const OV g = OV::load(rJ, lane);
                        OV zz, dd;
                        double acc = 0.0;
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            zz[j] = 0.0;
                            dd[j] = -g[j];
                            acc   = acc + g[j] * g[j];
                        }
                        zz.store(z, lane);
                        g.store(r, lane);
                        dd.store(d, lane);
                        r_sq = uniform(wave_sum(acc));
                    } else {
                        for (int j = 0; j < nj; ++j) {
                            int i    = j * WAVE + lane;
                            double g = rJ[i];
                            z[i]     = 0.0;
                            r[i]     = g;
                            d[i]     = -g;
                        }
                        r_sq = pb.dot(r, r);
                    }
                    double grad_mag = __builtin_sqrt(r_sq);
                    if (!(grad_mag > 0.0)) {
                        for (int j = 0; j < nj; ++j) {
                            int i = j * WAVE + lane;
                            if (i < n)
                                step[i] = 0.0;
                        }
                        q_model = grad_mag == 0.0 ? 0.0 : __builtin_nan("");
                        break;
                    }
                    double rooted = P.tol_scale_root == 0.5 ? __builtin_sqrt(grad_mag)
                                                            : pow(grad_mag, P.tol_scale_root);
                    tolerance = fmin(P.tol_max, (P.tol_scale * grad_mag) * fmin(1.0, rooted));
                    state     = CG;
                    vin       = d;
                }
                // the one Hessian-vector product site: work = H(xbar) * vin (tape replay)
                PANTR_PHASE(PH_DIRECTION, pb.hess_prod(X_(prox), vin, work));
                ++st.n_hess_psi_prod;
                ++st.n_hess_psi_prod_calls;
                if (state == KTERM) {
                    if constexpr (NJ > 0) {
                        OV rj = OV::load(rJ, lane);
                        const OV w = OV::load(work, lane);
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            rj[j] = (mask >> j) & 1u ? rj[j] + P.hessian_vec_factor * w[j] : rj[j];
                        rj.store(rJ, lane);
                    } else {
                        for (int j = 0; j < nj; ++j) {
                            int i = j * WAVE + lane;
                            if (i < n && ((mask >> j) & 1u))
                                rJ[i] = rJ[i] + P.hessian_vec_factor * work[i];
                        }
                    }
                    state = CGSTART;
                    continue;
                }
                if (state == CG) {
                    // One CG iteration = three passes over the vectors (each lane its nj elements) and
                    // three reductions, two of them side by side.  Bd = (H d) restricted to J is formed
                    // on the fly from `work`; the arithmetic is that of
                    //   Bd = mask(work); dBd = d'Bd; step = z + alpha d; |step|^2; r += alpha Bd; |r|^2;
                    //   d = -r + beta d; z = step
                    // element by element and sum by sum.
                    ++cg_its;
                    if constexpr (NJ > 0) {
                        // everything a CG iteration reads, at once; its three passes work on registers
                        const OV dv = OV::load(d, lane), wv = OV::load(work, lane), zv = OV::load(z, lane),
                                 rv = OV::load(r, lane);
                        PANTR_FINE(9);
                        OV bv;
                        double acc = 0.0;
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            bv[j] = (mask >> j) & 1u ? wv[j] : 0.0;
                            acc   = acc + dv[j] * bv[j];
                        }
                        double dBd = uniform(wave_sum(acc));
                        bool to_boundary = false, neg_curv = dBd <= 0.0;
                        double alpha = 0, r_next_sq = 0;
                        PANTR_FINE(10);
                        OV sv, rn;
                        if (!neg_curv) {
                            alpha = r_sq / dBd;
                            double a2 = 0.0, a3 = 0.0;
#pragma unroll
                            for (int j = 0; j < NJ; ++j) {
                                sv[j] = zv[j] + alpha * dv[j];
                                a2    = a2 + sv[j] * sv[j];
                                rn[j] = rv[j] + alpha * bv[j]; // unused if the step reaches the boundary
                                a3    = a3 + rn[j] * rn[j];
                            }
                            sv.store(step, lane);
                            rn.store(r, lane);
                            wave_sum2(a2, a3);
                            to_boundary = __builtin_sqrt(uniform(a2)) >= radius;
                            r_next_sq   = uniform(a3);
                        }
                        PANTR_FINE(11);
                        if (neg_curv || to_boundary) {
                            // ||z + t d|| = radius
                            double zz = 0.0, zd = 0.0, dd = 0.0;
#pragma unroll
                            for (int j = 0; j < NJ; ++j) {
                                zz = zz + zv[j] * zv[j];
                                zd = zd + zv[j] * dv[j];
                                dd = dd + dv[j] * dv[j];
                            }
                            wave_sum3(zz, zd, dd);
                            zz = uniform(zz), zd = uniform(zd), dd = uniform(dd);
                            double qa_ = dd, qb_ = 2.0 * zd, qc_ = zz - radius * radius;
                            double sqd = __builtin_sqrt(qb_ * qb_ - (4.0 * qa_) * qc_);
                            double aux = qb_ + copysign(sqd, qb_);
                            double t1 = (-aux) / (2.0 * qa_), t2 = (-(2.0 * qc_)) / aux;
                            double ta = fmin(t1, t2), tb = fmax(t1, t2);
                            // (the padding stays zero whatever ta and tb are)
                            if (neg_curv) {
                                OV ra, db;
#pragma unroll
                                for (int j = 0; j < NJ; ++j) {
                                    const bool own = j * WAVE + lane < n;
                                    ra[j] = own ? zv[j] + ta * dv[j] : 0.0;
                                    db[j] = own ? zv[j] + tb * dv[j] : 0.0;
                                }
                                ra.store(r, lane);
                                db.store(d, lane);
                                state = EVAL_A;
                                vin   = r;
                            } else {
                                OV sb;
#pragma unroll
                                for (int j = 0; j < NJ; ++j)
                                    sb[j] = j * WAVE + lane < n ? zv[j] + tb * dv[j] : 0.0;
                                sb.store(step, lane);
                                if ((model_deferred = defer_model())) {
                                    ++st.n_hess_psi_prod;
                                    ++st.n_hess_psi_prod_calls;
                                    break;
                                }
                                state = EVAL_STEP;
                                vin   = step;
                            }
                            continue;
                        }
                        if (__builtin_sqrt(r_next_sq) < tolerance || cg_i > cg_max) {
                            if ((model_deferred = defer_model())) {
                                ++st.n_hess_psi_prod;
                                ++st.n_hess_psi_prod_calls;
                                break;
                            }
                            state = EVAL_STEP;
                            vin   = step;
                            continue;
                        }
                        double beta = r_next_sq / r_sq;
                        r_sq        = r_next_sq;
                        OV dn;
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            dn[j] = (-rn[j]) + beta * dv[j];
                        dn.store(d, lane);
                        sv.store(z, lane);
                        ++cg_i;
                        vin = d;
                        continue;
                    } else {
                        PANTR_FINE(9);
                        double acc = 0.0;
                        for (int j = 0; j < nj; ++j) {
                            int i    = j * WAVE + lane;
                            double b = (mask >> j) & 1u ? work[i] : 0.0;
                            acc      = acc + d[i] * b;
                        }
                        double dBd = uniform(wave_sum(acc));
                        bool to_boundary = false, neg_curv = dBd <= 0.0;
                        double alpha = 0, r_next_sq = 0;
                        PANTR_FINE(10);
                        if (!neg_curv) {
                            alpha = r_sq / dBd;
                            double a2 = 0.0, a3 = 0.0;
                            for (int j = 0; j < nj; ++j) {
                                int i     = j * WAVE + lane;
                                double si = z[i] + alpha * d[i];
                                step[i]   = si;
                                a2        = a2 + si * si;
                                double b  = (mask >> j) & 1u ? work[i] : 0.0;
                                double ri = r[i] + alpha * b; // unused if the step reaches the boundary
                                r[i]      = ri;
                                a3        = a3 + ri * ri;
                            }
                            a2          = wave_sum(a2);
                            a3          = wave_sum(a3);
                            to_boundary = __builtin_sqrt(uniform(a2)) >= radius;
                            r_next_sq   = uniform(a3);
                        }
                        PANTR_FINE(11);
                        if (neg_curv || to_boundary) {
                            // ||z + t d|| = radius
                            double zz = pb.dot(z, z), zd = pb.dot(z, d), dd = pb.dot(d, d);
                            double qa_ = dd, qb_ = 2.0 * zd, qc_ = zz - radius * radius;
                            double sqd = __builtin_sqrt(qb_ * qb_ - (4.0 * qa_) * qc_);
                            double aux = qb_ + copysign(sqd, qb_);
                            double t1 = (-aux) / (2.0 * qa_), t2 = (-(2.0 * qc_)) / aux;
                            double ta = fmin(t1, t2), tb = fmax(t1, t2);
                            if (neg_curv) {
                                for (int j = 0; j < nj; ++j) {
                                    int i = j * WAVE + lane;
                                    if (i < n) {
                                        double zi = z[i], di = d[i];
                                        r[i] = zi + ta * di;
                                        d[i] = zi + tb * di;
                                    }
                                }
                                state = EVAL_A;
                                vin   = r;
                            } else {
                                for (int j = 0; j < nj; ++j) {
                                    int i = j * WAVE + lane;
                                    if (i < n)
                                        step[i] = z[i] + tb * d[i];
                                }
                                if ((model_deferred = defer_model())) {
                                    ++st.n_hess_psi_prod;
                                    ++st.n_hess_psi_prod_calls;
                                    break;
                                }
                                state = EVAL_STEP;
                                vin   = step;
                            }
                            continue;
                        }
                        if (__builtin_sqrt(r_next_sq) < tolerance || cg_i > cg_max) {
                            if ((model_deferred = defer_model())) {
                                ++st.n_hess_psi_prod;
                                ++st.n_hess_psi_prod_calls;
                                break;
                            }
                            state = EVAL_STEP;
                            vin   = step;
                            continue;
                        }
                        double beta = r_next_sq / r_sq;
                        r_sq        = r_next_sq;
                        for (int j = 0; j < nj; ++j) {
                            int i = j * WAVE + lane;
                            d[i]  = (-r[i]) + beta * d[i];
                            z[i]  = step[i];
                        }
                        ++cg_i;
                        vin = d;
                        continue;
                    }
                }
                // model value m(p) = p'g + 1/2 p'Hp of the vector just multiplied
                double mval;
                if constexpr (NJ > 0) {
                    const OV vv = OV::load(vin, lane), gv = OV::load(rJ, lane), wv = OV::load(work, lane);
                    double a1 = 0.0, a2 = 0.0;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        a1 = a1 + vv[j] * gv[j];
                        a2 = a2 + vv[j] * ((mask >> j) & 1u ? wv[j] : 0.0);
                    }
                    wave_sum2(a1, a2);
                    mval = uniform(a1) + 0.5 * uniform(a2);
                } else {
                    mval = pb.dot(vin, rJ) + 0.5 * pb.dot_masked(vin, work, mask);
                }
                if (state == EVAL_A) {
                    qa    = mval;
                    state = EVAL_B;
                    vin   = d;
                    continue;
                }
                if (state == EVAL_B) {
                    bool a_best = qa < mval;
                    const double *best = a_best ? r : d;
                    for (int j = 0; j < nj; ++j) {
                        int i = j * WAVE + lane;
                        if (i < n)
                            step[i] = best[i];
                    }
                    q_model = a_best ? qa : mval;
                    break;
                }
                q_model = mval; // EVAL_STEP
                break;
            }
            PANTR_REGION(1);
            PANTR_FINE(12);
            st.cg_iterations += cg_its;
            // full step: Newton part on J, prox step on the active set
            bool fin = true;
            OV xc;             // NJ > 0: the candidate xbar + q and q'q, formed in the same pass
            double qq_acc = 0.0; // q'q
            if constexpr (NJ > 0) {
                const OV sv = OV::load(step, lane), pp = OV::load(P_(prox), lane), xp = OV::load(X_(prox), lane);
                OV qv;
#pragma unroll
                for (int j = 0; j < NJ; ++j) { // (padding: outside J, p = 0)
                    qv[j]  = (mask >> j) & 1u ? sv[j] : pp[j];
                    fin    = fin && is_finite(qv[j]);
                    xc[j]  = xp[j] + qv[j];
                    qq_acc = qq_acc + qv[j] * qv[j];
                }
                qv.store(q, lane);
                qq_acc = uniform(wave_sum(qq_acc));
            } else {
                for (int j = 0; j < nj; ++j) {
                    int i = j * WAVE + lane;
                    if (i < n) {
                        double qi = (mask >> j) & 1u ? step[i] : P_(prox)[i];
                        q[i]      = qi;
                        fin       = fin && is_finite(qi);
                    }
                }
            }
            bool q_finite = __builtin_amdgcn_readfirstlane(__all(fin ? 1 : 0)) != 0;
            // the model value, if it is still with the helper (the candidate below is then evaluated on
            // the expectation q_model < 0 and discarded - iterate `cand` is scratch - should it not hold)
            auto settle_model = [&]() {
                if constexpr (PB::HAS_SPEC) {
                    if (model_deferred && !pb.spec_take_model(q_model)) { // not delivered: evaluate it here
                        pb.hess_prod(X_(prox), step, work);
                        q_model = pb.dot(step, rJ) + 0.5 * pb.dot_masked(step, work, mask);
                    }
                }
                model_deferred = false;
            };
            if (!q_finite)
                settle_model();
            PANTR_FINE(13);
            if (q_finite && (model_deferred || q_model < 0)) {
                if constexpr (NJ > 0) {
                    xc.store(X_(cand), lane);
                } else {
                    for (int j = 0; j < nj; ++j) {
                        int i = j * WAVE + lane;
                        if (i < n)
                            X_(cand)[i] = X_(prox)[i] + q[i];
                    }
                }
                PANTR_PHASE(PH_PSI_GRAD, cand.psi_x = pb.eval_psi_grad(X_(cand), G_(cand)));
                cand.gamma = prox.gamma;
                cand.L     = prox.L;
                PANTR_FINE(14);
                pb.prox_grad_step(cand.gamma, X_(cand), G_(cand), XH_(cand), P_(cand), cand.pTp,
                                  cand.gradTp);
                settle_model();
            }
            PANTR_FINE(15);
            if (q_finite && q_model < 0) {
                ++st.n_psi_grad_psi;
                ++st.n_prox_grad_step;
                double phi_k = prox.fbe(), phi_n = cand.fbe();
                double tr_eps = (1.0 + fabs(phi_k)) * P.TR_tolerance_factor;
                rho    = ((phi_k - phi_n) + tr_eps) / (-q_model);
                accept = rho >= P.ratio_threshold_acceptable;
                double qn;
                if constexpr (NJ > 0)
                    qn = __builtin_sqrt(qq_acc);
                else
                    qn = __builtin_sqrt(pb.dot(q, q));
                double nr = rho >= P.ratio_threshold_good ? fmax(P.radius_factor_good * qn, radius)
                            : rho >= P.ratio_threshold_acceptable ? radius * P.radius_factor_acceptable
                                                                  : P.radius_factor_rejected * qn;
                if (rho != rho)
                    nr = P.radius_factor_rejected * qn;
                radius = fmax(nr, P.min_radius);
            }
        }
        if (trace && k < trace_cap && lane == 0) {
            double *o = trace + (size_t)k * TRACE_COLS;
            o[0] = eps_k; o[1] = curr.psi_x; o[2] = __builtin_sqrt(curr.pTp); o[3] = curr.gamma;
            o[4] = radius; o[5] = rho; o[6] = q_model; o[7] = curr.fbe(); o[8] = nJ; o[9] = cg_its;
            o[10] = accept ? 1.0 : 0.0;
        }
        if (accept) {
            which = 1;
            ++st.accepted;
        } else {
            which = 2;
            ++st.rejected;
        }
    }

    if constexpr (PB::HAS_SPEC)
        pb.spec_exit();
#ifdef PANTR_REST_TIMING
    PANTR_REGION(1);
    PANTR_FINE(0);
    for (int i = 0; i < 4; ++i)
        st.phase_s[i] += rest_bucket_[i];
#endif
    // ---- results ----
    if ((status == Converged || status == Interrupted || P.always_overwrite_results) && is_finite(curr.L)) {
        pb.update_multipliers(curr.yh); // err_z = Sigma^{-1} (yhat - y), y <- yhat  (m > 0)
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + lane;
            if (i < n)
                x_io[i] = XH_(curr)[i];
        }
        wave_sync();
    } else {
        // nothing to report (the Lipschitz estimate was not finite): x_io keeps the initial guess.  Its slot is
        // scratch during the solve (CG step; sweep scratch of the scalar models), so it is written back
        // from the iterate that still holds it.
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + lane;
            if (i < n)
                x_io[i] = X_(curr)[i];
        }
        wave_sync();
    }
    return {status, k, eps_k, curr.gamma, curr.psi_xh, curr.fbe(), radius};
}

/// The inner solver `method` names: 0 = pantr_inner() above, 1 = PANOC with L-BFGS, 2 = PANOC with
/// structured L-BFGS (panoc_kernel.hip.h).
template <int METHOD, class PB>
__device__ __forceinline__ InnerResult inner_solve(PB &pb, const KernelParams &P, double tolerance,
                                                   double deadline_s, long long t_start, double *x_io,
                                                   DeviceStats &st, double *trace, int trace_cap,
                                                   double *lbfgs_ws);

// ---------------------------------------------------------------------------------------
// The solve kernel for problems without general constraints (m = 0: ALM = one inner solve)
// ---------------------------------------------------------------------------------------
/// How far ahead the products of the two PANTR kernels of the hanging chain request their tape rows: two stages, both.
/// One wavefront per instance is what large batches run: a thousand and more wavefronts stream their tapes at once and
/// the memory system delivers bytes-in-flight / latency (round 4: N = 120, 1024 instances: 704 -> 521 ms per pass).  Two
/// wavefronts per instance is what small batches run; in round 4 a stage of a product took 870 cycles there, one stage
/// ahead covered a trip to the Infinity Cache and the second stage in flight only cost instructions (67.6 -> 74.9 ms).
/// With both substage chains side by side (round 5) a stage is 640 / 870 cycles and one stage ahead no longer covers the
/// trip while the batch is busy: two stages ahead takes the mean solve of the bench batch from 23.2 to 21.2 ms (the pass,
/// which ends on one straggler running alone out of L2, from 54.6 to 53.9 ms).  LDS of a workgroup: 49 KiB (three per CU).
constexpr int CHAIN_SOLVE_PREFETCH_DEPTH = 2;
#ifndef PANTR_SOLVE_RECOMPUTE // development switch (A/B): 0 = the one-wavefront kernel keeps the tangent tape
#define PANTR_SOLVE_RECOMPUTE 1
#endif
/// The one-wavefront kernel - large, memory-bound batches - runs the products that recompute (WaveProblem, RECOMP).
constexpr bool CHAIN_SOLVE_RECOMPUTE = PANTR_SOLVE_RECOMPUTE != 0;
#ifndef PANTR_SPEC_DEPTH // development switch (A/B)
#define PANTR_SPEC_DEPTH 2
#endif
constexpr int CHAIN_SPEC_PREFETCH_DEPTH  = PANTR_SPEC_DEPTH;
template <class M, int METHOD, int NJ = 0>
using SolveProblemT = WaveProblem<M, METHOD == 0, false, METHOD == 0 ? CHAIN_SOLVE_PREFETCH_DEPTH : 1, NJ, METHOD == 0 && CHAIN_SOLVE_RECOMPUTE>;
template <class M, int NJ = 0>
using SpecProblemT = WaveProblem<M, true, true, CHAIN_SPEC_PREFETCH_DEPTH, NJ>;

/// One instance, start to finish, on the calling wavefront (the body of both solve kernels).
template <int METHOD, class PB, class M>
__device__ __forceinline__ void solve_chain_instance(const SolveArgs &a, PB &pb, const typename M::Lane &, int lane,
                                                     int inst, long long t_start, double *lds) {
    const KernelParams &P = a.prm;
    const int n = a.n, nj = a.nj;
    for (int i = lane; i < PB::LDS_SLOTS * a.npad; i += WAVE)
        lds[i] = 0.0;
    if (METHOD == 0)
        for (int i = lane; i < NUM_ITER_VEC * a.npad; i += WAVE)
            pb.giter[i] = 0.0;
    wave_sync();
    double *x_io = pb.vec(PB::XIO_SLOT);
    for (int j = 0; j < nj; ++j) {
        int i = j * WAVE + lane;
        if (i < n)
            x_io[i] = a.u[(size_t)inst * n + i];
    }
    DeviceStats st{};
    const bool trace = a.trace != nullptr && inst < a.trace_instances;
    InnerResult r = inner_solve<METHOD>(
        pb, P, P.tolerance, P.max_time_s, t_start, x_io, st,
        trace ? a.trace + (size_t)inst * a.trace_cap * TRACE_COLS : nullptr, a.trace_cap,
        a.lbfgs ? a.lbfgs + (size_t)inst * 2 * P.lbfgs_memory * a.npad : nullptr);
    wait_vmem(); // the tape rows the last product left in flight for a successor
    for (int j = 0; j < nj; ++j) { // solution (or the untouched guess when nothing was overwritten)
        int i = j * WAVE + lane;
        if (i < n)
            a.u[(size_t)inst * n + i] = x_io[i];
    }
    if (lane == 0) {
        st.status       = r.status;
        st.iterations   = r.iterations;
        st.eps          = r.eps;
        st.final_gamma  = r.gamma;
        st.final_psi    = r.psi;
        st.final_fbe    = r.fbe;
        st.final_radius = r.radius;
        st.elapsed_s    = (double)(wall_clock64() - t_start) * P.tick_s;
        st.outer_iterations = 1;
        st.inner_convergence_failures = r.status == Converged ? 0 : 1;
        st.spec_timeouts = pb.spec_timeouts;
#ifdef PANTR_PHASE_TIMING
        st.sum_tau = (double)pb.fwd_cycles; // profiling build, method=pantr: shader cycles in the forward halves of the products
#endif
        write_stats(a.stats + inst, st);
    }
}

template <class M, int METHOD, int NJ = 0>
__global__ __launch_bounds__(WAVE, 2) void pantr_solve_kernel(SolveArgs a) { // <= 256 VGPRs: two wavefronts per SIMD (uncapped it would take 260)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    const int inst = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    if (inst >= a.B)
        return;
    const long long t_start = wall_clock64(); // constant-rate clock, P.tick_s seconds per tick
    typename M::Lane L(lane);

    using PB = SolveProblemT<M, METHOD, NJ>; // PANTR: iterates in HBM; the comparators: everything in LDS
    __shared__ __attribute__((aligned(16))) char tape_buf0[PB::LANDING_BYTES];
    __shared__ __attribute__((aligned(16))) char tape_buf1[PB::LANDING_BYTES];
    PB pb{L, a.prm.box, lds, a.N, a.n, a.npad, a.nj, {}, nullptr, nullptr, nullptr, nullptr, tape_buf0, tape_buf1};
    if constexpr (PB::PREFETCH_DEPTH == 2) {
        __shared__ __attribute__((aligned(16))) char tape_buf2[PB::LANDING_BYTES];
        pb.tbuf2 = tape_buf2;
    }
    pb.x0 = M::load_x0(L, a.x0 + (size_t)inst * M::nx);
    pb.bind_workspace(a.ws, inst);
    pb.xrow = lds + PB::LDS_SLOTS * a.npad;
    pb.lbv = a.lbv; pb.ubv = a.ubv;
    solve_chain_instance<METHOD, PB, M>(a, pb, L, lane, inst, t_start, lds);
}

/// method = pantr with speculation (SpecBox): two wavefronts per instance.  Wavefront 0 is the solver
/// of pantr_solve_kernel<M, 0>; wavefront 1 serves its requests with its own LDS vectors (slots 0..4:
/// it uses the product output, the sweeps' staging copy and the gradient staging), exchange row,
/// landing buffers, trajectory and tangent rows.  Dynamic LDS: spec_lds_doubles(npad) doubles.
template <class M>
__host__ __device__ constexpr size_t spec_lds_doubles(size_t npad) {
    using PB = SpecProblemT<M>;
    return (PB::LDS_SLOTS + PB::HELPER_SLOTS) * npad + 2 * PB::XS_DOUBLES;
}
template <class M, int NJ = 0>
__global__ __launch_bounds__(2 * WAVE, 2) void pantr_solve_spec_kernel(SolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (WAVE - 1);
    const int inst = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    if (inst >= a.B)
        return;
    const long long t_start = wall_clock64();
    typename M::Lane L(lane);

    __shared__ __attribute__((aligned(16))) char tape_buf0[SpecProblemT<M>::LANDING_BYTES];
    __shared__ __attribute__((aligned(16))) char tape_buf1[SpecProblemT<M>::LANDING_BYTES];
    __shared__ __attribute__((aligned(16))) char helper_buf0[SpecProblemT<M>::LANDING_BYTES];
    __shared__ __attribute__((aligned(16))) char helper_buf1[SpecProblemT<M>::LANDING_BYTES];
    __shared__ SpecBox box;
    using PB = SpecProblemT<M, NJ>;
    if (threadIdx.x == 0) {
        box.req_seq = 0;
        box.done_seq = 0;
        box.cmd = 0;
        box.abort = 0;
        box.stopped = 0;
    }
    __syncthreads(); // the only workgroup barrier: the mailbox exists before either wavefront reads it
    // The two wavefronts build their problem objects in branches of their own, so that each sees its landing
    // buffers as the constants they are (an address selected by the wavefront's index would be a register value:
    // every LDS-DMA transfer would fetch it with v_readfirstlane before it can write M0).
    auto bind = [&](PB &pb, double *my, int slots) {
        pb.x0 = M::load_x0(L, a.x0 + (size_t)inst * M::nx);
        pb.bind_workspace(a.ws, inst);
        pb.xrow   = my + slots * a.npad;
        pb.wg_lds = lds;
        pb.spec   = &box;
        pb.spec_tick_s = a.prm.tick_s;
        pb.spec_patience_s = a.prm.spec_patience_s;
        pb.lbv = a.lbv; pb.ubv = a.ubv;
    };
    if (wave == 1) {
        double *my = lds + PB::LDS_SLOTS * a.npad + PB::XS_DOUBLES;
        PB pb{L, a.prm.box, my, a.N, a.n, a.npad, a.nj, {}, nullptr, nullptr, nullptr, nullptr, helper_buf0, helper_buf1};
        if constexpr (PB::PREFETCH_DEPTH == 2) {
            __shared__ __attribute__((aligned(16))) char helper_buf2[PB::LANDING_BYTES];
            pb.tbuf2 = helper_buf2;
        }
        bind(pb, my, PB::HELPER_SLOTS);
        for (int i = lane; i < PB::HELPER_SLOTS * a.npad; i += WAVE)
            my[i] = 0.0;
        wave_sync();
        pb.X  = pb.helper_X();
        pb.Xt = pb.helper_Xt();
        pb.st = pb.helper_st();
        pb.spec_serve(a.prm, t_start);
        return;
    }
    double *my = lds;
    PB pb{L, a.prm.box, my, a.N, a.n, a.npad, a.nj, {}, nullptr, nullptr, nullptr, nullptr, tape_buf0, tape_buf1};
    if constexpr (PB::PREFETCH_DEPTH == 2) {
        __shared__ __attribute__((aligned(16))) char tape_buf2[PB::LANDING_BYTES];
        pb.tbuf2 = tape_buf2;
    }
    bind(pb, my, PB::LDS_SLOTS);
    // where a SIMD holds a helper of one instance and the solver of another, the solver issues first
    __builtin_amdgcn_s_setprio(3);
    solve_chain_instance<0, PB, M>(a, pb, L, lane, inst, t_start, my);
}

// ---------------------------------------------------------------------------------------
// Stand-alone problem-function kernels (parity tests of rows a7-a9, a13; same device code)
// ---------------------------------------------------------------------------------------
enum EvalMode : int32_t { EVAL_PSI = 0, EVAL_PSI_GRAD = 1, EVAL_HESS_PROD = 2, EVAL_PROX = 3,
                          EVAL_HESS_PROD_TAPE = 4 };

struct EvalArgs {
    int32_t B, N, n, npad, nj, mode, reps, pad0;
    const double *x0; ///< [B][nx]
    const double *u;  ///< [B][n]
    const double *v;  ///< [B][n]  (HESS_PROD: direction; PROX: gradient)
    double *out_s;    ///< [B][4]  psi | (pTp, gradTp, eps_unit, nJ)
    double *out_v;    ///< [B][n]  grad | Hv | xhat
    double *out_v2;   ///< [B][n]  PROX: p
    double *ws;       ///< [B][WaveProblem::workspace_doubles(N)]
    double gamma;
    BoxData box;
};

/// DEPTH: how far ahead the products of this kernel request their tape rows (both forms of hess_prod_tape are
/// tested against the oracle through it).
template <class M, int DEPTH_ = 1>
__global__ __launch_bounds__(WAVE) void pantr_eval_kernel(EvalArgs a) {
    constexpr int DEPTH = DEPTH_ == 3 ? 2 : DEPTH_; // 3: two stages ahead, recomputing (the one-wavefront solve kernel's form)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x, inst = blockIdx.x;
    if (inst >= a.B)
        return;
    typename M::Lane L(lane);
    using PB = WaveProblem<M, false, false, DEPTH, 0, DEPTH_ == 3>;
    __shared__ __attribute__((aligned(16))) char landing[(DEPTH + 1) * PB::LANDING_BYTES];
    PB pb{L, a.box, lds, a.N, a.n, a.npad, a.nj, {}, nullptr, nullptr, nullptr, nullptr, landing, landing + PB::LANDING_BYTES};
    if constexpr (DEPTH == 2)
        pb.tbuf2 = landing + 2 * PB::LANDING_BYTES;
    pb.x0 = M::load_x0(L, a.x0 + (size_t)inst * M::nx);
    pb.bind_workspace(a.ws, inst);
    pb.xrow = lds + 4 * a.npad;
    double *u = pb.vec(0), *v = pb.vec(1), *o = pb.vec(2), *o2 = pb.vec(3);
    for (int i = lane; i < 4 * a.npad; i += WAVE)
        lds[i] = 0.0;
    wave_sync();
    for (int i = lane; i < a.n; i += WAVE) {
        u[i] = a.u[(size_t)inst * a.n + i];
        if (a.v)
            v[i] = a.v[(size_t)inst * a.n + i];
    }
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (a.mode == EVAL_PSI) {
        s0 = pb.template psi_forward<false>(u);
    } else if (a.mode == EVAL_PSI_GRAD) {
        s0 = pb.psi_grad(u, o);
#ifdef PANTR_PHASE_TIMING
        // profiling build: shader cycles of the three first-order sweeps, `reps` runs each (scripts/sweep_timing.py)
        const int reps = a.reps > 0 ? a.reps : 1;
        long long c0 = clock64();
        for (int r = 0; r < reps; ++r)
            s0 = pb.template psi_forward<true>(u);
        long long c1 = clock64();
        for (int r = 0; r < reps; ++r)
            pb.template grad_reverse<false>(u, o);
        long long c2 = clock64();
        for (int r = 0; r < reps; ++r)
            pb.template grad_reverse<true>(u, o);
        long long c3 = clock64();
        s1 = (double)(c1 - c0), s2 = (double)(c2 - c1), s3 = (double)(c3 - c2);
#endif
    } else if (a.mode == EVAL_HESS_PROD_TAPE) {
        s0 = pb.template psi_grad<true>(u, o2); // record at u, then replay (the solver's path)
        long long c0 = clock64(), w0 = wall_clock64();
        for (int r = 0; r < (a.reps > 0 ? a.reps : 1); ++r)
            pb.hess_prod_tape(v, o);
        s1 = (double)(clock64() - c0);      // shader cycles
        s2 = (double)(wall_clock64() - w0); // constant-rate clock ticks
#ifdef PANTR_PHASE_TIMING
        s3 = (double)pb.fwd_cycles;         // of which the forward sweeps (profiling build)
#endif
        wait_vmem();                        // the rows left in flight for a successor
    } else {
        pb.prox_grad_step(a.gamma, u, v, o, o2, s0, s1);
        s2 = pb.proj_grad_unit_norm(u, v);
        int nJ;
        pb.inactive_mask(a.gamma, u, v, nJ);
        s3 = nJ;
    }
    for (int i = lane; i < a.n; i += WAVE) {
        if (a.out_v)
            a.out_v[(size_t)inst * a.n + i] = o[i];
        if (a.out_v2)
            a.out_v2[(size_t)inst * a.n + i] = o2[i];
    }
    if (lane == 0 && a.out_s) {
        double *os = a.out_s + (size_t)inst * 4;
        os[0] = s0; os[1] = s1; os[2] = s2; os[3] = s3;
    }
}

/// Plant step x+ = f_d(x, u) for a batch (formulation.hpp:258-261).  u rows are `u_stride`
/// doubles apart (nu for a plain input batch, n to take the first block of a decision vector);
/// xn may alias x.
template <class M>
__global__ __launch_bounds__(WAVE) void pantr_simulate_kernel(int B, const double *x, const double *u,
                                                              int u_stride, double *xn) {
    const int lane = threadIdx.x, inst = blockIdx.x;
    if (inst >= B)
        return;
    typename M::Lane L(lane);
    __shared__ double xs[M::PAIR_XS_DOUBLES]; // exchange rows of the component gather
    auto s    = M::load_x0(L, x + (size_t)inst * M::nx);
    double uc = M::load_input(L, u + (size_t)inst * u_stride);
    auto r    = M::rk4(L, s, uc, xs);
    M::store_x(L, xn + (size_t)inst * M::nx, r);
}

/// Warm-start shift of the closed loop (mpc-driver.cpp:168-170): u[0:n-nu] = u[nu:n], last block
/// kept; or, cold, u = the initial guess (mpc-driver.cpp:192-193; zero for this problem).
__global__ void pantr_shift_kernel(int B, int n, int nu, int warm, double *u) {
    const int inst = blockIdx.x;
    if (inst >= B)
        return;
    double *ui = u + (size_t)inst * n;
    for (int base = 0; base < n; base += blockDim.x) { // ascending blocks: reads stay ahead of writes
        int i    = base + threadIdx.x;
        double v = 0.0;
        if (warm && i < n)
            v = i + nu < n ? ui[i + nu] : ui[i];
        __syncthreads();
        if (i < n)
            ui[i] = v;
        __syncthreads();
    }
}

} // namespace pantr
