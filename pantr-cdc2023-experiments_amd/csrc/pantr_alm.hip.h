// Problems with general constraints on gfx950: single shooting with an augmented-Lagrangian term,
// scalar stage models (quadcopter family), and the ALM outer loop around pantr_inner().
//
// What it replaces:
//   psi, psi_grad_psi, hess_psi_prod with zeta = g + y/Sigma, zhat = clamp(zeta, D),
//   d = zeta - zhat, yhat = Sigma d, psi = f + 1/2 yhat'd      problems/problems/formulation.hpp:79-99
//   constraints on x_0..x_{N-1} and x_N                          python/.../formulations/ss.py:39-40
//   alpaqa::ALMSolver::operator() (m > 0)                        drivers/mpc-driver.cpp:104-115,217
//   ALM options                                                  new-benchmarks-paper/dodo.py:12-22,101-105
// One wavefront per MPC instance as for the chain.  The nine-state stage is small and its states are a
// cascade (attitude -> velocities -> positions), so very little is a recursion over the stages, and what
// is not runs with lane = stage (trajectories, multipliers and the sin/cos values of every substage
// state live in LDS):
//   forward sweep   running fmas for velocities and positions (quadcopter: the angles too); only the
//                   attitude of realistic_quadcopter is integrated stage by stage        forward_cascade
//   adjoint sweep   running sums; realistic_quadcopter's attitude adjoints as a wavefront scan of
//                   affine 2 x 2 maps                                                      reverse_cascade
//   recording       one (stage, direction) pair per lane on dual numbers, 7 directions + the position
//                   block: the nonzero blocks of J_k and H_k, coefficient-major in HBM     eval_grad_xhat
//   products        tape rows into registers, a few dozen FMAs per lane, wavefront prefix sums (and
//                   the same affine scans): no loop over the stages                        hess_prod
// The summation / composition orders of the scans are those of the oracle's prefix_scan / affine_scan
// (oracle/quad_models.hpp), which is what keeps the results bit-identical.
#pragma once
#include "pantr_kernel.hip.h"
#include <type_traits>

namespace pantr {

/// ALM options as the kernel consumes them.
struct AlmKernelParams {
    double tolerance, dual_tolerance, max_time_s, initial_penalty, penalty_update_factor,
        initial_tolerance, tolerance_update_factor, rel_penalty_increase_threshold, max_multiplier,
        max_penalty;
    int32_t max_iter, single_penalty_factor;
};

struct AlmSolveArgs {
    int32_t B, N, n, npad, nj, m, mpad, mj;
    int32_t trace_cap, trace_instances;
    const double *x0; ///< [B][nx]
    double *u;        ///< [B][n] in/out
    double *y;        ///< [B][m] in/out
    double *ws;       ///< [B][ScalarALProblem::workspace_doubles(N)] second-order tape
    double *lbfgs;    ///< [B][2 * lbfgs_memory * npad] L-BFGS rings (method != pantr) or nullptr
    DeviceStats *stats;
    double *trace;
    const int32_t *order;
    KernelParams prm;
    AlmKernelParams alm;
};

/// Wave-cooperative single-shooting problem with an augmented-Lagrangian term, scalar model M.
/// NJ_: 0, or the number of elements of an n-vector every lane owns (nj, known at compile time: the element-wise
/// passes of the solver are then straight-line code, Owned<NJ> in pantr_kernel.hip.h).
/// SHORT_: which tape product the instantiation carries - 0: hess_prod()'s general routine (lane = stage), 1: the one
/// for horizons of up to 16 stages (hess_prod_short(); quadcopter), 2: both, chosen by the horizon (the evaluation
/// kernel).  Solve kernels carry one: a second routine in the same kernel moves the first one's registers.
template <class M, bool SPEC = false, int NJ_ = 0, int SHORT_ = 0>
struct ScalarALProblem {
    static_assert(SHORT_ == 0 || M::angles_are_integrated_inputs, "hess_prod_short() is quadcopter's");
    static constexpr int nx = M::nx, nu = M::nu, nc = M::nc;
    static constexpr bool HAS_SPEC = SPEC; ///< a helper wavefront evaluates ahead (pantr_alm_spec_kernel; SpecBox)
    static constexpr int NJ = NJ_;
    int lane;
    const BoxData *box; ///< bounds of u (box C) and of the constraints (box D), kernel argument
    double *lds; ///< n-vectors: NUM_VEC slots of npad doubles, then the two of the bounds
    /// LDS n-vectors: box C tiled over the horizon (element i: the bounds of input i mod nu), padding 0 - so that a
    /// pass reads its bounds like any other operand (a select among the kernel arguments on i mod nu compiles to
    /// nested exec-mask branches: 1.5 us per pass), and a padding element's prox step is exactly 0.
    double *lbv, *ubv;
    static constexpr int BOX_VECS = 2;
    int N, n, npad, nj, m, mpad, mj;
    const double *x0; ///< LDS: the initial state (problem parameter) of this instance, nx doubles
    double *mv;  ///< m-vectors: y, Sigma, e1, e2, yh0, yh1, yh2, y/Sigma, mpad doubles each
    double *X;   ///< LDS: primal trajectory (N+1) x nx
    double *Xt;  ///< LDS: tangent trajectory (N+1) x nx of a product | adjoints lam_k while the tape is recorded
    double2 *TG; ///< LDS: trig store, 12 (sin, cos) pairs per stage (QuadModelT::trig_load)
    double *scr;  ///< LDS: scratch_doubles(N) of scratch for the cascade sweeps (16-byte aligned)
    double *coef; ///< LDS: coef_doubles(N), see there
    double *tape; ///< HBM: this instance's second-order tape (see workspace_doubles)
    enum { MV_Y = 0, MV_SIG = 1, MV_E1 = 2, MV_E2 = 3, MV_YH = 4, MV_YS = 7, MV_COUNT = 8 };

    // Second-order tape (oracle: SingleShootingAL::StageTape).  With z = (x_k, u_k), x = (p, v, a),
    // u = (thrust, w): J = d x_{k+1} / d z and H = d/dz of the stage's adjoint outputs at fixed
    // lam_{k+1} have the block structure of the cascade attitude -> velocity -> position,
    //   p+ = p + cpv v + Jp (a,u)     v+ = v + Jv (a,u)     a+ = Ja (a,w)      H = blockdiag(Hpp, hvv I, Hau)
    // with (a,u) the NAU = 7 directions that need a dual evaluation of the stage.  The tape holds the
    // blocks coefficient-major - entry c of stage k at tape[c * pitch + k] - so that the products,
    // which work on all stages at once (lane = stage), read contiguous rows:
    //   C_JP + 7 i + e : d p+_i / d (a,u)_e        C_JV + 7 i + e : d v+_i / d (a,u)_e
    //   C_JA + 6 i + e : d a+_i / d (a_0..2, w_0..2)_e
    //   C_HAU + 7 l + e: derivative along (a,u)_l of adjoint output (a,u)_e
    //   C_HPP + 3 l + e: derivative along p_l of adjoint output p_e
    // and after them the terminal block: HNpp[3][3], HNaa[3][3].  cpv = c6 * 6 (d p+ / d v) and
    // hvv = 2 q_v are constants.
    static constexpr int nz = nx + nu, NAU = 3 + nu;
    static constexpr int C_JP = 0, C_JV = C_JP + 3 * NAU, C_JA = C_JV + 3 * NAU, C_HAU = C_JA + 18,
                         C_HPP = C_HAU + NAU * NAU, NCOEF = C_HPP + 9, HNB = 18;
    static constexpr double cpv = M::c6 * 6.0, hvv = 2.0 * 1.0, hvvN = 2.0 * 25.0;
    /// Row pitch of the tape (doubles).  Up to 64 stages it is the constant FIXED_PITCH: the kernels that know
    /// their elements per lane at compile time (NJ_ > 0: n = nu N <= 256) address the 118 rows with immediate
    /// offsets from one base instead of a scalar address computation per row.
    static constexpr int FIXED_PITCH = 64;
    static_assert(NJ_ == 0 || NJ_ * WAVE <= nu * FIXED_PITCH, "NJ_ > 0 implies N <= FIXED_PITCH");
    __host__ __device__ static int tape_pitch(int N) { return N <= FIXED_PITCH ? FIXED_PITCH : (N + 15) & ~15; }
    __host__ __device__ static size_t workspace_doubles(int N) { return (size_t)NCOEF * tape_pitch(N) + HNB; }

    static constexpr int CG_BASE = NUM_ITER_VEC; ///< all 20 vectors in LDS: iterates first, then the CG block
    static constexpr int XIO_SLOT = CG_BASE + C_XIO;
    static constexpr int X0_DOUBLES = 16;        ///< the initial state, at the end of the dynamic LDS
    __device__ __forceinline__ double *vec(int slot) const { return lds + slot * npad; }
    __device__ __forceinline__ double *ivec(int slot) const { return lds + slot * npad; }
    __device__ __forceinline__ double *mvec(int slot) const { return mv + slot * mpad; }
    /// dedicated_scratch: the cascade sweeps' scratch gets LDS of its own (the comparators, whose direction
    /// keeps data in the CG block; PANTR and the eval kernel lend them the five vector slots z, r, d, (Bd), step,
    /// which are dead during sweeps: 5 npad >= 20 N covers 18 (N + 1) + 2 for every N, checked at handle creation)
    __host__ __device__ static size_t scratch_doubles(int N) { return (size_t)18 * (N + 1) + 2; }
    /// realistic_quadcopter: the coefficients of the angle-rate adjoints (5 per substage), evaluated for
    /// all stages at once ahead of the sequential angle-adjoint recursion
    static constexpr int COEF = 20;
    __host__ __device__ static size_t coef_doubles(int N) { return M::angles_are_integrated_inputs ? 0 : (size_t)COEF * N; }
    __host__ __device__ static size_t lds_doubles(int N, int npad, int mpad, bool dedicated_scratch = false) {
        return (size_t)(NUM_VEC + BOX_VECS) * npad + (size_t)MV_COUNT * mpad + (size_t)2 * (N + 1) * nx +
               (size_t)(N + 1) * M::TRIG_DOUBLES_PER_STAGE + X0_DOUBLES + coef_doubles(N) +
               (dedicated_scratch ? scratch_doubles(N) : 0);
    }

    /// One RK4 step of the full state, every lane on its own state (the simulate kernel); t1 = Trig of s.
    /// The sweeps of the solver never take this path: they integrate the cascade (forward_cascade).
    template <class T>
    __device__ __forceinline__ void rk4(const typename M::template Trig<T> &t1, const T (&s)[nx], const T (&u)[nu],
                                        T (&xn)[nx]) const {
        T k1[nx], k2[nx], k3[nx], k4[nx], w[nx];
        M::fc(t1, s, u, k1);
#pragma unroll
        for (int l = 0; l < nx; ++l)
            w[l] = madd(M::hTs, k1[l], s[l]);
        M::fc(M::trig(w), w, u, k2);
#pragma unroll
        for (int l = 0; l < nx; ++l)
            w[l] = madd(M::hTs, k2[l], s[l]);
        M::fc(M::trig(w), w, u, k3);
#pragma unroll
        for (int l = 0; l < nx; ++l)
            w[l] = madd(M::Ts, k3[l], s[l]);
        M::fc(M::trig(w), w, u, k4);
#pragma unroll
        for (int l = 0; l < nx; ++l)
            xn[l] = madd(M::c6, madd(2.0, k3[l], madd(2.0, k2[l], k1[l])) + k4[l], s[l]);
    }
    /// The same step and its adjoint on dual numbers whose values and tangents differ per lane (one
    /// (stage, direction) pair per lane): the substage states are rebuilt for their tangents, the
    /// trigonometric VALUES come from the trig store.  Also returns the step itself (x_next), whose
    /// tangent parts are a column of the stage Jacobian.
    __device__ __forceinline__ void rk4_adj_dual(const double2 *tg, const typename M::template Trig<dual> &t1,
                                                 const dual (&s)[nx], const dual (&u)[nu], const dual (&mu)[nx],
                                                 dual (&sbar)[nx], dual (&ubar)[nu], dual (&xn)[nx]) const {
        dual k[nx], ksum[nx];
        double a2[3], a3[3], a4[3]; // angle tangents of the substage states (all the adjoint needs of them)
        M::fc(t1, s, u, k);
        {
            dual w[nx];
#pragma unroll
            for (int l = 0; l < nx; ++l) {
                w[l]    = madd(M::hTs, k[l], s[l]);
                ksum[l] = k[l];
            }
            a2[0] = w[6].t; a2[1] = w[7].t; a2[2] = w[8].t;
            M::fc(M::trig_load(tg + 3, a2[0], a2[1], a2[2]), w, u, k);
        }
        {
            dual w[nx];
#pragma unroll
            for (int l = 0; l < nx; ++l) {
                w[l]    = madd(M::hTs, k[l], s[l]);
                ksum[l] = madd(2.0, k[l], ksum[l]);
            }
            a3[0] = w[6].t; a3[1] = w[7].t; a3[2] = w[8].t;
            M::fc(M::trig_load(tg + 6, a3[0], a3[1], a3[2]), w, u, k);
        }
        {
            dual w[nx];
#pragma unroll
            for (int l = 0; l < nx; ++l) {
                w[l]    = madd(M::Ts, k[l], s[l]);
                ksum[l] = madd(2.0, k[l], ksum[l]);
            }
            a4[0] = w[6].t; a4[1] = w[7].t; a4[2] = w[8].t;
            M::fc(M::trig_load(tg + 9, a4[0], a4[1], a4[2]), w, u, k);
#pragma unroll
            for (int l = 0; l < nx; ++l)
                xn[l] = madd(M::c6, ksum[l] + k[l], s[l]);
        }
        dual kb1[nx], kb2[nx], kb3[nx], kb4[nx], a[nx], ua[nu];
#pragma unroll
        for (int l = 0; l < nx; ++l) {
            dual sg = M::c6 * mu[l];
            kb1[l]  = sg;
            kb2[l]  = 2.0 * sg;
            kb3[l]  = 2.0 * sg;
            kb4[l]  = sg;
            sbar[l] = mu[l];
        }
        M::fc_adj(M::trig_load(tg + 9, a4[0], a4[1], a4[2]), u, kb4, a, ubar);
#pragma unroll
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb3[l]  = madd(M::Ts, a[l], kb3[l]);
        }
        M::fc_adj(M::trig_load(tg + 6, a3[0], a3[1], a3[2]), u, kb3, a, ua);
#pragma unroll
        for (int c = 0; c < nu; ++c)
            ubar[c] = ubar[c] + ua[c];
#pragma unroll
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb2[l]  = madd(M::hTs, a[l], kb2[l]);
        }
        M::fc_adj(M::trig_load(tg + 3, a2[0], a2[1], a2[2]), u, kb2, a, ua);
#pragma unroll
        for (int c = 0; c < nu; ++c)
            ubar[c] = ubar[c] + ua[c];
#pragma unroll
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb1[l]  = madd(M::hTs, a[l], kb1[l]);
        }
        M::fc_adj(t1, u, kb1, a, ua);
#pragma unroll
        for (int c = 0; c < nu; ++c)
            ubar[c] = ubar[c] + ua[c];
#pragma unroll
        for (int l = 0; l < nx; ++l)
            sbar[l] = sbar[l] + a[l];
    }

    /// The sequential part of the sweeps: y <- step(operands of the stage, y) over the stages, component j3
    /// of a 3-vector on every lane (lane j < 3 of every 16-lane row: component j; the other lanes shadow
    /// component 2 - every lane stores, lanes of equal j3 the same value to the same address).  UP: stages
    /// 0..N-1, y(k) stored to out[k nx] and the final value to out[N nx]; otherwise stages N-1..0 with
    /// y(k+1) stored to out[(k+1) nx]  (`out` points at the component's column).  prep(k) READS the
    /// operands of stage k (no arithmetic: the wavefront would wait for the reads right there); those of the
    /// next group of G stages are requested before the current group runs, and `step` derives what it needs
    /// from them ahead of its dependent chain.  G x (LDS reads of prep) <= 12: the hardware counts 15
    /// outstanding reads.
    template <bool UP, int G, class Prep, class Step>
    __device__ __forceinline__ double stage_recursion(double y, double *out, Prep prep, Step step) const {
        using Ops = decltype(prep(0));
        auto at   = [&](int i) { return UP ? i : N - 1 - i; }; // i-th stage visited
        auto fetch = [&](int i0, Ops (&o)[G]) {
#pragma unroll
            for (int q = 0; q < G; ++q)
                o[q] = prep(at(i0 + q));
        };
        auto run = [&](int i0, const Ops (&o)[G]) {
#pragma unroll
            for (int q = 0; q < G; ++q) {
                out[(UP ? at(i0 + q) : at(i0 + q) + 1) * nx] = y;
                y = step(o[q], y);
            }
        };
        int i = 0;
        if (G <= N) {
            Ops oA[G], oB[G];
            fetch(0, oA);
            while (true) {
                const bool more = i + 2 * G <= N;
                if (more)
                    fetch(i + G, oB);
                run(i, oA);
                i += G;
                if (!more)
                    break;
                const bool more2 = i + 2 * G <= N;
                if (more2)
                    fetch(i + G, oA);
                run(i, oB);
                i += G;
                if (!more2)
                    break;
            }
        }
        for (; i < N; ++i) { // fewer than G stages left
            out[(UP ? at(i) : at(i) + 1) * nx] = y;
            y = step(prep(at(i)), y);
        }
        if (UP)
            out[N * nx] = y;
        return y;
    }

    /// Forward sweep.  The states form a cascade angles -> velocities -> positions, so little is
    /// sequential: three short scans (one fma per stage and component) and, for realistic_quadcopter,
    /// the attitude (A+B below as one stage-by-stage integration of the three angles; for quadcopter,
    /// whose angle rates are inputs, A is a scan as well and B parallel).  Everything else is evaluated
    /// for all stages at once, lane = stage (or lane = (stage, substage, angle) for the trigonometry).
    /// Every scalar is computed by the expression a stage-by-stage RK4 sweep over the full state uses
    /// for it (oracle: SingleShootingAL::forward; same operands, same order), so the results - psi, the
    /// trajectory, yhat, the trig store - are the same bits:
    ///   A  angles:     a_{k+1} = fma(c6, fma(2,w,fma(2,w,w)) + w, a_k), w = u_k[1+j]        (scan)
    ///   B  sin/cos of the substage angles a_k, fma(Ts/2,w,a_k) (twice), fma(Ts,w,a_k)      (parallel)
    ///   C  accelerations of the four substages and their Runge-Kutta combination         (parallel)
    ///   D  velocities: v_{k+1} = fma(c6, C_k, v_k)                                        (scan)
    ///   E  position slopes v_k, fma(Ts/2,a1,v_k), fma(Ts/2,a2,v_k), fma(Ts,a3,v_k), combined (parallel)
    ///   F  positions:  p_{k+1} = fma(c6, E_k, p_k)                                        (scan)
    ///   G  constraints, multiplier estimates, stage costs                                (parallel)
    ///   H  the two sums psi is made of, in stage order                                   (scan)
    /// Scratch: rows of the tangent trajectory Xt (accelerations) and scr (5 doubles per stage: the four
    /// augmented-Lagrangian products and the stage cost; then 3 per stage: the combined slopes).
    __device__ __forceinline__ double forward_cascade(const double *u, double *yh) const {
        wave_sync();
        const double *yS = mvec(MV_YS), *Sig = mvec(MV_SIG);
        // the scans run component j3 (of 3) on every lane: lanes with the same j3 hold, and store, the same value
        const int j3 = (lane & 15) < 3 ? (lane & 15) : 2;
        // state_{k+1} = fma(c6, slope(k), state_k) for k = 0..N-1, state_k stored to X[k][off + j3] on the way
        auto scan = [&](int off, auto read, auto slope) {
            stage_recursion<true, 8>(x0[off + j3], X + off + j3, read,
                                     [&](double raw, double acc) { return madd(M::c6, slope(raw), acc); });
        };
        auto as_is = [](double c) { return c; };
        if constexpr (M::angles_are_integrated_inputs) {
            // ---- A: angles ----
            scan(6, [&](int k) { return u[nu * k + 1 + j3]; },
                 [](double w) { return madd(2.0, w, madd(2.0, w, w)) + w; });
            wave_sync();
            // ---- B: trigonometry of every substage state: task = (stage, {s, s + Ts/2 k, s + Ts k}, angle) ----
            for (int base = 0; base < 9 * (N + 1); base += WAVE) {
                const int t = base + lane, k = t / 9, r = t - 9 * k, sub = r / 3, j = r - 3 * sub;
                if (t < 9 * (N + 1) && (k < N || sub == 0)) {
                    const double a = X[k * nx + 6 + j];
                    const double w = k < N ? u[nu * k + 1 + j] : 0.0;
                    const double val = sub == 0 ? a : sub == 1 ? madd(M::hTs, w, a) : madd(M::Ts, w, a);
                    double sv, cv;
                    sincos_det(val, sv, cv);
                    double2 *tg = TG + 12 * k;
                    if (sub == 0) {
                        tg[j] = make_double2(sv, cv);
                    } else if (sub == 1) { // the second and third substage states have the same angles
                        tg[3 + j] = make_double2(sv, cv);
                        tg[6 + j] = make_double2(sv, cv);
                    } else {
                        tg[9 + j] = make_double2(sv, cv);
                    }
                }
            }
        } else {
            // ---- A+B: the attitude, stage by stage (values the same on every lane; sin/cos: one angle per
            //      lane and a broadcast), leaving the angle rows of X and the trig store ----
            double a[3] = {x0[6], x0[7], x0[8]}, uk[nu], un[nu];
#pragma unroll
            for (int c = 0; c < nu; ++c)
                un[c] = u[c];
            for (int k = 0; k < N; ++k) {
#pragma unroll
                for (int c = 0; c < nu; ++c)
                    uk[c] = un[c];
                if (k + 1 < N)
#pragma unroll
                    for (int c = 1; c < nu; ++c)
                        un[c] = u[nu * (k + 1) + c];
                X[k * nx + 6 + j3] = j3 == 0 ? a[0] : j3 == 1 ? a[1] : a[2];
                double2 *tg = TG + 12 * k;
                double k1[3], k2[3], k3[3], k4[3];
                M::angle_rates(M::trig_uniform(a[0], a[1], a[2], tg), uk, k1);
                M::angle_rates(M::trig_uniform(madd(M::hTs, k1[0], a[0]), madd(M::hTs, k1[1], a[1]),
                                               madd(M::hTs, k1[2], a[2]), tg + 3), uk, k2);
                M::angle_rates(M::trig_uniform(madd(M::hTs, k2[0], a[0]), madd(M::hTs, k2[1], a[1]),
                                               madd(M::hTs, k2[2], a[2]), tg + 6), uk, k3);
                M::angle_rates(M::trig_uniform(madd(M::Ts, k3[0], a[0]), madd(M::Ts, k3[1], a[1]),
                                               madd(M::Ts, k3[2], a[2]), tg + 9), uk, k4);
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    a[j] = madd(M::c6, madd(2.0, k3[j], madd(2.0, k2[j], k1[j])) + k4[j], a[j]);
            }
            X[N * nx + 6 + j3] = j3 == 0 ? a[0] : j3 == 1 ? a[1] : a[2];
            M::trig_uniform(a[0], a[1], a[2], TG + 12 * N);
        }
        wave_sync();
        // ---- C: accelerations of the four substages (rows of Xt: a1 | a2 | a3) and their combination (cmb) ----
        double *cmb = scr + 5 * (N + 1);
        for (int k = lane; k < N; k += WAVE) {
            const double2 *tg = TG + 12 * k;
            const double u0 = u[nu * k];
            double a1[3], a2[3], a3[3], a4[3];
            M::accel(M::trig_load(tg), u0, a1);
            M::accel(M::trig_load(tg + 3), u0, a2);
            M::accel(M::trig_load(tg + 6), u0, a3);
            M::accel(M::trig_load(tg + 9), u0, a4);
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                Xt[k * nx + m]     = a1[m];
                Xt[k * nx + 3 + m] = a2[m];
                Xt[k * nx + 6 + m] = a3[m];
                cmb[3 * k + m]     = madd(2.0, a3[m], madd(2.0, a2[m], a1[m])) + a4[m];
            }
        }
        wave_sync();
        // ---- D: velocities ----
        scan(3, [&](int k) { return cmb[3 * k + j3]; }, as_is);
        wave_sync();
        // ---- E: position slopes, combined ----
        for (int k = lane; k < N; k += WAVE) {
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const double v  = X[k * nx + 3 + m];
                const double k2 = madd(M::hTs, Xt[k * nx + m], v);
                const double k3 = madd(M::hTs, Xt[k * nx + 3 + m], v);
                const double k4 = madd(M::Ts, Xt[k * nx + 6 + m], v);
                cmb[3 * k + m]  = madd(2.0, k3, madd(2.0, k2, v)) + k4;
            }
        }
        wave_sync();
        // ---- F: positions ----
        scan(0, [&](int k) { return cmb[3 * k + j3]; }, as_is);
        wave_sync();
        // ---- G: constraints, multiplier estimates, costs ----
        for (int k = lane; k <= N; k += WAVE) {
            double s[nx], c[nc], uk[nu];
#pragma unroll
            for (int l = 0; l < nx; ++l)
                s[l] = X[k * nx + l];
            M::constr(M::trig_load(TG + 12 * k), s, c);
#pragma unroll
            for (int i = 0; i < nc; ++i) {
                const int j       = nc * k + i;
                const double zeta = c[i] + yS[j];
                const double zhat = tmax(tmin(zeta, box->d_ub[i]), box->d_lb[i]);
                const double d    = zeta - zhat;
                const double yhv  = Sig[j] * d;
                yh[j]             = yhv;
                scr[5 * k + i]     = yhv * d;
            }
            if (k < N) {
#pragma unroll
                for (int cc = 0; cc < nu; ++cc)
                    uk[cc] = u[nu * k + cc];
                scr[5 * k + 4] = M::stage_cost(s, uk);
            } else {
                scr[5 * k + 4] = M::term_cost(s);
            }
        }
        wave_sync();
        // ---- H: psi = (sum of the costs) + 1/2 (sum of the augmented-Lagrangian products), four stages'
        //         terms fetched ahead of their (dependent) additions ----
        double accf = 0.0, accal = 0.0;
        int k = 0;
        for (; k + 4 <= N + 1; k += 4) {
            double e[20];
#pragma unroll
            for (int i = 0; i < 20; ++i)
                e[i] = scr[5 * k + i];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 0; i < nc; ++i)
                    accal = accal + e[5 * q + i];
                accf = accf + e[5 * q + 4];
            }
        }
        for (; k <= N; ++k) {
#pragma unroll
            for (int i = 0; i < nc; ++i)
                accal = accal + scr[5 * k + i];
            accf = accf + scr[5 * k + 4];
        }
        wave_sync();
        return accf + (0.5 * accal);
    }

    /// Forward sweep: psi, trajectory, yhat, and the trig store of every substage state.
    __device__ __forceinline__ double forward(const double *u, double *yh) const { return forward_cascade(u, yh); }
    /// Adjoint sweep, as a cascade in the opposite direction: position adjoints feed the velocity
    /// adjoints feed the angle adjoints, and no adjoint feeds back.  Sequential: running sums of a few
    /// dependent additions per stage and component (stage_recursion); for realistic_quadcopter the
    /// angle adjoint of a stage is an affine function of the next stage's (through the four substage
    /// adjoints of the angle rates), and phase 4 is a wavefront scan of those maps (oracle: reverse_affine).
    /// All the rest - the substage adjoints of f_c with their trigonometric coefficients, the input
    /// gradients - is evaluated for all stages at once, lane = stage, by the pieces M::fc_adj is made
    /// of, on the operands a stage-by-stage adjoint sweep would hand them (oracle:
    /// SingleShootingAL::reverse).  For quadcopter value for value the arithmetic of the stage-by-stage
    /// sweep (what that sweep adds of exact zeros - the position adjoint of f_c is 0 * u_0 - is skipped:
    /// same values, possibly another sign of a zero).
    ///   1  lam_p(k) = (lam_p(k+1) + dl/dp) - (dc/dp)' yhat                                       (scan)
    ///   2  lam_v(k) = ((((lam_v + sg) + 2 sg) + 2 sg) + sg) + dl/dv,   sg = c6 lam_p(k+1)         (scan)
    ///   3  velocity parts of the four substage adjoint seeds, fc_adj on them: A_s (angle parts), u_0 gradient
    ///   4  lam_a(k) = (((((lam_a + A_4) + A_3) + A_2) + A_1) + dl/da) + (dc/da)' yhat             (scan)
    ///      realistic_quadcopter: lam_a(k) = M_k lam_a(k+1) + c_k, the maps from the stage adjoint on zero and
    ///      unit seeds; yaw: running sum, (phi, theta): wavefront scan of 2 x 2 affine maps, then the gradients
    ///   5  quadcopter: angle parts of the seeds -> gradients of the three rate inputs; + cost gradient -> grad
    /// lam_(k+1) of every stage is left in LAM (= rows of Xt) for the second-order recording.
    __device__ __forceinline__ void reverse_cascade(const double *u, const double *yh, double *grad) const {
        wave_sync();
        const int j3 = (lane & 15) < 3 ? (lane & 15) : 2;
        double *LAM = Xt;
        // lam(k) = step(prep(k), lam(k+1)) for k = N-1..0, lam(k+1) stored to LAM[k+1][off + j3] on the way
        auto scan_down = [&](int off, double lam, auto prep, auto step) {
            stage_recursion<false, sizeof(decltype(prep(0))) <= 16 ? 8 : 4>(lam, LAM + off + j3, prep, step);
        };
        // The operands of the two running sums are prepared for all stages at once (lane = stage), two per
        // stage and component at scr[6 k + 2 j], so that a stage of the recursion is one 16-byte read, its
        // additions and a store (a lone wavefront issues an instruction every ~8 cycles: the recursions
        // cost what they count).
        auto pair_at = [&](int k) { return *reinterpret_cast<const double2 *>(scr + 6 * k + 2 * j3); };
        // ---- 1: position adjoints ----
        for (int k = lane; k < N; k += WAVE) {
            const double y3 = yh[nc * k + 3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double sk    = X[k * nx + j];
                scr[6 * k + 2 * j]     = (2.0 * 10.0) * (sk - M::pf(j));
                scr[6 * k + 2 * j + 1] = j < 2 ? (2.0 * sk) * y3 : 0.0;
            }
        }
        wave_sync();
        {
            const double sN = X[N * nx + j3], pf = j3 == 2 ? 0.5 : 0.25; // M::pf(j3)
            double lam = 0.0 + (2.0 * 250.0) * (sN - pf);
            if (j3 < 2)
                lam = lam - ((2.0 * sN) * yh[nc * N + 3]);
            scan_down(0, lam, pair_at, [](const double2 &o, double mu) { return (mu + o.x) - o.y; });
        }
        wave_sync();
        // ---- 2: velocity adjoints ----
        for (int k = lane; k < N; k += WAVE) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                scr[6 * k + 2 * j]     = M::c6 * LAM[(k + 1) * nx + j];
                scr[6 * k + 2 * j + 1] = (2.0 * 1.0) * X[k * nx + 3 + j];
            }
        }
        wave_sync();
        {
            double lam = 0.0 + (2.0 * 25.0) * X[N * nx + 3 + j3];
            scan_down(3, lam, pair_at, [](const double2 &o, double mu) {
                const double t2 = 2.0 * o.x;
                return ((((mu + o.x) + t2) + t2) + o.x) + o.y;
            });
        }
        wave_sync();
        // ---- 3: substage adjoints of f_c on the velocity parts of their seeds (lane = stage).  Leaves, per
        //         stage and angle j, the record R = (A_4, A_3, A_2, A_1, dl/da, (dc/da)' yhat)_j at scr[18 k + 6 j]
        //         (what the recursion of lam_a adds, in the order it adds it), and the thrust gradient ----
        for (int k = lane; k < N; k += WAVE) {
            const double2 *tg = TG + 12 * k;
            double uk[nu], kb[nx], sb[nx], ub[nu], ub0[4];
#pragma unroll
            for (int c = 0; c < nu; ++c)
                uk[c] = u[nu * k + c];
#pragma unroll
            for (int l = 0; l < nx; ++l)
                kb[l] = 0.0;
            double sgp[3], sgv[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                sgp[i] = M::c6 * LAM[(k + 1) * nx + i];
                sgv[i] = M::c6 * LAM[(k + 1) * nx + 3 + i];
            }
            double *R = scr + 18 * k;
            // substage 4: seed c6 lam
#pragma unroll
            for (int i = 0; i < 3; ++i)
                kb[3 + i] = sgv[i];
            M::fc_adj_translation(M::trig_load(tg + 9), uk, kb, sb, ub);
            ub0[0] = ub[0];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                R[6 * j] = sb[6 + j];
            // substage 3: seed 2 c6 lam + Ts a_4, a_4[v] = the position part of seed 4
#pragma unroll
            for (int i = 0; i < 3; ++i)
                kb[3 + i] = madd(M::Ts, sgp[i], 2.0 * sgv[i]);
            M::fc_adj_translation(M::trig_load(tg + 6), uk, kb, sb, ub);
            ub0[1] = ub[0];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                R[6 * j + 1] = sb[6 + j];
            // substage 2: seed 2 c6 lam + Ts/2 a_3
#pragma unroll
            for (int i = 0; i < 3; ++i)
                kb[3 + i] = madd(M::hTs, 2.0 * sgp[i], 2.0 * sgv[i]);
            M::fc_adj_translation(M::trig_load(tg + 3), uk, kb, sb, ub);
            ub0[2] = ub[0];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                R[6 * j + 2] = sb[6 + j];
            // substage 1: seed c6 lam + Ts/2 a_2
#pragma unroll
            for (int i = 0; i < 3; ++i)
                kb[3 + i] = madd(M::hTs, 2.0 * sgp[i], sgv[i]);
            const auto t1 = M::trig_load(tg);
            M::fc_adj_translation(t1, uk, kb, sb, ub);
            ub0[3] = ub[0];
            const double *yb = yh + nc * k;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                R[6 * j + 3] = sb[6 + j];
                R[6 * j + 4] = (2.0 * 1.0) * X[k * nx + 6 + j];
            }
            // (dc/da)' yhat: c_0 = phi, c_1 = theta, c_2 = cos phi cos theta
            R[5]  = yb[0] - ((t1.s0 * t1.c1) * yb[2]);
            R[11] = yb[1] - ((t1.c0 * t1.s1) * yb[2]);
            R[17] = 0.0;
            grad[nu * k] = (((ub0[0] + ub0[1]) + ub0[2]) + ub0[3]) + (2e-4 * uk[0]);
            if constexpr (!M::angles_are_integrated_inputs) {
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) { // in the order the recursion visits them: substage 4 first
                    const auto c = M::angle_rates_adj_coef(M::trig_load(tg + 9 - 3 * sub), uk);
                    double *o    = coef + COEF * k + 5 * sub;
                    o[0] = c.ic;
                    o[1] = c.x;
                    o[2] = c.y;
                    o[3] = c.num;
                    o[4] = c.z;
                }
            }
        }
        wave_sync();
        // ---- 4: angle adjoints ----
        {
            // terminal stage: dl_N/da + (dc/da)' yhat_N
            const auto tN     = M::trig_load(TG + 12 * N);
            const double *ybN = yh + nc * N;
            double muN[3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                muN[j] = 0.0 + (2.0 * 10.0) * X[N * nx + 6 + j];
            muN[0] = muN[0] + (ybN[0] - ((tN.s0 * tN.c1) * ybN[2]));
            muN[1] = muN[1] + (ybN[1] - ((tN.c0 * tN.s1) * ybN[2]));
            if constexpr (M::angles_are_integrated_inputs) {
                struct Six {
                    double2 a, b, c; // (A_4, A_3), (A_2, A_1), (dl/da, (dc/da)' yhat)
                };
                scan_down(
                    6, j3 == 0 ? muN[0] : j3 == 1 ? muN[1] : muN[2],
                    [&](int k) {
                        const double2 *r = reinterpret_cast<const double2 *>(scr + 18 * k + 6 * j3);
                        return Six{r[0], r[1], r[2]};
                    },
                    [&](const Six &o, double mu) {
                        const double l = ((((mu + o.a.x) + o.a.y) + o.b.x) + o.b.y) + o.c.x;
                        return j3 < 2 ? l + o.c.y : l;
                    });
            } else {
                // The angle rates depend on the angles: the four substage adjoints of a stage chain through
                // angle_rates_adj, on top of the A_s of phase 3 - an AFFINE function of lam_a(k+1).  With
                // lane = N - 1 - stage: the chain on a zero seed (with the A_s: the constant term) and on the
                // three unit seeds (the columns of the linear part and of d grad_w / d lam_a), then the yaw
                // adjoint as a running sum (nothing depends on the yaw), (lam_phi, lam_theta) as a wavefront
                // scan of affine 2 x 2 maps, the rate gradients as fma chains (oracle: reverse_affine).
                double lpsi = muN[2], la2[2] = {muN[0], muN[1]};
                for (int j0 = 0; j0 < N; j0 += WAVE) {
                    const int cnt = N - j0 < WAVE ? N - j0 : WAVE;
                    const bool on = lane < cnt;
                    const int k   = on ? N - 1 - (j0 + lane) : 0;
                    const double *R   = scr + 18 * k;
                    const double2 *tg = TG + 12 * k;
                    double uk[nu], Rv[18], co[COEF], tr[4][3]; // tr: sin phi, sin theta, cos theta of the substage
#pragma unroll
                    for (int c = 0; c < nu; ++c)
                        uk[c] = u[nu * k + c];
#pragma unroll
                    for (int q = 0; q < 18; ++q)
                        Rv[q] = R[q];
#pragma unroll
                    for (int q = 0; q < COEF; ++q)
                        co[q] = coef[COEF * k + q];
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub) {
                        const double2 e0 = tg[9 - 3 * sub], e1 = tg[10 - 3 * sub];
                        tr[sub][0] = e0.x; tr[sub][1] = e1.x; tr[sub][2] = e1.y;
                    }
                    auto rates_adj = [&](int sub, const double (&kba)[3], double (&sba)[3], double (&uba)[3]) {
                        const double *o = co + 5 * sub;
                        M::angle_rates_adj_apply(tr[sub][0], tr[sub][1], tr[sub][2],
                                                 typename M::template RateAdjCoef<double>{o[0], o[1], o[2], o[3], o[4]}, kba, sba, uba);
                    };
                    // sbar = mu + a_4 + a_3 + a_2 + a_1 and the summed input adjoints for the seed mu
                    auto chain = [&](const double (&mu)[3], bool withA, double (&sbar)[3], double (&ubar)[3]) {
                        double kb1[3], kb2[3], kb3[3], kb4[3], a[3], ua[3];
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const double sg = M::c6 * mu[j];
                            kb1[j]  = sg;
                            kb2[j]  = 2.0 * sg;
                            kb3[j]  = 2.0 * sg;
                            kb4[j]  = sg;
                            sbar[j] = mu[j];
                            a[j]    = withA ? Rv[6 * j] : 0.0;
                        }
                        rates_adj(0, kb4, a, ubar);
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            sbar[j] = sbar[j] + a[j];
                            kb3[j]  = madd(M::Ts, a[j], kb3[j]);
                            a[j]    = withA ? Rv[6 * j + 1] : 0.0;
                        }
                        rates_adj(1, kb3, a, ua);
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            ubar[j] = ubar[j] + ua[j];
                            sbar[j] = sbar[j] + a[j];
                            kb2[j]  = madd(M::hTs, a[j], kb2[j]);
                            a[j]    = withA ? Rv[6 * j + 2] : 0.0;
                        }
                        rates_adj(2, kb2, a, ua);
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            ubar[j] = ubar[j] + ua[j];
                            sbar[j] = sbar[j] + a[j];
                            kb1[j]  = madd(M::hTs, a[j], kb1[j]);
                            a[j]    = withA ? Rv[6 * j + 3] : 0.0;
                        }
                        rates_adj(3, kb1, a, ua);
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            ubar[j] = ubar[j] + ua[j];
                            sbar[j] = sbar[j] + a[j];
                        }
                    };
                    double cpre[3], cu[3], Mc[3][3], Uc[3][3]; // Mc[e][j] = d lam_a(k)[j] / d lam_a(k+1)[e]
                    {
                        const double zero[3] = {0.0, 0.0, 0.0};
                        chain(zero, true, cpre, cu);
                    }
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        const double seed[3] = {e == 0 ? 1.0 : 0.0, e == 1 ? 1.0 : 0.0, e == 2 ? 1.0 : 0.0};
                        chain(seed, false, Mc[e], Uc[e]);
                    }
                    const double ct0 = (cpre[0] + Rv[4]) + Rv[5], ct1 = (cpre[1] + Rv[10]) + Rv[11], ct2 = cpre[2] + Rv[16];
                    // running sum / affine scan over the lanes (as in hess_prod)
                    double lan[3];
                    {
                        const double sps = wave_running_sum(on ? ct2 : 0.0);
                        lan[2]           = lpsi + lane_below(sps);
                        lpsi             = lpsi + lane_value(sps, cnt - 1);
                    }
                    {
                        const Affine2 mm = on ? Affine2{Mc[0][0], Mc[1][0], Mc[0][1], Mc[1][1], __builtin_fma(Mc[2][0], lan[2], ct0),
                                                        __builtin_fma(Mc[2][1], lan[2], ct1)}
                                              : affine_identity();
                        const Affine2 P  = wave_affine_scan(mm);
                        double y0, y1;
                        P.apply(la2[0], la2[1], y0, y1);
                        const double b0 = lane_below(y0), b1 = lane_below(y1);
                        lan[0] = lane == 0 ? la2[0] : b0;
                        lan[1] = lane == 0 ? la2[1] : b1;
                        la2[0] = lane_value(y0, cnt - 1);
                        la2[1] = lane_value(y1, cnt - 1);
                    }
                    if (on) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            LAM[(k + 1) * nx + 6 + i] = lan[i];
                            double t = cu[i];
#pragma unroll
                            for (int e = 0; e < 3; ++e)
                                t = __builtin_fma(Uc[e][i], lan[e], t);
                            grad[nu * k + 1 + i] = t + ((2e-4 * uk[1 + i]) + (20.0 * uk[1 + i]));
                        }
                    }
                }
            }
        }
        wave_sync();
        // ---- 5: gradients of the rate inputs (lane = stage; realistic_quadcopter: done by the recursion) ----
        if constexpr (M::angles_are_integrated_inputs) {
            for (int k = lane; k < N; k += WAVE) {
                const double *R = scr + 18 * k;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double sga = M::c6 * LAM[(k + 1) * nx + 6 + j];
                    const double kb4 = sga;
                    const double kb3 = madd(M::Ts, R[6 * j], 2.0 * sga);
                    const double kb2 = madd(M::hTs, R[6 * j + 1], 2.0 * sga);
                    const double kb1 = madd(M::hTs, R[6 * j + 2], sga);
                    const double uj  = u[nu * k + 1 + j];
                    grad[nu * k + 1 + j] = (((kb4 + kb3) + kb2) + kb1) + ((2e-4 * uj) + (20.0 * uj));
                }
            }
            wave_sync();
        }
    }

    /// Adjoint sweep over the stored trajectory and trig store with multipliers yh: grad_i; the adjoints
    /// lam_{k+1} every stage receives are left in LAM (which shares the tangent trajectory's LDS) for the
    /// second-order recording.
    template <bool STORE_LAM>
    __device__ __forceinline__ void reverse(const double *u, const double *yh, double *grad) const {
        reverse_cascade(u, yh, grad);
    }

    // ---- the interface pantr_inner() drives ----------------------------------------------------
    __device__ __forceinline__ double eval_psi_grad(const double *u, double *grad) const {
        double *yh = mvec(MV_E2); // scratch (e2 is rewritten at the end of the inner solve)
        double psi = forward(u, yh);
        reverse<false>(u, yh, grad);
        return psi;
    }
    __device__ __forceinline__ double eval_psi_xhat(const double *u, int yh_slot) const {
        return forward(u, mvec(MV_YH + yh_slot));
    }
    /// yhat_k(x_k) on dual numbers (the value part repeats forward()'s arithmetic).
    __device__ __forceinline__ void stage_multipliers(const typename M::template Trig<dual> &t1,
                                                      const dual (&s)[nx], int k, dual (&yk)[nc]) const {
        const double *yS = mvec(MV_YS), *Sig = mvec(MV_SIG);
        dual c[nc];
        M::constr(t1, s, c);
#pragma unroll
        for (int i = 0; i < nc; ++i) {
            int j     = nc * k + i;
            dual zeta = c[i] + yS[j];
            dual zhat = tmax(tmin(zeta, box->d_ub[i]), box->d_lb[i]);
            dual d    = zeta - zhat;
            yk[i]     = Sig[j] * d;
        }
    }
    /// grad psi at the point eval_psi_xhat(., yh_slot) was last called at (trajectory, multiplier
    /// estimates and trig store still in LDS), and the second-order tape there, in two steps:
    ///   1. the ordinary adjoint sweep (stage after stage, values the same on every lane): grad and the
    ///      adjoints lam_{k+1} of every stage;
    ///   2. the tape: with lam_{k+1} known, the stage Jacobian J_k and Hessian H_k of every stage are
    ///      independent of each other.  One (stage k, direction d of z = (x_k, u_k)) pair per lane: the
    ///      stage's step and adjoint on dual numbers seeded with e_d give column d of J_k (tangents of
    ///      x_next) and of H_k (tangents of the adjoint outputs) - N (nx + nu) independent evaluations,
    ///      64 at a time, instead of one sweep of N dependent ones.
    /// The arithmetic of every entry is the one of the sweep it replaces (value parts = step 1).
    __device__ __forceinline__ void eval_grad_xhat(const double *u, double *grad, int yh_slot) const {
        reverse<true>(u, mvec(MV_YH + yh_slot), grad);
        int pitch = tape_pitch(N);
        if constexpr (NJ_ > 0)
            pitch = FIXED_PITCH;
        else
            asm volatile("" : "+s"(pitch)); // see hess_prod()
        // terminal block: lanes 0..2 seed the position directions of x_N, lanes 3..5 the attitude's
        if (lane < 6) {
            const int blk = lane < 3 ? 0 : 6, d = lane < 3 ? lane : lane + 3;
            dual s[nx], yk[nc], lt[nx], gu[nu];
#pragma unroll
            for (int l = 0; l < nx; ++l) {
                s[l]  = {X[N * nx + l], d == l ? 1.0 : 0.0};
                lt[l] = dual{};
            }
            const auto tN = M::trig_load(TG + 12 * N, s[6].t, s[7].t, s[8].t);
            stage_multipliers(tN, s, N, yk);
            M::cost_grad(s, static_cast<const dual *>(nullptr), lt, gu);
            M::constr_adj(tN, s, yk, lt);
            double *HN = tape + (size_t)NCOEF * pitch + 3 * lane; // HNpp rows, then HNaa rows
#pragma unroll
            for (int e = 0; e < 3; ++e)
                HN[e] = (blk == 0 ? lt[e] : lt[6 + e]).t;
        }
        // position block of every stage: one (stage, position direction) per lane
        for (int base = 0; base < 3 * N; base += WAVE) {
            const int task = base + lane;
            if (task < 3 * N) {
                const int k = task / 3, d = task - 3 * k;
                dual s[nx], uk[nu], yk[nc], lt[nx], gu[nu];
#pragma unroll
                for (int l = 0; l < nx; ++l) {
                    s[l]  = {X[k * nx + l], d == l ? 1.0 : 0.0};
                    lt[l] = dual{};
                }
#pragma unroll
                for (int c = 0; c < nu; ++c)
                    uk[c] = {u[nu * k + c], 0.0};
                const auto t1 = M::trig_load(TG + 12 * k, s[6].t, s[7].t, s[8].t);
                stage_multipliers(t1, s, k, yk);
                M::cost_grad(s, uk, lt, gu);
                M::constr_adj(t1, s, yk, lt);
#pragma unroll
                for (int e = 0; e < 3; ++e)
                    tape[(size_t)(C_HPP + 3 * d + e) * pitch + k] = lt[e].t;
            }
        }
        // the (a, u) directions: one (stage, direction) per lane
        const int tasks = N * NAU;
        for (int base = 0; base < tasks; base += WAVE) {
            const int task  = base + lane;
            const bool live = task < tasks;
            const int tt    = live ? task : tasks - 1; // idle lanes of the last round repeat a task, unstored
            const int k = tt / NAU, l = tt - k * NAU, d = 6 + l;
            dual s[nx], uk[nu], xn[nx], yk[nc], mu[nx], sbar[nx], ubar[nu], gu[nu];
#pragma unroll
            for (int i = 0; i < nx; ++i) {
                s[i]  = {X[k * nx + i], d == i ? 1.0 : 0.0};
                mu[i] = {Xt[(k + 1) * nx + i], 0.0}; // lam_{k+1} (reverse_cascade)
            }
#pragma unroll
            for (int c = 0; c < nu; ++c)
                uk[c] = {u[nu * k + c], d == nx + c ? 1.0 : 0.0};
            const double2 *tg = TG + 12 * k;
            double *col       = tape + k;
            const auto t1     = M::trig_load(tg, s[6].t, s[7].t, s[8].t);
            stage_multipliers(t1, s, k, yk);
            rk4_adj_dual(tg, t1, s, uk, mu, sbar, ubar, xn);
            if (live) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    col[(size_t)(C_JP + NAU * i + l) * pitch] = xn[i].t;
                    col[(size_t)(C_JV + NAU * i + l) * pitch] = xn[3 + i].t;
                    if (l != 3) // the attitude does not depend on the thrust
                        col[(size_t)(C_JA + 6 * i + (l < 3 ? l : l - 1)) * pitch] = xn[6 + i].t;
                }
            }
            M::cost_grad(s, uk, sbar, gu);
#pragma unroll
            for (int c = 0; c < nu; ++c) {
                dual g = ubar[c] + gu[c];
                if (live)
                    col[(size_t)(C_HAU + NAU * l + 3 + c) * pitch] = g.t;
            }
            M::constr_adj(t1, s, yk, sbar);
            if (live)
#pragma unroll
                for (int e = 0; e < 3; ++e)
                    col[(size_t)(C_HAU + NAU * l + e) * pitch] = sbar[6 + e].t;
        }
        wave_sync();
        wait_vmem(); // the products read the tape back through global loads
    }

    /// Hessian-vector product over the tape recorded by eval_grad_xhat(), in the order of the oracle's
    /// SingleShootingAL::hess_prod_tape.  The tangent states and the second-order adjoints are each
    /// "initial value + running sum of increments that do not involve them", so with lane = stage
    /// (blocks of 64 stages, the running totals carried from block to block) a product is: tape rows
    /// into registers (contiguous rows: entry c of stage k at tape[c * pitch + k]), a few dozen
    /// multiply-adds per lane, and wavefront scans (wave_running_sum: 6 DPP levels) - no recursion over
    /// the stages at all for quadcopter.  The backward half runs on the mirrored assignment lane = N - 1 -
    /// stage, so that its sums over the stages ABOVE a stage are running sums over the lanes below again.
    ///   forward   a' = scan(cw w)            | realistic_quadcopter: 3 x 3 recursion, stage by stage (LDS)
    ///             Rv = Jv (a', v_k), Rp = Jp (a', v_k);  v' = scan(Rv);  p' = scan(fma(cpv, v', Rp))
    ///   backward  lp = lp_N + scan(Hpp p');  lv = lv_N + scan(fma(cpv, lp+, hvv v'))
    ///             T = Hau (a', v_k) + Jp' lp+ + Jv' lv+;   Hv[thrust] = T
    ///             la = la_N + scan(T_a), Hv[rates] = fma(cw, la+, T_w)   | realistic: recursion, then T_w + Ja_w' la+
    /// (x+ : the value of stage k + 1.)  LDS: x' in the tangent trajectory rows Xt (forward -> backward
    /// hand-over), for realistic_quadcopter lam_a' in the trajectory rows X and per-stage scratch in the
    /// trig store (W): all rebuilt by the next forward sweep before anything reads them (a product only
    /// ever follows the recording at the same point).
    __device__ __forceinline__ void hess_prod(const double *, const double *v, double *Hv) const {
        if constexpr (SHORT_ == 1) {
            hess_prod_short(v, Hv);
            return;
        } else if constexpr (SHORT_ == 2) {
            if (N <= SHORT_STAGES) {
                hess_prod_short(v, Hv);
                return;
            }
        }
        wave_sync();
        int pitch = tape_pitch(N);
        // the row addresses are loop invariants of the whole solve: computed ahead they would occupy
        // (and spill) a few hundred registers for the lifetime of the kernel; they are scalar work of a
        // few dozen cycles per product instead
        if constexpr (NJ_ > 0)
            pitch = FIXED_PITCH; // (N <= 64) the rows are at compile-time offsets from the tape's base
        else
            asm volatile("" : "+s"(pitch));
        constexpr bool RATES_ARE_INPUTS = M::angles_are_integrated_inputs;
        constexpr int NJA = RATES_ARE_INPUTS ? 3 : 18, NJ = 6 * NAU, NH = NAU * NAU + 9;
        auto load_ja = [&](int k, double (&ja)[NJA]) {
#pragma unroll
            for (int q = 0; q < NJA; ++q)
                ja[q] = tape[(size_t)(RATES_ARE_INPUTS ? C_JA + 6 * q + 3 + q : C_JA + q) * pitch + k];
        };
        auto load_j = [&](int k, double (&cj)[NJ]) { // Jp rows, then Jv rows
#pragma unroll
            for (int q = 0; q < NJ; ++q)
                cj[q] = tape[(size_t)(C_JP + q) * pitch + k];
        };
        auto load_h = [&](int k, double (&ch)[NH]) { // Hau rows, then Hpp rows
#pragma unroll
            for (int q = 0; q < NH; ++q)
                ch[q] = tape[(size_t)(C_HAU + q) * pitch + k];
        };
        // running sum over the stages of a block: returns what the stages before this lane's add up to
        // (carry included) and moves the carry on to the next block; `cnt` stages in the block
        auto scan = [&](double inc, double &carry, int cnt) {
            const double s   = wave_running_sum(inc);
            const double out = carry + lane_below(s);
            carry            = carry + lane_value(s, cnt - 1);
            return out;
        };

        // (phi, theta) of realistic_quadcopter: y <- A y + b over the stages; returns y of the stages before this
        // lane's (the block's initial value on lane 0) and moves the carry on
        auto scan2 = [&](const Affine2 &m, double (&carry)[2], int cnt, double &o0, double &o1) {
            const Affine2 P = wave_affine_scan(m);
            double y0, y1;
            P.apply(carry[0], carry[1], y0, y1); // y after this lane's stage
            const double b0 = lane_below(y0), b1 = lane_below(y1);
            o0       = lane == 0 ? carry[0] : b0;
            o1       = lane == 0 ? carry[1] : b1;
            carry[0] = lane_value(y0, cnt - 1);
            carry[1] = lane_value(y1, cnt - 1);
        };
        // ---- forward: lane = stage ----
        double ca[3] = {0.0, 0.0, 0.0}, cv[3] = {0.0, 0.0, 0.0}, cp[3] = {0.0, 0.0, 0.0}, ca2[2] = {0.0, 0.0};
        for (int k0 = 0; k0 < N; k0 += WAVE) {
            const int cnt = N - k0 < WAVE ? N - k0 : WAVE;
            const bool on = lane < cnt;
            const int k   = on ? k0 + lane : N - 1; // idle lanes shadow the last stage; they add 0 and store nothing
            double cj[NJ], z[NAU];
            load_j(k, cj);
            z[3] = v[nu * k]; z[4] = v[nu * k + 1]; z[5] = v[nu * k + 2]; z[6] = v[nu * k + 3];
            if constexpr (RATES_ARE_INPUTS) {
                double cw[NJA];
                load_ja(k, cw);
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    z[i] = scan(on ? cw[i] * z[4 + i] : 0.0, ca[i], cnt);
            } else {
                double ja[NJA], sw[3];
                load_ja(k, ja);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    sw[i] = ja[6 * i + 3] * z[4];
                    sw[i] = __builtin_fma(ja[6 * i + 4], z[5], sw[i]);
                    sw[i] = __builtin_fma(ja[6 * i + 5], z[6], sw[i]);
                }
                scan2(on ? Affine2{ja[0], ja[1], ja[6], ja[7], sw[0], sw[1]} : affine_identity(), ca2, cnt, z[0], z[1]);
                z[2] = scan(on ? __builtin_fma(ja[13], z[1], __builtin_fma(ja[12], z[0], sw[2])) : 0.0, ca[2], cnt);
            }
            double pt[3], vt[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double rp = cj[NAU * i] * z[0], rv = cj[3 * NAU + NAU * i] * z[0];
#pragma unroll
                for (int e = 1; e < NAU; ++e) {
                    rp = __builtin_fma(cj[NAU * i + e], z[e], rp);
                    rv = __builtin_fma(cj[3 * NAU + NAU * i + e], z[e], rv);
                }
                vt[i] = scan(on ? rv : 0.0, cv[i], cnt);
                pt[i] = scan(on ? __builtin_fma(cpv, vt[i], rp) : 0.0, cp[i], cnt);
            }
            if (on) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    Xt[k * nx + i]     = pt[i];
                    Xt[k * nx + 3 + i] = vt[i];
                    Xt[k * nx + 6 + i] = z[i];
                }
            }
        }
        if constexpr (!RATES_ARE_INPUTS) {
            ca[0] = ca2[0];
            ca[1] = ca2[1];
        }
        if (lane < 3) { // x'_N: the totals
            Xt[N * nx + lane]     = lane == 0 ? cp[0] : lane == 1 ? cp[1] : cp[2];
            Xt[N * nx + 3 + lane] = lane == 0 ? cv[0] : lane == 1 ? cv[1] : cv[2];
            Xt[N * nx + 6 + lane] = lane == 0 ? ca[0] : lane == 1 ? ca[1] : ca[2];
        }
        wave_sync();
        // ---- lam'_N (same values on every lane) ----
        double lp[3], lv[3], la[3];
        {
            const double *HN = tape + (size_t)NCOEF * pitch;
            const double *xN = Xt + N * nx;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                lp[i] = HN[3 * i] * xN[0];
                lp[i] = __builtin_fma(HN[3 * i + 1], xN[1], lp[i]);
                lp[i] = __builtin_fma(HN[3 * i + 2], xN[2], lp[i]);
                lv[i] = hvvN * xN[3 + i];
                la[i] = HN[9 + 3 * i] * xN[6];
                la[i] = __builtin_fma(HN[9 + 3 * i + 1], xN[7], la[i]);
                la[i] = __builtin_fma(HN[9 + 3 * i + 2], xN[8], la[i]);
            }
        }
        double la2[2] = {la[0], la[1]};
        // ---- backward: lane = N - 1 - stage ----
        for (int j0 = 0; j0 < N; j0 += WAVE) {
            const int cnt = N - j0 < WAVE ? N - j0 : WAVE;
            const bool on = lane < cnt;
            const int k   = on ? N - 1 - (j0 + lane) : 0;
            double ch[NH], cj[NJ], z[NAU], xp[3], xv[3], lpn[3], lvn[3];
            load_h(k, ch);
            load_j(k, cj);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                xp[i] = Xt[k * nx + i];
                xv[i] = Xt[k * nx + 3 + i];
                z[i]  = Xt[k * nx + 6 + i];
            }
            z[3] = v[nu * k]; z[4] = v[nu * k + 1]; z[5] = v[nu * k + 2]; z[6] = v[nu * k + 3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double q = ch[NAU * NAU + 3 * i] * xp[0];
                q        = __builtin_fma(ch[NAU * NAU + 3 * i + 1], xp[1], q);
                q        = __builtin_fma(ch[NAU * NAU + 3 * i + 2], xp[2], q);
                lpn[i]   = scan(on ? q : 0.0, lp[i], cnt); // lp of stage k + 1
                lvn[i]   = scan(on ? __builtin_fma(cpv, lpn[i], hvv * xv[i]) : 0.0, lv[i], cnt);
            }
            double T[NAU];
#pragma unroll
            for (int l = 0; l < NAU; ++l) {
                double t = ch[NAU * l] * z[0];
#pragma unroll
                for (int e = 1; e < NAU; ++e)
                    t = __builtin_fma(ch[NAU * l + e], z[e], t);
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    t = __builtin_fma(cj[NAU * i + l], lpn[i], t);
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    t = __builtin_fma(cj[3 * NAU + NAU * i + l], lvn[i], t);
                T[l] = t;
            }
            if (on)
                Hv[nu * k] = T[3];
            if constexpr (RATES_ARE_INPUTS) {
                double cw[NJA];
                load_ja(k, cw);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double lan = scan(on ? T[i] : 0.0, la[i], cnt);
                    if (on)
                        Hv[nu * k + 1 + i] = __builtin_fma(cw[i], lan, T[4 + i]);
                }
            } else {
                double ja[NJA], lan[3];
                load_ja(k, ja);
                lan[2] = scan(on ? T[2] : 0.0, la[2], cnt); // the yaw does not feed back: a running sum
                scan2(on ? Affine2{ja[0], ja[6], ja[1], ja[7], __builtin_fma(ja[12], lan[2], T[0]),
                                   __builtin_fma(ja[13], lan[2], T[1])}
                         : affine_identity(),
                      la2, cnt, lan[0], lan[1]);
                if (on) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        double t = T[4 + j];
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                            t = __builtin_fma(ja[6 * i + 3 + j], lan[i], t);
                        Hv[nu * k + 1 + j] = t;
                    }
                }
            }
        }
        wave_sync();
    }
    /// The product for horizons of up to 16 stages (quadcopter; BASELINE.json configs[0] is N = 12), where lane = stage
    /// leaves three quarters of the wavefront idle: lane = (row r, stage k), k = lane & 15, and row r < 3 works on
    /// component r of every 3-vector (row 3 shadows row 2 and stores nothing but the thrust entry).  One running sum
    /// inside the rows (four DPP levels) then serves all three components - 6 scans of 4 levels instead of 18 of 6 -,
    /// a lane evaluates one row of Jp, Jv, Hpp and two of the seven rows of T instead of all of them, and reads a
    /// third of the tape rows; the components of a vector reach the other rows through permlane swaps (three
    /// gathers).  Every value is computed by the expression hess_prod() below uses for it, and the in-row sums are the
    /// additions of wave_running_sum() on lanes 0..15: same bits (tests/test_gpu_quadcopter.py at N = 1, 7, 12).
    static constexpr int SHORT_STAGES = 16;
    __device__ __forceinline__ void hess_prod_short(const double *v, double *Hv) const {
        wave_sync();
        constexpr int pitch = FIXED_PITCH; // (N <= 16 < 64)
        const int r = lane >> 4, kk = lane & 15, i = r < 3 ? r : 2;
        const bool on = kk < N, row = r < 3;
        auto T = [&](int c, int k) { return tape[(size_t)c * pitch + k]; };
        // ---- forward: stage kk ----
        // (the tape entries of a half into registers first, as in hess_prod(): requested together, one wait)
        {
            const int k = on ? kk : N - 1; // idle lanes shadow the last stage; they add 0 and store nothing
            double jp[NAU], jv[NAU];
#pragma unroll
            for (int e = 0; e < NAU; ++e) {
                jp[e] = T(C_JP + NAU * i + e, k);
                jv[e] = T(C_JV + NAU * i + e, k);
            }
            const double cw = T(C_JA + 6 * i + 3 + i, k);
            double z[NAU];
            z[3] = v[nu * k]; z[4] = v[nu * k + 1]; z[5] = v[nu * k + 2]; z[6] = v[nu * k + 3];
            const double zi = i == 0 ? z[4] : i == 1 ? z[5] : z[6];
            const double sa = row_running_sum(on ? cw * zi : 0.0);
            const double at = 0.0 + row_from_prev(sa); // a'_i of this stage: what the stages before add up to
            rows_gather3(at, z[0], z[1], z[2]);
            double rp = jp[0] * z[0], rv = jv[0] * z[0];
#pragma unroll
            for (int e = 1; e < NAU; ++e) {
                rp = __builtin_fma(jp[e], z[e], rp);
                rv = __builtin_fma(jv[e], z[e], rv);
            }
            const double sv = row_running_sum(on ? rv : 0.0);
            const double vt = 0.0 + row_from_prev(sv);
            const double sp = row_running_sum(on ? __builtin_fma(cpv, vt, rp) : 0.0);
            const double pt = 0.0 + row_from_prev(sp);
            if (on && row) {
                Xt[k * nx + i]     = pt;
                Xt[k * nx + 3 + i] = vt;
                Xt[k * nx + 6 + i] = at;
                if (kk == N - 1) { // x'_N: the totals
                    Xt[N * nx + i]     = 0.0 + sp;
                    Xt[N * nx + 3 + i] = 0.0 + sv;
                    Xt[N * nx + 6 + i] = 0.0 + sa;
                }
            }
        }
        wave_sync();
        // ---- lam'_N, component i ----
        double lp, lv, la;
        {
            const double *HN = tape + (size_t)NCOEF * pitch;
            const double *xN = Xt + N * nx;
            lp = HN[3 * i] * xN[0];
            lp = __builtin_fma(HN[3 * i + 1], xN[1], lp);
            lp = __builtin_fma(HN[3 * i + 2], xN[2], lp);
            lv = hvvN * xN[3 + i];
            la = HN[9 + 3 * i] * xN[6];
            la = __builtin_fma(HN[9 + 3 * i + 1], xN[7], la);
            la = __builtin_fma(HN[9 + 3 * i + 2], xN[8], la);
        }
        // ---- backward: stage N - 1 - kk ----
        {
            const int k = on ? N - 1 - kk : 0;
            // rows r (attitude row of rows 0..2, the thrust row of row 3) and 4 + i (rate row) of Hau, the matching
            // columns of Jp and Jv, row i of Hpp
            double ha[NAU], hw[NAU], pa[3], pw[3], va[3], vw[3], hp[3];
#pragma unroll
            for (int e = 0; e < NAU; ++e) {
                ha[e] = T(C_HAU + NAU * r + e, k);
                hw[e] = T(C_HAU + NAU * (4 + i) + e, k);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                pa[j] = T(C_JP + NAU * j + r, k);
                pw[j] = T(C_JP + NAU * j + 4 + i, k);
                va[j] = T(C_JV + NAU * j + r, k);
                vw[j] = T(C_JV + NAU * j + 4 + i, k);
                hp[j] = T(C_HPP + 3 * i + j, k);
            }
            const double cw = T(C_JA + 6 * i + 3 + i, k);
            double z[NAU], xp[3], lpn[3], lvn[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                xp[j] = Xt[k * nx + j];
                z[j]  = Xt[k * nx + 6 + j];
            }
            const double xv = Xt[k * nx + 3 + i];
            z[3] = v[nu * k]; z[4] = v[nu * k + 1]; z[5] = v[nu * k + 2]; z[6] = v[nu * k + 3];
            double q = hp[0] * xp[0];
            q        = __builtin_fma(hp[1], xp[1], q);
            q        = __builtin_fma(hp[2], xp[2], q);
            const double lpn_i = lp + row_from_prev(row_running_sum(on ? q : 0.0)); // lp of stage k + 1
            const double lvn_i = lv + row_from_prev(row_running_sum(on ? __builtin_fma(cpv, lpn_i, hvv * xv) : 0.0));
            rows_gather3(lpn_i, lpn[0], lpn[1], lpn[2]);
            rows_gather3(lvn_i, lvn[0], lvn[1], lvn[2]);
            // a row of T = Hau (a', v_k) + Jp' lp+ + Jv' lv+
            auto Trow = [&](const double (&h)[NAU], const double (&jpc)[3], const double (&jvc)[3]) {
                double t = h[0] * z[0];
#pragma unroll
                for (int e = 1; e < NAU; ++e)
                    t = __builtin_fma(h[e], z[e], t);
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    t = __builtin_fma(jpc[j], lpn[j], t);
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    t = __builtin_fma(jvc[j], lvn[j], t);
                return t;
            };
            const double Ta = Trow(ha, pa, va), Tw = Trow(hw, pw, vw);
            if (on && !row)
                Hv[nu * k] = Ta; // T[3]
            const double lan = la + row_from_prev(row_running_sum(on ? Ta : 0.0));
            if (on && row)
                Hv[nu * k + 1 + i] = __builtin_fma(cw, lan, Tw);
        }
        wave_sync();
    }

    // ---- speculation on a rejected candidate (SPEC: two-wavefront workgroups, pantr_alm_spec_kernel) ----
    // As for the chain (pantr_kernel.hip.h, SpecBox): while the solver wavefront runs the CG loop and
    // evaluates the candidate, the helper evaluates what a rejection needs next - psi(xhat) of the prox
    // iterate with its multiplier estimates, the quadratic-upper-bound test, grad_L(xhat) and the tape
    // there - on its own trajectory, trig store, scratch and tape buffer.  It reads xhat from the solver's
    // iterate vectors and y, Sigma, y/Sigma from the shared m-vectors (all constant meanwhile) and writes
    // yhat to the m-vector slot of the prox iterate (which nothing else touches meanwhile), the gradient
    // to a vector of its own.  No model-value requests here: a request keeps the helper busy until the
    // ratio test.
    SpecBox *spec = nullptr;
    double *hG    = nullptr; ///< LDS: the helper's gradient vector
    double *tape0 = nullptr; ///< HBM: the first of this instance's two tape buffers
    int tape_cur  = 0;       ///< which of them `tape` is
    bool spec_pending = false, spec_dead = false;
    int spec_timeouts = 0; ///< results this wavefront stopped waiting for (stats.spec_timeouts)
    double spec_tick_s = 1e-8, spec_patience_s = 1.0;
    double spec_give_up_s = 1.0; ///< the helper's own deadline (spec_serve): how long an ended helper is waited for at most

    __device__ __forceinline__ void spec_request(int xh_slot, double psi_x, double gradTp, double pTp, double Lip,
                                                 const KernelParams &P, int yh_slot) {
        if (!spec || spec_dead)
            return;
        wave_sync(); // xhat is written
        if (lane == 0) {
            spec->cmd = SPEC_EVAL; spec->xh_slot = xh_slot; spec->yh_slot = yh_slot; spec->tape_sel = 1 - tape_cur;
            spec->psi_x = psi_x; spec->gradTp = gradTp; spec->pTp = pTp; spec->L = Lip;
            spec->qub_factor = P.quadratic_upperbound_tolerance_factor; spec->L_max = P.L_max;
            __hip_atomic_store(&spec->req_seq, spec->req_seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        spec_pending = true;
    }
    /// Wait for the posted request (bounded work; the wait is bounded too: false = nothing of it is used).
    __device__ __forceinline__ bool spec_wait() {
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
                if (lane == 0)
                    __hip_atomic_store(&spec->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                // The helper of this problem family writes into what the solver is about to use again - yhat goes to
                // the prox iterate's m-vector slot, and it reads xhat, y, Sigma - and looks at the flag only between
                // its sweeps: the solver goes on alone only once the helper has said that it ended (at most one sweep
                // away; bounded like the helper's own life, should the wavefront never run again).
                const long long t1 = wall_clock64();
                while (__hip_atomic_load(&spec->stopped, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0 &&
                       (double)(wall_clock64() - t1) * spec_tick_s < spec_give_up_s)
                    __builtin_amdgcn_s_sleep(2);
                return false;
            }
        }
        return true;
    }
    __device__ __forceinline__ void spec_drop() {
        if (spec && spec_pending)
            spec_wait();
        spec_pending = false;
    }
    __device__ __forceinline__ bool spec_take(double &psi_xh) {
        if (!spec || !spec_pending)
            return false;
        const bool done = spec_wait();
        spec_pending    = false;
        const bool ok   = done && uniform(spec->ok) != 0;
        if (ok)
            psi_xh = uniform(spec->psi_xh);
        return ok;
    }
    __device__ __forceinline__ bool spec_post_model(const double *, const double *, unsigned) { return false; }
    __device__ __forceinline__ bool spec_take_model(double &) { return false; }
    /// grad <- the helper's gradient, tape <- the buffer it recorded into (yhat is in the iterate's slot already).
    __device__ __forceinline__ void spec_adopt(double *grad) {
        wave_sync();
        if constexpr (NJ > 0) {
            Owned<NJ>::load(hG, lane).store(grad, lane);
        } else {
            for (int i = lane; i < n; i += WAVE)
                grad[i] = hG[i];
        }
        tape_cur = 1 - tape_cur;
        tape     = tape0 + (size_t)tape_cur * workspace_doubles(N);
        wave_sync();
    }
    /// End of an inner solve: the helper idles until the next one (set_al / update_multipliers write the
    /// m-vectors it reads).
    __device__ __forceinline__ void spec_exit() { spec_drop(); }
    /// End of the instance.
    __device__ __forceinline__ void spec_shutdown() {
        if (!spec)
            return;
        spec_drop();
        if (lane == 0) {
            spec->cmd = SPEC_EXIT;
            __hip_atomic_store(&spec->req_seq, spec->req_seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    /// The helper wavefront (`this`: its own X, Xt, TG, scr, coef; the solver's vectors and m-vectors).
    __device__ __forceinline__ void spec_serve(long long t_start, double tick_s, double give_up_s) {
        spec_serve_loop(t_start, tick_s, give_up_s);
        wave_sync(); // whatever this wavefront wrote last is written
        if (lane == 0)
            __hip_atomic_store(&spec->stopped, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ void spec_serve_loop(long long t_start, double tick_s, double give_up_s) {
        int seen = 0;
        auto aborted = [&]() {
            return uniform(__hip_atomic_load(&spec->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0;
        };
        while (true) {
            int r;
            while ((r = __hip_atomic_load(&spec->req_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == seen) {
                __builtin_amdgcn_s_sleep(8);
                if (aborted() || (double)(wall_clock64() - t_start) * tick_s > give_up_s)
                    return;
            }
            seen = r;
            if (uniform(spec->cmd) == SPEC_EXIT || aborted())
                return;
            const double *xh = ivec(uniform(spec->xh_slot));
            const int yh     = uniform(spec->yh_slot);
            tape             = tape0 + (size_t)uniform(spec->tape_sel) * workspace_doubles(N);
            const double psi_xh = eval_psi_xhat(xh, yh);
            const double psi_x = uniform(spec->psi_x), gradTp = uniform(spec->gradTp), pTp = uniform(spec->pTp),
                         Lip = uniform(spec->L);
            const double margin = (1.0 + fabs(psi_x)) * uniform(spec->qub_factor);
            const bool violated = psi_xh > ((psi_x + gradTp) + (0.5 * Lip) * pTp) + margin;
            const bool ok       = !(Lip < uniform(spec->L_max) && violated);
            if (aborted()) // the solver stopped waiting (and evaluates this itself): nothing of the result is used
                return;
            if (ok)
                eval_grad_xhat(xh, hG, yh);
            wave_sync(); // gradient, yhat and (acknowledged) tape are written before the result is announced
            if (lane == 0) {
                spec->psi_xh = psi_xh;
                spec->ok     = ok ? 1 : 0;
                __hip_atomic_store(&spec->done_seq, seen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    /// LDS doubles of the helper: trajectory, adjoint rows, trig store, rate-adjoint coefficients, sweep
    /// scratch of its own (the solver's is lent from its CG vectors, which are busy meanwhile), gradient.
    __host__ __device__ static size_t helper_lds_doubles(int N, int npad) {
        return (size_t)2 * (N + 1) * nx + (size_t)(N + 1) * M::TRIG_DOUBLES_PER_STAGE + coef_doubles(N) +
               scratch_doubles(N) + (size_t)npad;
    }

    /// y / Sigma for the sweeps; call whenever y or Sigma changed (they are fixed during an inner solve).
    __device__ __forceinline__ void set_al() const {
        wave_sync();
        const double *y = mvec(MV_Y), *Sig = mvec(MV_SIG);
        double *yS = mvec(MV_YS);
        for (int j = lane; j < m; j += WAVE)
            yS[j] = y[j] / Sig[j];
        wave_sync();
    }
    /// err_z = Sigma^{-1} (yhat - y) into e2, then y <- yhat.
    __device__ __forceinline__ void update_multipliers(int yh_slot) const {
        wave_sync();
        double *y = mvec(MV_Y), *e2 = mvec(MV_E2);
        const double *Sig = mvec(MV_SIG), *yh = mvec(MV_YH + yh_slot);
        for (int j = lane; j < m; j += WAVE) {
            e2[j] = (yh[j] - y[j]) / Sig[j];
            y[j]  = yh[j];
        }
        wave_sync();
    }

    // ---- vector helpers over the LDS n-vectors (same reduction order as the chain problem) -------
    /// The bounds of u as n-vectors (lbv, ubv); once per instance, after the LDS was cleared.
    __device__ __forceinline__ void fill_bounds() const {
        for (int i = lane; i < n; i += WAVE) {
            lbv[i] = box_pick(box->u_lb, i % nu);
            ubv[i] = box_pick(box->u_ub, i % nu);
        }
        wave_sync();
    }
    using OV = Owned<(NJ > 0 ? NJ : 1)>;
    __device__ __forceinline__ double dot(const double *a, const double *b) const {
        double acc = 0.0;
        if constexpr (NJ > 0) {
            const OV av = OV::load(a, lane), bv = OV::load(b, lane);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc = acc + av[j] * bv[j];
        } else {
            for (int j = 0; j < nj; ++j) {
                int i = j * WAVE + lane;
                if (i < n)
                    acc = acc + a[i] * b[i];
            }
        }
        return uniform(wave_sum(acc));
    }
    __device__ __forceinline__ double dot_masked(const double *a, const double *b, uint32_t mask) const {
        double acc = 0.0;
        if constexpr (NJ > 0) {
            const OV av = OV::load(a, lane), bv = OV::load(b, lane);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc = acc + av[j] * ((mask >> j) & 1u ? bv[j] : 0.0);
        } else {
            for (int j = 0; j < nj; ++j) {
                int i = j * WAVE + lane;
                if (i < n)
                    acc = acc + a[i] * ((mask >> j) & 1u ? b[i] : 0.0);
            }
        }
        return uniform(wave_sum(acc));
    }
    __device__ __forceinline__ double norm_inf(const double *a) const {
        double r = 0.0;
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + lane;
            if (i < n)
                r = fmax(r, fabs(a[i]));
        }
        return uniform(wave_max(r));
    }
    __device__ __forceinline__ void prox_grad_step(double gamma, const double *x, const double *grad,
                                                   double *xh, double *p, double &pTp, double &gradTp) const {
        double a1 = 0.0, a2 = 0.0;
        if constexpr (NJ > 0) {
            const OV xv = OV::load(x, lane), gv = OV::load(grad, lane), lo = OV::load(lbv, lane),
                     hi = OV::load(ubv, lane);
            OV pv, xhv;
#pragma unroll
            for (int j = 0; j < NJ; ++j) { // (padding: x = grad = lb = ub = 0, so p = 0 and xhat = 0)
                double t = (-gamma) * gv[j];
                pv[j]    = fmin(fmax(t, lo[j] - xv[j]), hi[j] - xv[j]);
                xhv[j]   = xv[j] + pv[j];
                a1       = a1 + pv[j] * pv[j];
                a2       = a2 + pv[j] * gv[j];
            }
            pv.store(p, lane);
            xhv.store(xh, lane);
            wave_sum2(a1, a2);
            pTp    = uniform(a1);
            gradTp = uniform(a2);
        } else {
            for (int j = 0; j < nj; ++j) {
                int i = j * WAVE + lane;
                if (i < n) {
                    double xi = x[i], gi = grad[i];
                    double t  = (-gamma) * gi;
                    double pi = fmin(fmax(t, lbv[i] - xi), ubv[i] - xi);
                    p[i]      = pi;
                    xh[i]     = xi + pi;
                    a1        = a1 + pi * pi;
                    a2        = a2 + pi * gi;
                }
            }
            pTp    = uniform(wave_sum(a1));
            gradTp = uniform(wave_sum(a2));
        }
    }
    __device__ __forceinline__ double proj_grad_unit_norm(const double *x, const double *grad) const {
        double r = 0.0;
        if constexpr (NJ > 0) {
            const OV xv = OV::load(x, lane), gv = OV::load(grad, lane), lo = OV::load(lbv, lane),
                     hi = OV::load(ubv, lane);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                double t  = (-1.0) * gv[j];
                double pi = fmin(fmax(t, lo[j] - xv[j]), hi[j] - xv[j]);
                r         = fmax(r, fabs(pi));
            }
        } else {
            for (int j = 0; j < nj; ++j) {
                int i = j * WAVE + lane;
                if (i < n) {
                    double xi = x[i];
                    double t  = (-1.0) * grad[i];
                    double pi = fmin(fmax(t, lbv[i] - xi), ubv[i] - xi);
                    r         = fmax(r, fabs(pi));
                }
            }
        }
        return uniform(wave_max(r));
    }
    __device__ __forceinline__ uint32_t inactive_mask(double gamma, const double *x, const double *grad,
                                                      int &nJ) const {
        uint32_t mask = 0;
        int cnt       = 0;
        if constexpr (NJ > 0) {
            const OV xv = OV::load(x, lane), gv = OV::load(grad, lane), lo = OV::load(lbv, lane),
                     hi = OV::load(ubv, lane);
#pragma unroll
            for (int j = 0; j < NJ; ++j) { // (padding: 0 < 0 is false)
                double xfw = xv[j] - gamma * gv[j];
                bool in    = lo[j] < xfw && xfw < hi[j];
                mask |= (in ? 1u : 0u) << j;
                cnt += __builtin_popcountll(__ballot(in ? 1 : 0));
            }
        } else {
            for (int j = 0; j < nj; ++j) {
                int i   = j * WAVE + lane;
                bool in = false;
                if (i < n) {
                    double xfw = x[i] - gamma * grad[i];
                    in = lbv[i] < xfw && xfw < ubv[i];
                }
                mask |= (in ? 1u : 0u) << j;
                cnt += __builtin_popcountll(__ballot(in ? 1 : 0)); // wave-uniform count, no reduction needed
            }
        }
        nJ = cnt;
        return mask;
    }
};

/// One instance, start to finish, on the calling wavefront: the body of both ALM kernels.
template <class M, int METHOD, class PB>
__device__ __forceinline__ void alm_solve_instance(const AlmSolveArgs &a, PB &pb, int lane, int inst, long long t_start,
                                                   double *x0_lds) {
    const KernelParams &P    = a.prm;
    const AlmKernelParams &A = a.alm;
    const int n = a.n, m = a.m;
    if (lane < M::nx) // the problem parameter of this instance (pb.x0 reads it)
        x0_lds[lane] = a.x0[(size_t)inst * M::nx + lane];
    pb.fill_bounds();
    double *x_io = pb.vec(S_XIO), *y = pb.mvec(PB::MV_Y), *Sig = pb.mvec(PB::MV_SIG), *e1 = pb.mvec(PB::MV_E1),
           *e2 = pb.mvec(PB::MV_E2);
    for (int i = lane; i < n; i += WAVE)
        x_io[i] = a.u[(size_t)inst * n + i];
    for (int j = lane; j < m; j += WAVE) {
        y[j]   = a.y[(size_t)inst * m + j];
        Sig[j] = A.initial_penalty;
    }
    wave_sync();

    DeviceStats st{};
    double eps = A.initial_tolerance, norm_e1 = __builtin_nan(""), norm_e2 = 0;
    int status = Busy, outer = 0, num_successful = 0, failures = 0, total_inner = 0;
    InnerResult r{};
    const bool trace = a.trace != nullptr && inst < a.trace_instances;
    for (int i = 0; i < A.max_iter; ++i) {
        // eval_proj_multipliers(y, M): no lower bound -> y <= 0 is impossible ... (y >= 0 side only), etc.
        for (int j = lane; j < m; j += WAVE) {
            int ci     = j % M::nc;
            double ylb = box_pick(P.box.d_lb, ci) == -__builtin_huge_val() ? 0.0 : -A.max_multiplier;
            double yub = box_pick(P.box.d_ub, ci) == __builtin_huge_val() ? 0.0 : A.max_multiplier;
            y[j]       = fmin(fmax(y[j], ylb), yub);
        }
        pb.set_al();
        // every inner solve has solver.max_time of its own, capped by what is left of alm.max_time
        const long long t_inner = wall_clock64();
        const double remaining  = A.max_time_s - (double)(t_inner - t_start) * P.tick_s;
        r = inner_solve<METHOD>(
            pb, P, eps, fmin(P.solver_max_time_s, remaining), t_inner, x_io, st,
            trace && i == 0 ? a.trace + (size_t)inst * a.trace_cap * TRACE_COLS : nullptr, a.trace_cap,
            a.lbfgs ? a.lbfgs + (size_t)inst * 2 * P.lbfgs_memory * a.npad : nullptr);
        total_inner += r.iterations;
        bool inner_converged = r.status == Converged;
        failures += inner_converged ? 0 : 1;
        double mx = 0.0;
        for (int j = lane; j < m; j += WAVE)
            mx = fmax(mx, fabs(e2[j]));
        norm_e2 = uniform(wave_max(mx));
        double elapsed     = (double)(wall_clock64() - t_start) * P.tick_s;
        bool alm_converged = r.eps <= A.tolerance && inner_converged && norm_e2 <= A.dual_tolerance;
        bool out_of_time = elapsed > A.max_time_s, out_of_iter = i + 1 == A.max_iter;
        outer = i + 1;
        if (alm_converged || out_of_iter || out_of_time) {
            status = alm_converged ? Converged : out_of_time ? MaxTime : MaxIter;
            break;
        }
        if (norm_e2 > A.dual_tolerance) { // update_penalty_weights (e2 = current, e1 = previous error)
            bool first = num_successful == 0;
            if (A.single_penalty_factor) {
                if (first || norm_e2 > A.rel_penalty_increase_threshold * norm_e1) {
                    double ns = fmin(A.max_penalty, A.penalty_update_factor * Sig[0]);
                    wave_sync();
                    for (int j = lane; j < m; j += WAVE)
                        Sig[j] = ns;
                }
            } else {
                for (int j = lane; j < m; j += WAVE)
                    if (first || fabs(e2[j]) > A.rel_penalty_increase_threshold * fabs(e1[j]))
                        Sig[j] = fmin(A.max_penalty,
                                      fmax((A.penalty_update_factor * fabs(e2[j])) / norm_e2, 1.0) * Sig[j]);
            }
        }
        eps = fmax(A.tolerance_update_factor * eps, A.tolerance);
        ++num_successful;
        for (int j = lane; j < m; j += WAVE) { // e1 <- e2
            double t = e2[j];
            e2[j]    = e1[j];
            e1[j]    = t;
        }
        norm_e1 = norm_e2;
        wave_sync();
    }
    for (int i = lane; i < n; i += WAVE)
        a.u[(size_t)inst * n + i] = x_io[i];
    for (int j = lane; j < m; j += WAVE)
        a.y[(size_t)inst * m + j] = y[j];
    double sq = 0.0;
    for (int j = lane; j < m; j += WAVE)
        sq = sq + Sig[j] * Sig[j];
    sq = wave_sum(sq);
    if (lane == 0) {
        st.status       = status == Busy ? (int)MaxIter : status;
        st.iterations   = total_inner;
        st.eps          = r.eps;
        st.final_gamma  = r.gamma;
        st.final_psi    = r.psi;
        st.final_fbe    = r.fbe;
        st.final_radius = r.radius;
        st.elapsed_s    = (double)(wall_clock64() - t_start) * P.tick_s;
        st.outer_iterations           = outer;
        st.inner_convergence_failures = failures;
        st.delta                      = norm_e2;
        st.penalty_norm               = __builtin_sqrt(sq / m);
        if constexpr (PB::HAS_SPEC)
            st.spec_timeouts = pb.spec_timeouts;
        write_stats(a.stats + inst, st);
    }
    if constexpr (PB::HAS_SPEC)
        pb.spec_shutdown();
}

/// The solver wavefront's problem object on the workgroup's dynamic LDS (layout: lds_doubles()).  Returns
/// where the instance's initial state goes.
template <class M, int METHOD, class PB>
__device__ __forceinline__ double *alm_bind(PB &pb, const AlmSolveArgs &a, double *lds, int lane, int inst) {
    pb.lane = lane; pb.box = &a.prm.box; pb.lds = lds; pb.N = a.N; pb.n = a.n; pb.npad = a.npad; pb.nj = a.nj;
    pb.m = a.m; pb.mpad = a.mpad; pb.mj = a.mj;
    pb.lbv = lds + (size_t)NUM_VEC * a.npad;
    pb.ubv = pb.lbv + a.npad;
    pb.mv = lds + (size_t)(NUM_VEC + PB::BOX_VECS) * a.npad;
    pb.X  = pb.mv + (size_t)PB::MV_COUNT * a.mpad;
    pb.Xt = pb.X + (size_t)(a.N + 1) * M::nx;
    pb.TG = reinterpret_cast<double2 *>(pb.Xt + (size_t)(a.N + 1) * M::nx);
    double *x0_lds = reinterpret_cast<double *>(pb.TG + 12 * (a.N + 1)); // after the trig store
    pb.x0 = x0_lds;
    // scratch of the cascade sweeps: PANTR lends the five vector slots from z on (dead outside the
    // direction computation); the comparators keep L-BFGS scalars in the CG block and get LDS of their own
    pb.coef = x0_lds + PB::X0_DOUBLES;
    pb.scr  = METHOD == 0 ? pb.vec(PB::CG_BASE + 3) : pb.coef + PB::coef_doubles(a.N);
    return x0_lds;
}

/// ALM outer loop (m > 0) around the inner solver METHOD names, one wavefront per instance.
template <class M, int METHOD, int NJ = 0, int SHORT = 0>
__global__ __launch_bounds__(WAVE) void pantr_alm_kernel(AlmSolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    const int inst = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    if (inst >= a.B)
        return;
    const long long t_start = wall_clock64();
    using PB = ScalarALProblem<M, false, NJ, SHORT>;
    PB pb;
    double *x0_lds = alm_bind<M, METHOD>(pb, a, lds, lane, inst);
    pb.tape = a.ws + (size_t)inst * PB::workspace_doubles(a.N);
    const size_t total = PB::lds_doubles(a.N, a.npad, a.mpad, METHOD != 0);
    for (size_t i = lane; i < total; i += WAVE)
        lds[i] = 0.0;
    wave_sync();
    alm_solve_instance<M, METHOD>(a, pb, lane, inst, t_start, x0_lds);
}

/// method = pantr with a helper wavefront per instance (ScalarALProblem: speculation).  Wavefront 0 is
/// the solver of pantr_alm_kernel<M, 0>; wavefront 1 serves its requests on LDS of its own behind the
/// solver's (helper_lds_doubles) and on the instance's second tape buffer.
template <class M, int NJ = 0, int SHORT = 0>
__global__ __launch_bounds__(2 * WAVE) void pantr_alm_spec_kernel(AlmSolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (WAVE - 1);
    const int inst = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    if (inst >= a.B)
        return;
    const long long t_start = wall_clock64();
    __shared__ SpecBox box;
    using PB = ScalarALProblem<M, true, NJ, SHORT>;
    PB pb;
    double *x0_lds = alm_bind<M, 0>(pb, a, lds, lane, inst);
    const size_t solver_doubles = PB::lds_doubles(a.N, a.npad, a.mpad, false);
    const size_t total          = solver_doubles + PB::helper_lds_doubles(a.N, a.npad);
    for (size_t i = threadIdx.x; i < total; i += 2 * WAVE)
        lds[i] = 0.0;
    if (threadIdx.x == 0) {
        box.req_seq  = 0;
        box.done_seq = 0;
        box.cmd      = 0;
        box.abort    = 0;
        box.stopped  = 0;
    }
    __syncthreads(); // the only workgroup barrier: LDS is cleared and the mailbox exists
    pb.spec        = &box;
    pb.spec_tick_s = a.prm.tick_s;
    pb.spec_patience_s = a.prm.spec_patience_s;
    pb.spec_give_up_s  = 2.0 * a.alm.max_time_s + 1.0;
    pb.tape0       = a.ws + (size_t)inst * 2 * PB::workspace_doubles(a.N);
    pb.tape        = pb.tape0;
    double *h      = lds + solver_doubles; // the helper's LDS
    pb.hG          = h + (size_t)2 * (a.N + 1) * M::nx + (size_t)(a.N + 1) * M::TRIG_DOUBLES_PER_STAGE +
            PB::coef_doubles(a.N) + PB::scratch_doubles(a.N);
    if (wave == 1) {
        pb.X    = h; // (the solver writes the initial state to LDS before its first request)
        pb.Xt   = pb.X + (size_t)(a.N + 1) * M::nx;
        pb.TG   = reinterpret_cast<double2 *>(pb.Xt + (size_t)(a.N + 1) * M::nx);
        pb.coef = reinterpret_cast<double *>(pb.TG + 12 * (a.N + 1));
        pb.scr  = pb.coef + PB::coef_doubles(a.N);
        pb.spec_serve(t_start, a.prm.tick_s, pb.spec_give_up_s);
        return;
    }
    __builtin_amdgcn_s_setprio(3);
    alm_solve_instance<M, 0>(a, pb, lane, inst, t_start, x0_lds);
}

} // namespace pantr

namespace pantr {

// ---------------------------------------------------------------------------------------
// Stand-alone problem-function kernels for the scalar AL problems (row-level parity tests)
// ---------------------------------------------------------------------------------------
enum AlmEvalMode : int32_t { ALM_EVAL_PSI_GRAD = 0, ALM_EVAL_HESS_PROD = 1, ALM_EVAL_GRAD_L = 2, ALM_EVAL_PROX = 3 };

struct AlmEvalArgs {
    int32_t B, N, n, npad, nj, m, mpad, mj, mode, reps; ///< reps: development aid (timing), normally 1
    const double *x0;  ///< [B][nx]
    const double *u;   ///< [B][n]
    const double *v;   ///< [B][n] (HESS_PROD)
    const double *y;   ///< [B][m]
    const double *Sig; ///< [B][m]
    double *out_s;     ///< [B] psi
    double *out_v;     ///< [B][n] grad | Hv
    double *out_m;     ///< [B][m] yhat
    double *ws;        ///< [B][ScalarALProblem::workspace_doubles(N)]
    double *out_v2;    ///< [B][n] PROX: p
    double *out_s4;    ///< [B][4] PROX: p'p, grad'p, ||Pi_C(x - grad) - x||_inf, |J(gamma)|
    double gamma;      ///< PROX: step size (u = x, v = grad)
    BoxData box;
};

template <class M>
__global__ __launch_bounds__(WAVE) void pantr_alm_eval_kernel(AlmEvalArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x, inst = blockIdx.x;
    if (inst >= a.B)
        return;
    using PB = ScalarALProblem<M, false, 0, M::angles_are_integrated_inputs ? 2 : 0>; // (both products, by the horizon)
    PB pb;
    pb.lane = lane; pb.box = &a.box; pb.lds = lds; pb.N = a.N; pb.n = a.n; pb.npad = a.npad; pb.nj = a.nj;
    pb.m = a.m; pb.mpad = a.mpad; pb.mj = a.mj;
    pb.lbv = lds + (size_t)NUM_VEC * a.npad;
    pb.ubv = pb.lbv + a.npad;
    pb.mv = lds + (size_t)(NUM_VEC + PB::BOX_VECS) * a.npad;
    pb.X  = pb.mv + (size_t)PB::MV_COUNT * a.mpad;
    pb.Xt = pb.X + (size_t)(a.N + 1) * M::nx;
    pb.TG = reinterpret_cast<double2 *>(pb.Xt + (size_t)(a.N + 1) * M::nx);
    pb.tape = a.ws + (size_t)inst * PB::workspace_doubles(a.N);
    const size_t total = PB::lds_doubles(a.N, a.npad, a.mpad);
    for (size_t i = lane; i < total; i += WAVE)
        lds[i] = 0.0;
    wave_sync();
    {
        double *x0_lds = reinterpret_cast<double *>(pb.TG + 12 * (a.N + 1)); // after the trig store
        if (lane < M::nx)
            x0_lds[lane] = a.x0[(size_t)inst * M::nx + lane];
        pb.x0 = x0_lds;
        pb.coef = x0_lds + PB::X0_DOUBLES;
        pb.scr  = pb.vec(PB::CG_BASE + 3); // five vector slots this kernel does not use
    }
    pb.fill_bounds();
    double *u = pb.vec(0), *v = pb.vec(1), *o = pb.vec(2);
    for (int i = lane; i < a.n; i += WAVE) {
        u[i] = a.u[(size_t)inst * a.n + i];
        if (a.v)
            v[i] = a.v[(size_t)inst * a.n + i];
    }
    for (int j = lane; j < a.m; j += WAVE) {
        pb.mvec(PB::MV_Y)[j]   = a.y[(size_t)inst * a.m + j];
        pb.mvec(PB::MV_SIG)[j] = a.Sig[(size_t)inst * a.m + j];
    }
    pb.set_al();
    double psi = 0.0;
    if (a.mode == ALM_EVAL_PSI_GRAD) {
        for (int rep = 0; rep < a.reps; ++rep) {
            psi = pb.eval_psi_xhat(u, 0);
            pb.eval_grad_xhat(u, o, 0);
        }
        for (int j = lane; j < a.m; j += WAVE)
            a.out_m[(size_t)inst * a.m + j] = pb.mvec(PB::MV_YH)[j];
    } else if (a.mode == ALM_EVAL_PROX) {
        double pTp, gTp;
        double *p = pb.vec(3);
        pb.prox_grad_step(a.gamma, u, v, o, p, pTp, gTp);
        const double eps = pb.proj_grad_unit_norm(u, v);
        int nJ;
        pb.inactive_mask(a.gamma, u, v, nJ);
        wave_sync();
        for (int i = lane; i < a.n; i += WAVE)
            a.out_v2[(size_t)inst * a.n + i] = p[i];
        if (lane == 0) {
            double *os = a.out_s4 + (size_t)inst * 4;
            os[0] = pTp; os[1] = gTp; os[2] = eps; os[3] = nJ;
        }
    } else if (a.mode == ALM_EVAL_GRAD_L) {
        // grad_x L(x, y) = grad f + grad g' y (formulation.hpp:98,170-174): the trajectory at u, then
        // the adjoint sweep with the caller's y as the multipliers (Sigma plays no role)
        for (int rep = 0; rep < a.reps; ++rep) { // reps > 1: timing aid (pantr_debug_set_eval_reps), same result
            pb.eval_psi_xhat(u, 0);
            pb.template reverse<false>(u, pb.mvec(PB::MV_Y), o);
        }
    } else {
        pb.eval_psi_xhat(u, 0);
        pb.eval_grad_xhat(u, pb.vec(3), 0); // records the tape at u
        for (int rep = 0; rep < a.reps; ++rep)
            pb.hess_prod(u, v, o);
    }
    wave_sync();
    for (int i = lane; i < a.n; i += WAVE)
        a.out_v[(size_t)inst * a.n + i] = o[i];
    if (lane == 0)
        a.out_s[inst] = psi;
}

/// Plant step x+ = f_d(x, u) for a batch of scalar-model instances (formulation.hpp:258-261).
template <class M>
__global__ void pantr_scalar_simulate_kernel(int B, const double *x, const double *u, int u_stride, double *xn) {
    const int inst = blockIdx.x * blockDim.x + threadIdx.x;
    if (inst >= B)
        return;
    ScalarALProblem<M> pb;
    double s[M::nx], uk[M::nu], r[M::nx];
    for (int l = 0; l < M::nx; ++l)
        s[l] = x[(size_t)inst * M::nx + l];
    for (int c = 0; c < M::nu; ++c)
        uk[c] = u[(size_t)inst * u_stride + c];
    pb.rk4(M::trig(s), s, uk, r); // one instance per lane: no wave-uniform shortcuts
    for (int l = 0; l < M::nx; ++l)
        xn[(size_t)inst * M::nx + l] = r[l];
}

/// Warm-start shift for u (by nu) and y (by nc), or cold reset (mpc-driver.cpp:168-175,192-197).
__global__ void pantr_shift_uy_kernel(int B, int n, int nu, int m, int nc, int warm, const double *u_init,
                                      double *u, double *y) {
    const int inst = blockIdx.x;
    if (inst >= B)
        return;
    double *ui = u + (size_t)inst * n, *yi = y + (size_t)inst * m;
    auto shift = [&](double *a, int len, int by, const double *init) {
        for (int base = 0; base < len; base += blockDim.x) {
            int i    = base + threadIdx.x;
            double v = 0.0;
            if (i < len)
                v = warm ? (i + by < len ? a[i + by] : a[i]) : (init ? init[i] : 0.0);
            __syncthreads();
            if (i < len)
                a[i] = v;
            __syncthreads();
        }
    };
    shift(ui, n, nu, u_init);
    shift(yi, m, nc, nullptr);
}

} // namespace pantr
