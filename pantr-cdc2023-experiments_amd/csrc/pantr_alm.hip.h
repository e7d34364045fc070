// Problems with general constraints on gfx950: single shooting with an augmented-Lagrangian term,
// scalar stage models (quadcopter family), and the ALM outer loop around pantr_inner().
//
// What it replaces:
//   psi, psi_grad_psi, hess_psi_prod with zeta = g + y/Sigma, zhat = clamp(zeta, D),
//   d = zeta - zhat, yhat = Sigma d, psi = f + 1/2 yhat'd      problems/problems/formulation.hpp:79-99
//   constraints on x_0..x_{N-1} and x_N                          python/.../formulations/ss.py:39-40
//   alpaqa::ALMSolver::operator() (m > 0)                        drivers/mpc-driver.cpp:104-115,217
//   ALM options                                                  new-benchmarks-paper/dodo.py:12-22,101-105
// One wavefront per MPC instance as for the chain.  The nine-state stage is small, so the first-order
// sweeps evaluate it identically on every lane (no cross-lane traffic; trajectories and multipliers
// live in LDS).  The second-order work is spread over lanes: the adjoint sweep of grad_L runs on dual
// numbers with lane d seeding direction e_d of z = (x_k, u_k), which yields the stage Jacobian J_k
// and the stage Hessian H_k of the Lagrangian-like recursion column by column in ONE sweep; each
// Hessian-vector product of the CG iterations is then 13x13 / 13x9 matrix-vector work per stage,
// lane l owning component l, streamed from the HBM tape one stage ahead of the arithmetic.
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
template <class M>
struct ScalarALProblem {
    static constexpr int nx = M::nx, nu = M::nu, nc = M::nc;
    int lane;
    const BoxData *box; ///< bounds of u (box C) and of the constraints (box D), kernel argument
    double *lds; ///< n-vectors: NUM_VEC slots of npad doubles
    int N, n, npad, nj, m, mpad, mj;
    double x0[nx];
    double *mv;  ///< m-vectors: y, Sigma, e1, e2, yh0, yh1, yh2, y/Sigma, mpad doubles each
    double *X;   ///< LDS: primal trajectory (N+1) x nx
    double *Xt;  ///< LDS: tangent trajectory (N+1) x nx of a product | adjoints lam_k while the tape is recorded
    double2 *TG; ///< LDS: trig store, 12 (sin, cos) pairs per stage (QuadModelT::trig_load)
    double *tape; ///< HBM: this instance's second-order tape (see workspace_doubles)
    enum { MV_Y = 0, MV_SIG = 1, MV_E1 = 2, MV_E2 = 3, MV_YH = 4, MV_YS = 7, MV_COUNT = 8 };

    // Second-order tape, per stage three blocks of 16 rows x 16 doubles: JT row i = (dx+_i/dz_d)_d,
    // JC row d = (dx+_i/dz_d)_i, HC row l = column l of H_k.  After the N stages one block with the
    // terminal Hessian (row l = column l).  Lane r < 16 reads row r; a block stores element e of
    // row r at elem(r, e) = pair-major [e/2][r][e%2], so that one 16-byte load per lane covers 256
    // contiguous bytes across the 16 reading lanes.
    static constexpr int nz = nx + nu, ROW = 16, BLOCK = 16 * ROW, STAGE = 3 * BLOCK;
    __host__ __device__ static size_t workspace_doubles(int N) { return (size_t)N * STAGE + BLOCK; }
    __device__ static __forceinline__ int elem(int r, int e) { return ((e >> 1) * ROW + r) * 2 + (e & 1); }

    static constexpr int CG_BASE = NUM_ITER_VEC; ///< all 20 vectors in LDS: iterates first, then the CG block
    __device__ __forceinline__ double *vec(int slot) const { return lds + slot * npad; }
    __device__ __forceinline__ double *ivec(int slot) const { return lds + slot * npad; }
    __device__ __forceinline__ double *mvec(int slot) const { return mv + slot * mpad; }
    __host__ __device__ static size_t lds_doubles(int N, int npad, int mpad) {
        return (size_t)NUM_VEC * npad + (size_t)MV_COUNT * mpad + (size_t)2 * (N + 1) * nx +
               (size_t)(N + 1) * M::TRIG_DOUBLES_PER_STAGE;
    }

    __device__ __forceinline__ void load_u(const double *u, const double *, int k, double (&uk)[nu]) const {
#pragma unroll
        for (int c = 0; c < nu; ++c)
            uk[c] = u[nu * k + c];
    }
    __device__ __forceinline__ void store_x(int k, const double (&s)[nx]) const {
        if (lane == 0)
#pragma unroll
            for (int l = 0; l < nx; ++l)
                X[k * nx + l] = s[l];
    }
    __device__ __forceinline__ void load_x(int k, double (&s)[nx]) const {
#pragma unroll
        for (int l = 0; l < nx; ++l)
            s[l] = X[k * nx + l];
    }
    __device__ __forceinline__ void store_yh(double *yh, int j, double v) const {
        if (lane == 0)
            yh[j] = v;
    }
    __device__ __forceinline__ void load_yh(const double *yh, int k, double (&o)[nc]) const {
#pragma unroll
        for (int i = 0; i < nc; ++i)
            o[i] = yh[nc * k + i];
    }

    /// One RK4 step.  UNIFORM: the state's values are the same on every lane (Trig by trig_uniform);
    /// t1 = Trig of s, which the caller usually needs as well.  tg (UNIFORM only, may be null): the
    /// stage's trig store; the values of substages 1..3 are written to tg[3], tg[6], tg[9] ...
    template <bool UNIFORM, class T>
    __device__ __forceinline__ void rk4(const typename M::template Trig<T> &t1, const T (&s)[nx], const T (&u)[nu],
                                        T (&xn)[nx], double2 *tg = nullptr) const {
        T k1[nx], k2[nx], k3[nx], k4[nx], w[nx];
        auto trig_sub = [&](int j) {
            if constexpr (UNIFORM)
                return M::trig_uniform(w, tg ? tg + 3 * j : nullptr);
            else
                return M::trig(w);
        };
        M::fc(t1, s, u, k1);
#pragma unroll
        for (int l = 0; l < nx; ++l)
            w[l] = madd(M::hTs, k1[l], s[l]);
        M::fc(trig_sub(1), w, u, k2);
#pragma unroll
        for (int l = 0; l < nx; ++l)
            w[l] = madd(M::hTs, k2[l], s[l]);
        M::fc(trig_sub(2), w, u, k3);
#pragma unroll
        for (int l = 0; l < nx; ++l)
            w[l] = madd(M::Ts, k3[l], s[l]);
        M::fc(trig_sub(3), w, u, k4);
#pragma unroll
        for (int l = 0; l < nx; ++l)
            xn[l] = madd(M::c6, madd(2.0, k3[l], madd(2.0, k2[l], k1[l])) + k4[l], s[l]);
    }
    /// Adjoint of one RK4 step over the stored trigonometric values of its four substage states
    /// (tg: the stage's entries of the trig store, written by the forward sweep at the same point):
    /// fc_adj needs only those, the input and the incoming adjoints, so nothing of the forward pass
    /// is evaluated again.  Values are the same on every lane.
    __device__ __forceinline__ void rk4_adj_stored(const double2 *tg, const double (&u)[nu], const double (&mu)[nx],
                                                   double (&sbar)[nx], double (&ubar)[nu]) const {
        double kb1[nx], kb2[nx], kb3[nx], kb4[nx], a[nx], ua[nu];
#pragma unroll
        for (int l = 0; l < nx; ++l) {
            double sg = M::c6 * mu[l];
            kb1[l]    = sg;
            kb2[l]    = 2.0 * sg;
            kb3[l]    = 2.0 * sg;
            kb4[l]    = sg;
            sbar[l]   = mu[l];
        }
        M::fc_adj(M::trig_load(tg + 9), u, kb4, a, ubar);
#pragma unroll
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb3[l]  = madd(M::Ts, a[l], kb3[l]);
        }
        M::fc_adj(M::trig_load(tg + 6), u, kb3, a, ua);
#pragma unroll
        for (int c = 0; c < nu; ++c)
            ubar[c] = ubar[c] + ua[c];
#pragma unroll
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb2[l]  = madd(M::hTs, a[l], kb2[l]);
        }
        M::fc_adj(M::trig_load(tg + 3), u, kb2, a, ua);
#pragma unroll
        for (int c = 0; c < nu; ++c)
            ubar[c] = ubar[c] + ua[c];
#pragma unroll
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb1[l]  = madd(M::hTs, a[l], kb1[l]);
        }
        M::fc_adj(M::trig_load(tg), u, kb1, a, ua);
#pragma unroll
        for (int c = 0; c < nu; ++c)
            ubar[c] = ubar[c] + ua[c];
#pragma unroll
        for (int l = 0; l < nx; ++l)
            sbar[l] = sbar[l] + a[l];
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

    /// Forward sweep: psi, trajectory, yhat, and the trig store of every substage state.
    __device__ __forceinline__ double forward(const double *u, double *yh) const {
        wave_sync();
        const double *yS = mvec(MV_YS), *Sig = mvec(MV_SIG);
        double s[nx], xn[nx], c[nc], uk[nu], yhv[nc];
#pragma unroll
        for (int l = 0; l < nx; ++l)
            s[l] = x0[l];
        double accf = 0.0, accal = 0.0;
        for (int k = 0; k <= N; ++k) {
            double2 *tg  = TG + 12 * k;
            const auto t1 = M::trig_uniform(s, tg);
            M::constr(t1, s, c);
#pragma unroll
            for (int i = 0; i < nc; ++i) {
                int j       = nc * k + i;
                double zeta = c[i] + yS[j]; // y / Sigma: set_al()
                double zhat = tmax(tmin(zeta, box->d_ub[i]), box->d_lb[i]);
                double d    = zeta - zhat;
                yhv[i]      = Sig[j] * d;
                accal       = accal + (yhv[i] * d);
            }
            store_x(k, s); // one lane-0 block per stage: state row and multiplier estimates
#pragma unroll
            for (int i = 0; i < nc; ++i)
                store_yh(yh, nc * k + i, yhv[i]);
            if (k == N)
                break;
            load_u(u, nullptr, k, uk);
            accf = accf + M::stage_cost(s, uk);
            rk4<true>(t1, s, uk, xn, tg);
#pragma unroll
            for (int l = 0; l < nx; ++l)
                s[l] = xn[l];
        }
        accf = accf + M::term_cost(s);
        wave_sync();
        return accf + (0.5 * accal);
    }
    /// Adjoint sweep over the stored trajectory and trig store with multipliers yh; OUT(i, value)
    /// receives grad_i.  STORE_LAM: the adjoints lam_{k+1} every stage receives are kept (rows of LAM,
    /// which shares the tangent trajectory's LDS) for the second-order recording.
    template <bool STORE_LAM, class OUT>
    __device__ __forceinline__ void reverse(const double *u, const double *yh, OUT &&out) const {
        wave_sync();
        double lam[nx], sbar[nx], ubar[nu], gu[nu], s[nx], uk[nu], yk[nc];
#pragma unroll
        for (int l = 0; l < nx; ++l)
            lam[l] = 0.0;
        load_x(N, s);
        M::cost_grad(s, static_cast<const double *>(nullptr), lam, gu);
        load_yh(yh, N, yk);
        M::constr_adj(M::trig_load(TG + 12 * N), s, yk, lam);
        for (int k = N - 1; k >= 0; --k) {
            if (STORE_LAM && lane == 0)
#pragma unroll
                for (int l = 0; l < nx; ++l)
                    Xt[(k + 1) * nx + l] = lam[l];
            load_x(k, s);
            load_u(u, nullptr, k, uk);
            const double2 *tg = TG + 12 * k;
            rk4_adj_stored(tg, uk, lam, sbar, ubar);
#pragma unroll
            for (int l = 0; l < nx; ++l)
                lam[l] = sbar[l];
            M::cost_grad(s, uk, lam, gu);
#pragma unroll
            for (int c = 0; c < nu; ++c)
                out(nu * k + c, ubar[c] + gu[c]);
            load_yh(yh, k, yk);
            M::constr_adj(M::trig_load(tg), s, yk, lam);
        }
        wave_sync();
    }

    // ---- the interface pantr_inner() drives ----------------------------------------------------
    __device__ __forceinline__ double eval_psi_grad(const double *u, double *grad) const {
        double *yh = mvec(MV_E2); // scratch (e2 is rewritten at the end of the inner solve)
        double psi = forward(u, yh);
        reverse<false>(u, yh, [&](int i, double g) {
            if (lane == 0)
                grad[i] = g;
        });
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
        reverse<true>(u, mvec(MV_YH + yh_slot), [&](int i, double g) {
            if (lane == 0)
                grad[i] = g;
        });
        // terminal block: lane d < nx seeds direction e_d of x_N
        if (lane < nx) {
            const int d = lane;
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
            double *HN = tape + (size_t)N * STAGE;
#pragma unroll
            for (int l = 0; l < nx; ++l)
                HN[elem(d, l)] = lt[l].t;
        }
        const int tasks = N * nz;
        for (int base = 0; base < tasks; base += WAVE) {
            const int task  = base + lane;
            const bool live = task < tasks;
            const int tt    = live ? task : tasks - 1; // idle lanes of the last round repeat a task, unstored
            const int k = tt / nz, d = tt - k * nz;
            dual s[nx], uk[nu], xn[nx], yk[nc], mu[nx], sbar[nx], ubar[nu], gu[nu];
#pragma unroll
            for (int l = 0; l < nx; ++l) {
                s[l]  = {X[k * nx + l], d == l ? 1.0 : 0.0};
                mu[l] = {Xt[(k + 1) * nx + l], 0.0}; // lam_{k+1} (reverse<true>)
            }
#pragma unroll
            for (int c = 0; c < nu; ++c)
                uk[c] = {u[nu * k + c], d == nx + c ? 1.0 : 0.0};
            const double2 *tg = TG + 12 * k;
            double *stage     = tape + (size_t)k * STAGE;
            const auto t1     = M::trig_load(tg, s[6].t, s[7].t, s[8].t);
            stage_multipliers(t1, s, k, yk);
            rk4_adj_dual(tg, t1, s, uk, mu, sbar, ubar, xn);
            if (live) {
#pragma unroll
                for (int i = 0; i < nx; ++i) {
                    stage[elem(i, d)]         = xn[i].t; // JT
                    stage[BLOCK + elem(d, i)] = xn[i].t; // JC
                }
            }
            M::cost_grad(s, uk, sbar, gu);
            double *HC = stage + 2 * BLOCK;
#pragma unroll
            for (int c = 0; c < nu; ++c) {
                dual g = ubar[c] + gu[c];
                if (live)
                    HC[elem(d, nx + c)] = g.t;
            }
            M::constr_adj(t1, s, yk, sbar);
            if (live)
#pragma unroll
                for (int l = 0; l < nx; ++l)
                    HC[elem(d, l)] = sbar[l].t;
        }
        wave_sync();
        wait_vmem(); // the products read the tape back through global loads
    }

    /// The first CNT elements of this lane's row (lane & 15) of a tape block.  Lanes 16..63 own
    /// nothing and fetch the same bytes as lanes 0..15 (same cache lines; masking them off would
    /// put a branch around every group of loads and split the stage bodies into separate
    /// scheduling regions).
    template <int CNT>
    __device__ __forceinline__ void load_row(const double *block, double (&o)[CNT]) const {
        const double2 *p = reinterpret_cast<const double2 *>(block) + (lane & (ROW - 1));
#pragma unroll
        for (int i = 0; i < (CNT + 1) / 2; ++i) {
            double2 t = p[i * ROW];
            o[2 * i]  = t.x;
            if (2 * i + 1 < CNT)
                o[2 * i + 1] = t.y;
        }
    }
    /// Sum of products in four partial sums (term t goes to partial t mod 4, each an fma chain from
    /// 0), combined as (p0 + p1) + (p2 + p3): four independent chains keep a lone wavefront's fp64
    /// pipe busy (a dependent v_fma_f64 costs ~12 cycles, independent ones ~5).
    struct Dot4 {
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
        template <int T>
        __device__ __forceinline__ double &part() {
            return (T & 3) == 0 ? p0 : (T & 3) == 1 ? p1 : (T & 3) == 2 ? p2 : p3;
        }
        /// term T: a * b
        template <int T>
        __device__ __forceinline__ void add(double a, double b) {
            part<T>() = __builtin_fma(a, b, part<T>());
        }
        /// term T: coef * (the value lane SRC of this 16-lane row holds in `own`), the broadcast
        /// fused into the fma (v_fmac_f64 with a row_newbcast DPP source; the only DPP form the
        /// fp64 ALU has).  FIRST: `own` was written by the instruction just before (DPP reads need
        /// two wait states after a VALU write; inline asm is outside the compiler's hazard pass).
        template <int T, int SRC, bool FIRST = false>
        __device__ __forceinline__ void add_bcast(double coef, double own) {
            if (FIRST)
                asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                             : "+v"(part<T>())
                             : "v"(own), "v"(coef), "n"(SRC));
            else
                asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                             : "+v"(part<T>())
                             : "v"(own), "v"(coef), "n"(SRC));
        }
        __device__ __forceinline__ double sum() const { return (p0 + p1) + (p2 + p3); }
    };
    /// acc.add_bcast<T0 + i, i>(coef[i], own) for i = 0..nx-1
    template <int T0, int I = 0>
    __device__ static __forceinline__ void add_bcast_all(Dot4 &acc, const double *coef, double own) {
        if constexpr (I < nx) {
            acc.template add_bcast<T0 + I, I, I == 0>(coef[I], own);
            add_bcast_all<T0, I + 1>(acc, coef, own);
        }
    }
    template <int T0, int CNT, int I = 0>
    __device__ static __forceinline__ void add_all(Dot4 &acc, const double *a, const double *b) {
        if constexpr (I < CNT) {
            acc.template add<T0 + I>(a[I], b[I]);
            add_all<T0, CNT, I + 1>(acc, a, b);
        }
    }
    /// Hessian-vector product over the tape recorded by eval_grad_xhat(): x_t' = J z forward, then
    /// [lam_t ; Hv_k] = H z + J' lam_t' backward; lane l (of the first 16-lane row) owns component l,
    /// every component is one Dot4 over its terms in index order.  The quantity each stage hands to
    /// the next (x_t, lam_t: one component per lane) enters the next stage's fmas as a DPP row
    /// broadcast; everything else a stage needs does not depend on the previous stage and is
    /// loaded ahead: v_k and the stored x_t from LDS one stage ahead (two alternating register
    /// sets), the tape rows from HBM/L2 DF (forward) or DB (backward) stages ahead: that many
    /// register sets, the stage loops unrolled by the depth so that no set is ever copied (a copy
    /// would make the wavefront wait for the youngest load every stage).  A stage is ~100 cycles of
    /// arithmetic and a tape row takes ~1000 cycles to arrive, so the depth is what bounds a
    /// product; the forward rows are the smaller ones (13 doubles against 22) and get the deeper
    /// pipeline.
    static constexpr int DF = 8, DB = 4;
    __device__ __forceinline__ const double *stage_block(int k, int block) const {
        k = k < 0 ? 0 : (k >= N ? N - 1 : k); // clamped: the prefetch runs past the ends
        return tape + (size_t)k * STAGE + block * BLOCK;
    }
    __device__ __forceinline__ void hess_prod(const double *, const double *v, double *Hv) const {
        wave_sync();
        static_assert(DF % 2 == 0 && DB % 2 == 0, "the LDS register sets alternate inside a group");
        // ---- forward: lane i < nx owns x_t[i] ----
        {
            double xt = 0.0;
            auto load_v = [&](int k, double (&vk)[nu]) {
                k = k >= N ? N - 1 : k;
#pragma unroll
                for (int c = 0; c < nu; ++c)
                    vk[c] = v[nu * k + c];
            };
            auto stage = [&](int k, const double (&Arow)[nz], const double (&vk)[nu]) {
                if (lane < nx)
                    Xt[k * nx + lane] = xt; // for the backward sweep
                Dot4 acc;
                add_bcast_all<0>(acc, Arow, xt);
                add_all<nx, nu>(acc, Arow + nx, vk);
                xt = acc.sum();
            };
            double A[DF][nz], va[nu], vb[nu];
#pragma unroll
            for (int j = 0; j < DF; ++j)
                load_row(stage_block(j, 0), A[j]);
            load_v(0, va);
            auto group = [&](int k0, auto guarded) { // full groups run branch-free, the last one guarded
#pragma unroll
                for (int j = 0; j < DF; ++j) {
                    const int k = k0 + j;
                    if (!decltype(guarded)::value || k < N) {
                        if (j % 2 == 0) {
                            load_v(k + 1, vb);
                            stage(k, A[j], va);
                        } else {
                            load_v(k + 1, va);
                            stage(k, A[j], vb);
                        }
                    }
                    load_row(stage_block(k + DF, 0), A[j]);
                }
            };
            int k0 = 0;
            for (; k0 + DF <= N; k0 += DF)
                group(k0, std::false_type{});
            if (k0 < N)
                group(k0, std::true_type{});
            if (lane < nx)
                Xt[N * nx + lane] = xt;
            wait_vmem(); // the clamped prefetches past the last stage
            wave_sync();
        }
        // ---- backward: lane l < nz owns component l of [lam_t ; Hv_k] ----
        // z = (x_t of the stage, v_k) does not depend on the previous stage
        double lt;
        auto load_z = [&](int k, double (&z)[nz]) {
            k = k < 0 ? 0 : k;
#pragma unroll
            for (int d = 0; d < nx; ++d)
                z[d] = Xt[k * nx + d];
#pragma unroll
            for (int c = 0; c < nu; ++c)
                z[nx + c] = v[nu * k + c];
        };
        auto stage = [&](int k, const double (&Crow)[nz], const double (&Brow)[nx], const double (&z)[nz]) {
            Dot4 acc;
            add_all<0, nz>(acc, Crow, z);
            add_bcast_all<nz>(acc, Brow, lt);
            double w = acc.sum();
            if (lane >= nx && lane < nz)
                Hv[nu * k + (lane - nx)] = w;
            lt = w;
        };
        double C[DB][nz], Bc[DB][nx];
#pragma unroll
        for (int j = 0; j < DB; ++j) {
            load_row(stage_block(N - 1 - j, 2), C[j]);
            load_row(stage_block(N - 1 - j, 1), Bc[j]);
        }
        {
            double HN[nx];
            load_row(tape + (size_t)N * STAGE, HN);
            Dot4 acc;
            add_all<0, nx>(acc, HN, Xt + N * nx);
            lt = acc.sum();
        }
        double z0[nz], z1[nz];
        load_z(N - 1, z0);
        auto group = [&](int k0, auto guarded) { // stages k0, k0 - 1, ... k0 - DB + 1
#pragma unroll
            for (int j = 0; j < DB; ++j) {
                const int k = k0 - j;
                if (!decltype(guarded)::value || k >= 0) {
                    if (j % 2 == 0) {
                        load_z(k - 1, z1);
                        stage(k, C[j], Bc[j], z0);
                    } else {
                        load_z(k - 1, z0);
                        stage(k, C[j], Bc[j], z1);
                    }
                }
                load_row(stage_block(k - DB, 2), C[j]);
                load_row(stage_block(k - DB, 1), Bc[j]);
            }
        };
        int k0 = N - 1;
        for (; k0 - DB + 1 >= 0; k0 -= DB)
            group(k0, std::false_type{});
        if (k0 >= 0)
            group(k0, std::true_type{});
        wait_vmem(); // the clamped prefetches past the first stage
        wave_sync();
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
    __device__ __forceinline__ double dot(const double *a, const double *b) const {
        double acc = 0.0;
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + lane;
            if (i < n)
                acc = acc + a[i] * b[i];
        }
        return uniform(wave_sum(acc));
    }
    __device__ __forceinline__ double dot_masked(const double *a, const double *b, uint32_t mask) const {
        double acc = 0.0;
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + lane;
            if (i < n)
                acc = acc + a[i] * ((mask >> j) & 1u ? b[i] : 0.0);
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
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + lane;
            if (i < n) {
                int c     = i % nu;
                double xi = x[i], gi = grad[i];
                double t  = (-gamma) * gi;
                double pi = fmin(fmax(t, box_pick(box->u_lb, c) - xi), box_pick(box->u_ub, c) - xi);
                p[i]      = pi;
                xh[i]     = xi + pi;
                a1        = a1 + pi * pi;
                a2        = a2 + pi * gi;
            }
        }
        pTp    = uniform(wave_sum(a1));
        gradTp = uniform(wave_sum(a2));
    }
    __device__ __forceinline__ double proj_grad_unit_norm(const double *x, const double *grad) const {
        double r = 0.0;
        for (int j = 0; j < nj; ++j) {
            int i = j * WAVE + lane;
            if (i < n) {
                int c     = i % nu;
                double xi = x[i];
                double t  = (-1.0) * grad[i];
                double pi = fmin(fmax(t, box_pick(box->u_lb, c) - xi), box_pick(box->u_ub, c) - xi);
                r         = fmax(r, fabs(pi));
            }
        }
        return uniform(wave_max(r));
    }
    __device__ __forceinline__ uint32_t inactive_mask(double gamma, const double *x, const double *grad,
                                                      int &nJ) const {
        uint32_t mask = 0;
        int cnt       = 0;
        for (int j = 0; j < nj; ++j) {
            int i   = j * WAVE + lane;
            bool in = false;
            if (i < n) {
                int c      = i % nu;
                double xfw = x[i] - gamma * grad[i];
                in = box_pick(box->u_lb, c) < xfw && xfw < box_pick(box->u_ub, c);
            }
            mask |= (in ? 1u : 0u) << j;
            cnt += __builtin_popcountll(__ballot(in ? 1 : 0)); // wave-uniform count, no reduction needed
        }
        nJ = cnt;
        return mask;
    }
};

/// ALM outer loop (m > 0) around the inner solver METHOD names, one wavefront per instance.
template <class M, int METHOD>
__global__ __launch_bounds__(WAVE) void pantr_alm_kernel(AlmSolveArgs a) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int inst = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    if (inst >= a.B)
        return;
    const long long t_start = wall_clock64();
    const KernelParams &P    = a.prm;
    const AlmKernelParams &A = a.alm;
    using PB = ScalarALProblem<M>;
    PB pb;
    pb.lane = lane; pb.box = &a.prm.box; pb.lds = lds; pb.N = a.N; pb.n = a.n; pb.npad = a.npad; pb.nj = a.nj;
    pb.m = a.m; pb.mpad = a.mpad; pb.mj = a.mj;
    pb.mv = lds + (size_t)NUM_VEC * a.npad;
    pb.X  = pb.mv + (size_t)PB::MV_COUNT * a.mpad;
    pb.Xt = pb.X + (size_t)(a.N + 1) * M::nx;
    pb.TG = reinterpret_cast<double2 *>(pb.Xt + (size_t)(a.N + 1) * M::nx);
    pb.tape = a.ws + (size_t)inst * PB::workspace_doubles(a.N);
#pragma unroll
    for (int l = 0; l < M::nx; ++l)
        pb.x0[l] = a.x0[(size_t)inst * M::nx + l];
    const int n = a.n, m = a.m;
    const size_t total = PB::lds_doubles(a.N, a.npad, a.mpad);
    for (size_t i = lane; i < total; i += WAVE)
        lds[i] = 0.0;
    wave_sync();
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
        a.stats[inst]                 = st;
    }
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
    extern __shared__ double lds[];
    const int lane = threadIdx.x, inst = blockIdx.x;
    if (inst >= a.B)
        return;
    using PB = ScalarALProblem<M>;
    PB pb;
    pb.lane = lane; pb.box = &a.box; pb.lds = lds; pb.N = a.N; pb.n = a.n; pb.npad = a.npad; pb.nj = a.nj;
    pb.m = a.m; pb.mpad = a.mpad; pb.mj = a.mj;
    pb.mv = lds + (size_t)NUM_VEC * a.npad;
    pb.X  = pb.mv + (size_t)PB::MV_COUNT * a.mpad;
    pb.Xt = pb.X + (size_t)(a.N + 1) * M::nx;
    pb.TG = reinterpret_cast<double2 *>(pb.Xt + (size_t)(a.N + 1) * M::nx);
    pb.tape = a.ws + (size_t)inst * PB::workspace_doubles(a.N);
#pragma unroll
    for (int l = 0; l < M::nx; ++l)
        pb.x0[l] = a.x0[(size_t)inst * M::nx + l];
    const size_t total = PB::lds_doubles(a.N, a.npad, a.mpad);
    for (size_t i = lane; i < total; i += WAVE)
        lds[i] = 0.0;
    wave_sync();
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
        psi = pb.eval_psi_xhat(u, 0);
        pb.eval_grad_xhat(u, o, 0);
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
        pb.eval_psi_xhat(u, 0);
        pb.template reverse<false>(u, pb.mvec(PB::MV_Y), [&](int i, double g) {
            if (lane == 0)
                o[i] = g;
        });
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
    pb.template rk4<false>(M::trig(s), s, uk, r); // one instance per lane: no wave-uniform shortcuts
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
