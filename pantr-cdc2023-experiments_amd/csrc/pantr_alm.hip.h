// Problems with general constraints on gfx950: single shooting with an augmented-Lagrangian term,
// scalar stage models (quadcopter family), and the ALM outer loop around pantr_inner().
//
// What it replaces:
//   psi, psi_grad_psi, hess_psi_prod with zeta = g + y/Sigma, zhat = clamp(zeta, D),
//   d = zeta - zhat, yhat = Sigma d, psi = f + 1/2 yhat'd      problems/problems/formulation.hpp:79-99
//   constraints on x_0..x_{N-1} and x_N                          python/.../formulations/ss.py:39-40
//   alpaqa::ALMSolver::operator() (m > 0)                        drivers/mpc-driver.cpp:104-115,217
//   ALM options                                                  new-benchmarks-paper/dodo.py:12-22,101-105
// One wavefront per MPC instance as for the chain; the nine-state stage is evaluated identically by
// every lane (no cross-lane traffic), trajectories and multipliers live in LDS.
#pragma once
#include "pantr_kernel.hip.h"

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
    double *lds; ///< n-vectors: NUM_VEC slots of npad doubles
    int N, n, npad, nj, m, mpad, mj;
    double x0[nx];
    double *mv;  ///< m-vectors: y, Sigma, e1, e2, yh0, yh1, yh2, yh_t (tangent of yhat), mpad doubles each
    double *X;   ///< LDS: primal trajectory (N+1) x nx
    double *Xt;  ///< LDS: tangent trajectory (N+1) x nx
    enum { MV_Y = 0, MV_SIG = 1, MV_E1 = 2, MV_E2 = 3, MV_YH = 4, MV_YHT = 7, MV_COUNT = 8 };

    __device__ __forceinline__ double *vec(int slot) const { return lds + slot * npad; }
    __device__ __forceinline__ double *mvec(int slot) const { return mv + slot * mpad; }
    __host__ __device__ static size_t lds_doubles(int N, int npad, int mpad) {
        return (size_t)NUM_VEC * npad + (size_t)MV_COUNT * mpad + (size_t)2 * (N + 1) * nx;
    }

    // typed views of the inputs / trajectory for T = double (primal sweep) and T = dual (Hv sweep)
    __device__ __forceinline__ void load_u(const double *u, const double *, int k, double (&uk)[nu]) const {
#pragma unroll
        for (int c = 0; c < nu; ++c)
            uk[c] = u[nu * k + c];
    }
    __device__ __forceinline__ void load_u(const double *u, const double *v, int k, dual (&uk)[nu]) const {
#pragma unroll
        for (int c = 0; c < nu; ++c)
            uk[c] = {u[nu * k + c], v[nu * k + c]};
    }
    __device__ __forceinline__ void store_x(int k, const double (&s)[nx]) const {
        if (lane == 0)
#pragma unroll
            for (int l = 0; l < nx; ++l)
                X[k * nx + l] = s[l];
    }
    __device__ __forceinline__ void store_x(int k, const dual (&s)[nx]) const {
        if (lane == 0)
#pragma unroll
            for (int l = 0; l < nx; ++l) {
                X[k * nx + l]  = s[l].v;
                Xt[k * nx + l] = s[l].t;
            }
    }
    __device__ __forceinline__ void load_x(int k, double (&s)[nx]) const {
#pragma unroll
        for (int l = 0; l < nx; ++l)
            s[l] = X[k * nx + l];
    }
    __device__ __forceinline__ void load_x(int k, dual (&s)[nx]) const {
#pragma unroll
        for (int l = 0; l < nx; ++l)
            s[l] = {X[k * nx + l], Xt[k * nx + l]};
    }
    __device__ __forceinline__ void store_yh(double *yh, int j, double v) const {
        if (lane == 0)
            yh[j] = v;
    }
    __device__ __forceinline__ void store_yh(double *yh, int j, dual v) const {
        if (lane == 0) {
            yh[j]            = v.v;
            mvec(MV_YHT)[j]  = v.t;
        }
    }
    __device__ __forceinline__ void load_yh(const double *yh, int k, double (&o)[nc]) const {
#pragma unroll
        for (int i = 0; i < nc; ++i)
            o[i] = yh[nc * k + i];
    }
    __device__ __forceinline__ void load_yh(const double *yh, int k, dual (&o)[nc]) const {
#pragma unroll
        for (int i = 0; i < nc; ++i)
            o[i] = {yh[nc * k + i], mvec(MV_YHT)[nc * k + i]};
    }

    template <class T>
    __device__ __forceinline__ void rk4(const T (&s)[nx], const T (&u)[nu], T (&xn)[nx]) const {
        T k1[nx], k2[nx], k3[nx], k4[nx], w[nx];
        M::fc(s, u, k1);
#pragma unroll
        for (int l = 0; l < nx; ++l)
            w[l] = s[l] + 0.5 * (M::Ts * k1[l]);
        M::fc(w, u, k2);
#pragma unroll
        for (int l = 0; l < nx; ++l)
            w[l] = s[l] + 0.5 * (M::Ts * k2[l]);
        M::fc(w, u, k3);
#pragma unroll
        for (int l = 0; l < nx; ++l)
            w[l] = s[l] + M::Ts * k3[l];
        M::fc(w, u, k4);
#pragma unroll
        for (int l = 0; l < nx; ++l)
            xn[l] = s[l] + M::c6 * (((k1[l] + 2.0 * k2[l]) + 2.0 * k3[l]) + k4[l]);
    }
    template <class T>
    __device__ __forceinline__ void rk4_adj(const T (&s)[nx], const T (&u)[nu], const T (&mu)[nx], T (&sbar)[nx],
                                            T (&ubar)[nu]) const {
        T k[nx], w2[nx], w3[nx], w4[nx];
        M::fc(s, u, k);
#pragma unroll
        for (int l = 0; l < nx; ++l)
            w2[l] = s[l] + 0.5 * (M::Ts * k[l]);
        M::fc(w2, u, k);
#pragma unroll
        for (int l = 0; l < nx; ++l)
            w3[l] = s[l] + 0.5 * (M::Ts * k[l]);
        M::fc(w3, u, k);
#pragma unroll
        for (int l = 0; l < nx; ++l)
            w4[l] = s[l] + M::Ts * k[l];
        T kb1[nx], kb2[nx], kb3[nx], kb4[nx], a[nx], ua[nu];
#pragma unroll
        for (int l = 0; l < nx; ++l) {
            T sg    = M::c6 * mu[l];
            kb1[l]  = sg;
            kb2[l]  = 2.0 * sg;
            kb3[l]  = 2.0 * sg;
            kb4[l]  = sg;
            sbar[l] = mu[l];
        }
        M::fc_adj(w4, u, kb4, a, ubar);
#pragma unroll
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb3[l]  = kb3[l] + M::Ts * a[l];
        }
        M::fc_adj(w3, u, kb3, a, ua);
#pragma unroll
        for (int c = 0; c < nu; ++c)
            ubar[c] = ubar[c] + ua[c];
#pragma unroll
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb2[l]  = kb2[l] + 0.5 * (M::Ts * a[l]);
        }
        M::fc_adj(w2, u, kb2, a, ua);
#pragma unroll
        for (int c = 0; c < nu; ++c)
            ubar[c] = ubar[c] + ua[c];
#pragma unroll
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb1[l]  = kb1[l] + 0.5 * (M::Ts * a[l]);
        }
        M::fc_adj(s, u, kb1, a, ua);
#pragma unroll
        for (int c = 0; c < nu; ++c)
            ubar[c] = ubar[c] + ua[c];
#pragma unroll
        for (int l = 0; l < nx; ++l)
            sbar[l] = sbar[l] + a[l];
    }

    /// Forward sweep: psi, trajectory, yhat.  T = dual carries the tangent v of u.
    template <class T>
    __device__ __forceinline__ T forward(const double *u, const double *v, double *yh) const {
        wave_sync();
        const double *y = mvec(MV_Y), *Sig = mvec(MV_SIG);
        T s[nx], xn[nx], c[nc], uk[nu];
#pragma unroll
        for (int l = 0; l < nx; ++l)
            s[l] = T{} + x0[l];
        T accf = T{}, accal = T{};
        for (int k = 0; k <= N; ++k) {
            store_x(k, s);
            M::constr(s, c);
#pragma unroll
            for (int i = 0; i < nc; ++i) {
                int j  = nc * k + i;
                T zeta = c[i] + (y[j] / Sig[j]);
                T zhat = tmax(tmin(zeta, M::d_ub(i)), M::d_lb(i));
                T d    = zeta - zhat;
                T yhv  = Sig[j] * d;
                store_yh(yh, j, yhv);
                accal = accal + (yhv * d);
            }
            if (k == N)
                break;
            load_u(u, v, k, uk);
            accf = accf + M::stage_cost(s, uk);
            rk4(s, uk, xn);
#pragma unroll
            for (int l = 0; l < nx; ++l)
                s[l] = xn[l];
        }
        accf = accf + M::term_cost(s);
        wave_sync();
        return accf + (0.5 * accal);
    }
    /// Adjoint sweep over the stored trajectory with multipliers yh; OUT(i, value) receives grad_i.
    template <class T, class OUT>
    __device__ __forceinline__ void reverse(const double *u, const double *v, const double *yh, OUT &&out) const {
        wave_sync();
        T lam[nx], sbar[nx], ubar[nu], gu[nu], s[nx], uk[nu], yk[nc];
#pragma unroll
        for (int l = 0; l < nx; ++l)
            lam[l] = T{};
        load_x(N, s);
        M::cost_grad(s, static_cast<const T *>(nullptr), lam, gu);
        load_yh(yh, N, yk);
        M::constr_adj(s, yk, lam);
        for (int k = N - 1; k >= 0; --k) {
            load_x(k, s);
            load_u(u, v, k, uk);
            rk4_adj(s, uk, lam, sbar, ubar);
#pragma unroll
            for (int l = 0; l < nx; ++l)
                lam[l] = sbar[l];
            M::cost_grad(s, uk, lam, gu);
#pragma unroll
            for (int c = 0; c < nu; ++c)
                out(nu * k + c, ubar[c] + gu[c]);
            load_yh(yh, k, yk);
            M::constr_adj(s, yk, lam);
        }
        wave_sync();
    }

    // ---- the interface pantr_inner() drives ----------------------------------------------------
    __device__ __forceinline__ double eval_psi_grad(const double *u, double *grad) const {
        double *yh = mvec(MV_E2); // scratch (e2 is rewritten at the end of the inner solve)
        double psi = forward<double>(u, nullptr, yh);
        reverse<double>(u, nullptr, yh, [&](int i, double g) {
            if (lane == 0)
                grad[i] = g;
        });
        return psi;
    }
    __device__ __forceinline__ double eval_psi_xhat(const double *u, int yh_slot) const {
        return forward<double>(u, nullptr, mvec(MV_YH + yh_slot));
    }
    /// grad psi at the point eval_psi_xhat() was last called at (trajectory and yhat still stored).
    __device__ __forceinline__ void eval_grad_xhat(const double *u, double *grad, int yh_slot) const {
        reverse<double>(u, nullptr, mvec(MV_YH + yh_slot), [&](int i, double g) {
            if (lane == 0)
                grad[i] = g;
        });
    }
    /// Hessian-vector product of psi at u: forward-over-reverse on dual numbers.
    __device__ __forceinline__ void hess_prod(const double *u, const double *v, double *Hv) const {
        double *yh = mvec(MV_E2); // scratch (e2 is rewritten at the end of the inner solve)
        forward<dual>(u, v, yh);
        reverse<dual>(u, v, yh, [&](int i, dual g) {
            if (lane == 0)
                Hv[i] = g.t;
        });
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
                double pi = fmin(fmax(t, M::u_lb(c) - xi), M::u_ub(c) - xi);
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
                double pi = fmin(fmax(t, M::u_lb(c) - xi), M::u_ub(c) - xi);
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
            int i = j * WAVE + lane;
            if (i < n) {
                int c      = i % nu;
                double xfw = x[i] - gamma * grad[i];
                bool in    = M::u_lb(c) < xfw && xfw < M::u_ub(c);
                mask |= (in ? 1u : 0u) << j;
                cnt += in ? 1 : 0;
            }
        }
#pragma unroll
        for (int mm = 32; mm >= 1; mm >>= 1)
            cnt += __shfl_xor(cnt, mm);
        nJ = uniform(cnt);
        return mask;
    }
};

/// ALM outer loop (m > 0) around pantr_inner(), one wavefront per instance.
template <class M>
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
    pb.lane = lane; pb.lds = lds; pb.N = a.N; pb.n = a.n; pb.npad = a.npad; pb.nj = a.nj;
    pb.m = a.m; pb.mpad = a.mpad; pb.mj = a.mj;
    pb.mv = lds + (size_t)NUM_VEC * a.npad;
    pb.X  = pb.mv + (size_t)PB::MV_COUNT * a.mpad;
    pb.Xt = pb.X + (size_t)(a.N + 1) * M::nx;
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
            double ylb = M::d_lb(ci) == -__builtin_huge_val() ? 0.0 : -A.max_multiplier;
            double yub = M::d_ub(ci) == __builtin_huge_val() ? 0.0 : A.max_multiplier;
            y[j]       = fmin(fmax(y[j], ylb), yub);
        }
        wave_sync();
        r = pantr_inner(pb, P, eps, A.max_time_s, t_start, x_io, st,
                        trace && i == 0 ? a.trace + (size_t)inst * a.trace_cap * TRACE_COLS : nullptr, a.trace_cap);
        total_inner += r.iterations;
        bool inner_converged = r.status == Converged;
        failures += inner_converged ? 0 : 1;
        double mx = 0.0;
        for (int j = lane; j < m; j += WAVE)
            mx = fmax(mx, fabs(e2[j]));
        norm_e2 = uniform(wave_max(mx));
        double elapsed     = (double)(wall_clock64() - t_start) * 1e-8;
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
        st.elapsed_s    = (double)(wall_clock64() - t_start) * 1e-8;
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
enum AlmEvalMode : int32_t { ALM_EVAL_PSI_GRAD = 0, ALM_EVAL_HESS_PROD = 1 };

struct AlmEvalArgs {
    int32_t B, N, n, npad, nj, m, mpad, mj, mode, pad0;
    const double *x0;  ///< [B][nx]
    const double *u;   ///< [B][n]
    const double *v;   ///< [B][n] (HESS_PROD)
    const double *y;   ///< [B][m]
    const double *Sig; ///< [B][m]
    double *out_s;     ///< [B] psi
    double *out_v;     ///< [B][n] grad | Hv
    double *out_m;     ///< [B][m] yhat
};

template <class M>
__global__ __launch_bounds__(WAVE) void pantr_alm_eval_kernel(AlmEvalArgs a) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x, inst = blockIdx.x;
    if (inst >= a.B)
        return;
    using PB = ScalarALProblem<M>;
    PB pb;
    pb.lane = lane; pb.lds = lds; pb.N = a.N; pb.n = a.n; pb.npad = a.npad; pb.nj = a.nj;
    pb.m = a.m; pb.mpad = a.mpad; pb.mj = a.mj;
    pb.mv = lds + (size_t)NUM_VEC * a.npad;
    pb.X  = pb.mv + (size_t)PB::MV_COUNT * a.mpad;
    pb.Xt = pb.X + (size_t)(a.N + 1) * M::nx;
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
    wave_sync();
    double psi = 0.0;
    if (a.mode == ALM_EVAL_PSI_GRAD) {
        psi = pb.eval_psi_xhat(u, 0);
        pb.eval_grad_xhat(u, o, 0);
        for (int j = lane; j < a.m; j += WAVE)
            a.out_m[(size_t)inst * a.m + j] = pb.mvec(PB::MV_YH)[j];
    } else {
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
    pb.rk4(s, uk, r);
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
