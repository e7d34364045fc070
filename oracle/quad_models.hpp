// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/README.md).
//
// Plain scalar restatement of the quadcopter and realistic_quadcopter optimal-control problems
// and of the single-shooting augmented-Lagrangian problem functions for problems with general
// constraints.  `method=pantr` loads the CasADi problem generated from the Python models
// (problems/generate_problem.py:26-61), so those are the text followed:
//   quadcopter ............ python/alpaqa_mpc_benchmarks/problems/quadcopter.py:6-112
//                           (C++ twin problems/problems/quadcopter.cpp:69-160)
//   realistic_quadcopter .. python/alpaqa_mpc_benchmarks/problems/realistic_quadcopter.py:6-118
//                           (C++ twin problems/problems/realistic_quadcopter.cpp:69-162)
//   single shooting ....... python/alpaqa_mpc_benchmarks/formulations/ss.py:20-55
//                           (constraints on x_0..x_{N-1} and x_N: ss.py:39-40)
//   psi, grad, Hv ......... problems/problems/formulation.hpp:79-99
//                           zeta = g + y/Sigma, zhat = clamp(zeta, D), d = zeta - zhat,
//                           yhat = Sigma d, psi = f + 1/2 yhat'd
// sin/cos: one short deterministic routine (2-term Cody-Waite reduction + the fdlibm kernel
// polynomials, explicit fma) used identically by the kernels, so trigonometric values agree
// bit for bit between CPU and GPU (libm and the device math library do not).
#pragma once
#include "chain_model.hpp"

namespace oracle {

// ---------------------------------------------------------------------------
// deterministic sin/cos
// ---------------------------------------------------------------------------
inline void sincos_det(double x, double &s, double &c) {
    const double kd = std::nearbyint(x * 6.36619772367581382433e-01); // x * 2/pi, ties to even
    double r        = std::fma(-kd, 1.57079632673412561417e+00, x);  // pi/2, leading bits
    r               = std::fma(-kd, 6.07710050650619224932e-11, r);  // pi/2, next bits
    const double z  = r * r;
    double ps = std::fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps        = std::fma(z, ps, 2.75573137070700676789e-06);
    ps        = std::fma(z, ps, -1.98412698298579493134e-04);
    ps        = std::fma(z, ps, 8.33333333332248946124e-03);
    ps        = std::fma(z, ps, -1.66666666666666324348e-01);
    const double sr = std::fma(r * z, ps, r);
    double pc = std::fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc        = std::fma(z, pc, -2.75573143513906633035e-07);
    pc        = std::fma(z, pc, 2.48015872894767294178e-05);
    pc        = std::fma(z, pc, -1.38888888888741095749e-03);
    pc        = std::fma(z, pc, 4.16666666666666019037e-02);
    const double cr = std::fma(z * z, pc, std::fma(-0.5, z, 1.0));
    const int q = static_cast<int>(kd) & 3; // |x| < 3e9 (64-bit conversions are emulated on the GPU)
    s = q == 0 ? sr : q == 1 ? cr : q == 2 ? -sr : -cr;
    c = q == 0 ? cr : q == 1 ? -sr : q == 2 ? -cr : sr;
}
inline void sincos_t(double x, double &s, double &c) { sincos_det(x, s, c); }
inline void sincos_t(Dual x, Dual &s, Dual &c) {
    double sv, cv;
    sincos_det(x.v, sv, cv);
    s = {sv, cv * x.t};
    c = {cv, -(sv * x.t)};
}
inline Dual operator-(Dual a) { return {-a.v, -a.t}; }
/// c k + s in one rounding (the Runge-Kutta combinations of the scalar models)
inline double madd(double c, double k, double s) { return std::fma(c, k, s); }
inline Dual madd(double c, Dual k, Dual s) { return {std::fma(c, k.v, s.v), std::fma(c, k.t, s.t)}; }
/// clamp pieces with the tangent of the selected branch
inline double tmin(double a, double b) { return a < b ? a : b; }
inline double tmax(double a, double b) { return a > b ? a : b; }
inline Dual tmin(Dual a, double b) { return a.v < b ? a : Dual{b, 0.0}; }
inline Dual tmax(Dual a, double b) { return a.v > b ? a : Dual{b, 0.0}; }

// ---------------------------------------------------------------------------
// quadcopter models: state (p[3], v[3], theta[3]), input (a_t, omega[3])
// ---------------------------------------------------------------------------
template <bool REALISTIC>
struct QuadcopterT {
    static constexpr int nx = 9, nu = 4, nc = 4;
    static constexpr bool realistic = REALISTIC;
    static constexpr double Ts = 0.1, hTs = 0.5 * Ts, g_grav = 9.81, c6 = Ts / 6;
    static constexpr double r2 = 0.1 * 0.1; ///< collision cylinder radius squared (quadcopter.py:70)
    static constexpr double pf[3] = {0.25, 0.25, 0.5}; // quadcopter.py:90
    static const char *name() { return REALISTIC ? "realistic_quadcopter" : "quadcopter"; }

    static double u_lb(int c) { return c == 0 ? 0.0 : -0.1; }          // quadcopter.py:55-58
    static double u_ub(int c) { return c == 0 ? 9.81 * 5 : 0.1; }
    static double d_lb(int i) { // quadcopter.py:72-75
        return i < 2 ? -1.5707963267948966 : i == 2 ? 0.8525245220595057 /* cos(1.1/2) */ : -inf_();
    }
    static double d_ub(int i) { return i < 2 ? 1.5707963267948966 : i == 2 ? inf_() : 0.0; }
    static constexpr double inf_() { return std::numeric_limits<double>::infinity(); }
    static void initial_state(double *x0) { // quadcopter.py:80-88
        const double v[9] = {-0.2, -0.25, 0.5, 0, 0, 0, 0, 0, 0};
        std::copy(v, v + 9, x0);
    }
    static double initial_guess(int c) { return c == 0 ? 9.81 : 0.0; } // quadcopter.py:111

    /// Third column of the body-to-world rotation and, for the realistic model, the Euler-rate
    /// map d_orient = Omega^{-1} omega, with the trigonometric values they are built from.
    template <class T>
    struct Trig {
        T s0, c0, s1, c1, s2, c2;
    };
    template <class T>
    static Trig<T> trig(const T *s) {
        Trig<T> t;
        sincos_t(s[6], t.s0, t.c0);
        sincos_t(s[7], t.s1, t.c1);
        sincos_t(s[8], t.s2, t.c2);
        return t;
    }

    /// f_c(state, input) (quadcopter.py:30-46, realistic_quadcopter.py:30-50)
    template <class T>
    static void fc(const T *s, const T *u, T *ds) {
        const Trig<T> t = trig(s);
        T r0, r1, r2_;
        if (!REALISTIC) { // (cr,sr)=(c0,s0) roll, (cp,sp)=(c1,s1) pitch, (cy,sy)=(c2,s2) yaw
            r0  = ((t.c2 * t.s1) * t.c0) + (t.s2 * t.s0);
            r1  = ((t.s2 * t.s1) * t.c0) - (t.c2 * t.s0);
            r2_ = t.c1 * t.c0;
        } else { // phi = s[6], theta = s[7], psi = s[8]
            r0  = (t.c2 * t.s1) + ((t.c1 * t.s0) * t.s2);
            r1  = (t.s2 * t.s1) - ((t.c2 * t.c1) * t.s0);
            r2_ = t.c0 * t.c1;
        }
        ds[0] = s[3];
        ds[1] = s[4];
        ds[2] = s[5];
        ds[3] = r0 * u[0];
        ds[4] = r1 * u[0];
        ds[5] = (r2_ * u[0]) - g_grav;
        if (!REALISTIC) {
            ds[6] = u[1];
            ds[7] = u[2];
            ds[8] = u[3];
        } else { // Omega [a b c]' = omega, Omega = [[c1,0,-c0 s1],[0,1,s0],[s1,0,c0 c1]], det = c0
            T a   = (t.c1 * u[1]) + (t.s1 * u[3]);
            T cc  = ((t.c1 * u[3]) - (t.s1 * u[1])) / t.c0;
            T b   = u[2] - (t.s0 * cc);
            ds[6] = a;
            ds[7] = b;
            ds[8] = cc;
        }
    }

    /// Adjoint of f_c: sb = (df/ds)' kb, ub = (df/du)' kb.
    template <class T>
    static void fc_adj(const T *s, const T *u, const T *kb, T *sb, T *ub) {
        const Trig<T> t = trig(s);
        const T at = u[0];
        T r0, r1, r2_, dr0[3], dr1[3], dr2[3]; // d r_i / d s[6+j]
        if (!REALISTIC) {
            r0     = ((t.c2 * t.s1) * t.c0) + (t.s2 * t.s0);
            r1     = ((t.s2 * t.s1) * t.c0) - (t.c2 * t.s0);
            r2_    = t.c1 * t.c0;
            dr0[0] = (t.s2 * t.c0) - ((t.c2 * t.s1) * t.s0);
            dr0[1] = (t.c2 * t.c1) * t.c0;
            dr0[2] = (t.c2 * t.s0) - ((t.s2 * t.s1) * t.c0);
            dr1[0] = -(((t.s2 * t.s1) * t.s0) + (t.c2 * t.c0));
            dr1[1] = (t.s2 * t.c1) * t.c0;
            dr1[2] = r0;
            dr2[0] = -(t.c1 * t.s0);
            dr2[1] = -(t.s1 * t.c0);
            dr2[2] = T{} * t.c0; // 0
        } else {
            r0     = (t.c2 * t.s1) + ((t.c1 * t.s0) * t.s2);
            r1     = (t.s2 * t.s1) - ((t.c2 * t.c1) * t.s0);
            r2_    = t.c0 * t.c1;
            dr0[0] = (t.c1 * t.c0) * t.s2;
            dr0[1] = (t.c2 * t.c1) - ((t.s1 * t.s0) * t.s2);
            dr0[2] = ((t.c1 * t.s0) * t.c2) - (t.s2 * t.s1);
            dr1[0] = -((t.c2 * t.c1) * t.c0);
            dr1[1] = (t.s2 * t.c1) + ((t.c2 * t.s1) * t.s0);
            dr1[2] = r0;
            dr2[0] = -(t.s0 * t.c1);
            dr2[1] = -(t.c0 * t.s1);
            dr2[2] = T{} * t.c0; // 0
        }
        const T a0 = kb[3], a1 = kb[4], a2 = kb[5];
        sb[0] = T{} * at; // no dependence on p
        sb[1] = sb[0];
        sb[2] = sb[0];
        sb[3] = kb[0];
        sb[4] = kb[1];
        sb[5] = kb[2];
        ub[0] = ((r0 * a0) + (r1 * a1)) + (r2_ * a2);
        const T w0 = at * a0, w1 = at * a1, w2 = at * a2;
        for (int j = 0; j < 3; ++j)
            sb[6 + j] = ((dr0[j] * w0) + (dr1[j] * w1)) + (dr2[j] * w2);
        if (!REALISTIC) {
            ub[1] = kb[6];
            ub[2] = kb[7];
            ub[3] = kb[8];
        } else {
            // a = c1 u1 + s1 u3 ; cc = (c1 u3 - s1 u1)/c0 ; b = u2 - s0 cc
            const T ka = kb[6], kb_ = kb[7], kc = kb[8];
            const T num = (t.c1 * u[3]) - (t.s1 * u[1]);
            const T ic  = 1.0 / t.c0;          // one reciprocal serves the three quotients by cos(phi)
            const T cc  = num * ic;
            const T kcc = kc - (t.s0 * kb_);   // total adjoint of cc
            const T q   = kcc * ic;            // adjoint of num
            ub[1] = (t.c1 * ka) - (t.s1 * q);
            ub[2] = kb_;
            ub[3] = (t.s1 * ka) + (t.c1 * q);
            // d/dphi (s[6]): b -> -c0 cc kb_ ; cc = num/c0 -> num s0/c0^2 = cc s0/c0
            sb[6] = sb[6] + ((((cc * t.s0) * ic) * kcc) - ((t.c0 * cc) * kb_));
            // d/dtheta (s[7]): a -> (-s1 u1 + c1 u3) ka = num ka ; num -> (-s1 u3 - c1 u1) q
            sb[7] = sb[7] + ((num * ka) - (((t.s1 * u[3]) + (t.c1 * u[1])) * q));
        }
    }

    /// l(x,u) (quadcopter.py:93-106): lu + lx
    template <class T>
    static T stage_cost(const T *s, const T *u) {
        T su  = (((u[0] * u[0]) + (u[1] * u[1])) + (u[2] * u[2])) + (u[3] * u[3]);
        T so  = ((u[1] * u[1]) + (u[2] * u[2])) + (u[3] * u[3]);
        T lu  = (1e-4 * su) + (10.0 * so);
        return lu + state_cost(s, 1.0, 10.0, 1.0);
    }
    /// l_N(x) (quadcopter.py:107-112)
    template <class T>
    static T term_cost(const T *s) { return state_cost(s, 25.0, 250.0, 10.0); }
    template <class T>
    static T state_cost(const T *s, double qv, double qp, double qth) {
        T d0 = s[0] - pf[0], d1 = s[1] - pf[1], d2 = s[2] - pf[2];
        T sp = ((d0 * d0) + (d1 * d1)) + (d2 * d2);
        T sv = ((s[3] * s[3]) + (s[4] * s[4])) + (s[5] * s[5]);
        T st = ((s[6] * s[6]) + (s[7] * s[7])) + (s[8] * s[8]);
        return ((qv * sv) + (qp * sp)) + (qth * st);
    }
    /// gs += dl/dx, gu = dl/du  (u == nullptr: terminal cost)
    template <class T>
    static void cost_grad(const T *s, const T *u, T *gs, T *gu) {
        const double qv = u ? 1.0 : 25.0, qp = u ? 10.0 : 250.0, qth = u ? 1.0 : 10.0;
        for (int i = 0; i < 3; ++i) {
            gs[i]     = gs[i] + (2.0 * qp) * (s[i] - pf[i]);
            gs[3 + i] = gs[3 + i] + (2.0 * qv) * s[3 + i];
            gs[6 + i] = gs[6 + i] + (2.0 * qth) * s[6 + i];
        }
        if (u) {
            gu[0] = 2e-4 * u[0];
            for (int j = 1; j < 4; ++j)
                gu[j] = (2e-4 * u[j]) + (20.0 * u[j]);
        }
    }

    /// c(x) (quadcopter.py:62-71)
    template <class T>
    static void constr(const T *s, T *c) {
        T s0, c0, s1, c1;
        sincos_t(s[6], s0, c0);
        sincos_t(s[7], s1, c1);
        c[0] = s[6];
        c[1] = s[7];
        c[2] = c0 * c1;
        c[3] = (r2 - (s[0] * s[0])) - (s[1] * s[1]);
    }
    /// sb += (dc/dx)' yb
    template <class T>
    static void constr_adj(const T *s, const T *yb, T *sb) {
        T s0, c0, s1, c1;
        sincos_t(s[6], s0, c0);
        sincos_t(s[7], s1, c1);
        sb[6] = sb[6] + (yb[0] - ((s0 * c1) * yb[2]));
        sb[7] = sb[7] + (yb[1] - ((c0 * s1) * yb[2]));
        sb[0] = sb[0] - ((2.0 * s[0]) * yb[3]);
        sb[1] = sb[1] - ((2.0 * s[1]) * yb[3]);
    }
};
using Quadcopter          = QuadcopterT<false>;
using RealisticQuadcopter = QuadcopterT<true>;

// ---------------------------------------------------------------------------
// Single shooting with general constraints (augmented Lagrangian), scalar models
// ---------------------------------------------------------------------------
template <class Model>
struct SingleShootingAL {
    int N;
    std::vector<double> x0;
    mutable EvalCounters cnt;
    const double *y = nullptr, *Sig = nullptr; ///< multipliers and penalties of the current ALM iteration
    static constexpr int nx = Model::nx, nu = Model::nu, nc = Model::nc;

    explicit SingleShootingAL(int N_) : N(N_), x0(nx) { Model::initial_state(x0.data()); }
    int n() const { return nu * N; }
    int m() const { return nc * N + nc; } // ss.py:39-40: c on x_0..x_{N-1}, c_N on x_N
    void set_al(const double *y_, const double *Sig_) {
        y           = y_;
        Sig         = Sig_;
        tape_valid_ = false;
    }
    double lb(int i) const { return Model::u_lb(i % nu); }
    double ub(int i) const { return Model::u_ub(i % nu); }
    double d_lb(int i) const { return Model::d_lb(i % nc); } // formulation.hpp:229-238
    double d_ub(int i) const { return Model::d_ub(i % nc); }

    template <class T>
    static void rk4(const T *s, const T *u, T *xn) {
        T k1[nx], k2[nx], k3[nx], k4[nx], w[nx];
        Model::fc(s, u, k1);
        for (int l = 0; l < nx; ++l)
            w[l] = madd(Model::hTs, k1[l], s[l]);
        Model::fc(w, u, k2);
        for (int l = 0; l < nx; ++l)
            w[l] = madd(Model::hTs, k2[l], s[l]);
        Model::fc(w, u, k3);
        for (int l = 0; l < nx; ++l)
            w[l] = madd(Model::Ts, k3[l], s[l]);
        Model::fc(w, u, k4);
        for (int l = 0; l < nx; ++l)
            xn[l] = madd(Model::c6, madd(2.0, k3[l], madd(2.0, k2[l], k1[l])) + k4[l], s[l]);
    }
    template <class T>
    static void rk4_adj(const T *s, const T *u, const T *mu, T *sbar, T *ubar) {
        T k[nx], w2[nx], w3[nx], w4[nx];
        Model::fc(s, u, k);
        for (int l = 0; l < nx; ++l)
            w2[l] = madd(Model::hTs, k[l], s[l]);
        Model::fc(w2, u, k);
        for (int l = 0; l < nx; ++l)
            w3[l] = madd(Model::hTs, k[l], s[l]);
        Model::fc(w3, u, k);
        for (int l = 0; l < nx; ++l)
            w4[l] = madd(Model::Ts, k[l], s[l]);
        T kb1[nx], kb2[nx], kb3[nx], kb4[nx], a[nx], ua[nu];
        for (int l = 0; l < nx; ++l) {
            T sg   = Model::c6 * mu[l];
            kb1[l] = sg;
            kb2[l] = 2.0 * sg;
            kb3[l] = 2.0 * sg;
            kb4[l] = sg;
            sbar[l] = mu[l];
        }
        Model::fc_adj(w4, u, kb4, a, ubar);
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb3[l]  = madd(Model::Ts, a[l], kb3[l]);
        }
        Model::fc_adj(w3, u, kb3, a, ua);
        for (int c = 0; c < nu; ++c)
            ubar[c] = ubar[c] + ua[c];
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb2[l]  = madd(Model::hTs, a[l], kb2[l]);
        }
        Model::fc_adj(w2, u, kb2, a, ua);
        for (int c = 0; c < nu; ++c)
            ubar[c] = ubar[c] + ua[c];
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb1[l]  = madd(Model::hTs, a[l], kb1[l]);
        }
        Model::fc_adj(s, u, kb1, a, ua);
        for (int c = 0; c < nu; ++c)
            ubar[c] = ubar[c] + ua[c];
        for (int l = 0; l < nx; ++l)
            sbar[l] = sbar[l] + a[l];
    }

    /// Forward sweep: psi (and yhat), trajectory.  u_t/X typed T for the dual-number product.
    template <class T>
    T forward(const T *u, T *X /*(N+1)*nx*/, T *yhat /*m or nullptr*/) const {
        T s[nx], xn[nx], c[nc];
        for (int l = 0; l < nx; ++l)
            s[l] = T{} + x0[l];
        T accf = T{}, accal = T{};
        for (int k = 0; k <= N; ++k) {
            for (int l = 0; l < nx; ++l)
                X[k * nx + l] = s[l];
            Model::constr(s, c);
            for (int i = 0; i < nc; ++i) {
                int j     = nc * k + i;
                T zeta    = c[i] + (y[j] / Sig[j]);
                T zhat    = tmax(tmin(zeta, d_ub(j)), d_lb(j));
                T d       = zeta - zhat;
                T yh      = Sig[j] * d;
                if (yhat)
                    yhat[j] = yh;
                accal = accal + (yh * d);
            }
            if (k == N)
                break;
            accf = accf + Model::stage_cost(s, u + nu * k);
            rk4(s, u + nu * k, xn);
            for (int l = 0; l < nx; ++l)
                s[l] = xn[l];
        }
        accf = accf + Model::term_cost(s);
        return accf + (0.5 * accal);
    }
    /// Adjoint sweep over a stored trajectory with multipliers yhat: grad = grad_u (f + yhat' g).
    template <class T>
    void reverse(const T *u, const T *X, const T *yhat, T *grad) const {
        if constexpr (std::is_same_v<T, double> && Model::realistic)
            return reverse_affine(u, X, yhat, grad, nullptr);
        else
            reverse_generic(u, X, yhat, grad);
    }
    /// Stage by stage through the full stage adjoint (what reverse() is for quadcopter, and the check of
    /// reverse_affine for realistic_quadcopter).
    template <class T>
    void reverse_generic(const T *u, const T *X, const T *yhat, T *grad) const {
        T lam[nx], sbar[nx], ubar[nu], gu[nu] = {};
        for (int l = 0; l < nx; ++l)
            lam[l] = T{};
        Model::cost_grad(X + N * nx, static_cast<const T *>(nullptr), lam, gu);
        Model::constr_adj(X + N * nx, yhat + nc * N, lam);
        for (int k = N - 1; k >= 0; --k) {
            const T *s = X + k * nx;
            rk4_adj(s, u + nu * k, lam, sbar, ubar);
            for (int l = 0; l < nx; ++l)
                lam[l] = sbar[l];
            Model::cost_grad(s, u + nu * k, lam, gu);
            for (int c = 0; c < nu; ++c)
                grad[nu * k + c] = ubar[c] + gu[c];
            Model::constr_adj(s, yhat + nc * k, lam);
        }
    }

    /// Adjoint sweep of realistic_quadcopter (grad, and lam_{k+1} of every stage in LAM if given), as the
    /// kernels form it.  The adjoints of position and velocity do not depend on the attitude's, and the
    /// attitude adjoint of a stage is an AFFINE function of the next stage's: lam_a(k) = M_k lam_a(k+1) +
    /// c_k.  c_k and the thrust gradient come from the stage adjoint with a zero attitude seed (which
    /// carries the position / velocity recursion exactly as reverse() does), the columns of M_k and of
    /// d grad_w / d lam_a from the stage adjoint on the three unit attitude seeds; then the yaw adjoint is a
    /// running sum (nothing depends on the yaw), (lam_phi, lam_theta) an affine_scan over the reversed
    /// stages, and the rate gradients fma chains over lam_a(k+1).  Equal to reverse() up to rounding.
    void reverse_affine(const double *u, const double *X, const double *yhat, double *grad, double *LAM) const {
        std::vector<double> lam_all((N + 1) * nx, 0.0), ctot(3 * N), cu(3 * N), cw(3 * N), Mc(9 * N), Uc(9 * N);
        double lam[nx], sbar[nx], ubar[nu], gu[nu] = {0.0, 0.0, 0.0, 0.0};
        for (int l = 0; l < nx; ++l)
            lam[l] = 0.0;
        Model::cost_grad(X + N * nx, static_cast<const double *>(nullptr), lam, gu);
        Model::constr_adj(X + N * nx, yhat + nc * N, lam);
        std::copy(lam, lam + nx, &lam_all[N * nx]);
        for (int k = N - 1; k >= 0; --k) {
            const double *s = X + k * nx, *uk = u + nu * k;
            double mu[nx] = {lam[0], lam[1], lam[2], lam[3], lam[4], lam[5], 0.0, 0.0, 0.0};
            rk4_adj(s, uk, mu, sbar, ubar);
            Model::cost_grad(s, uk, sbar, gu);
            grad[nu * k] = ubar[0] + gu[0];
            Model::constr_adj(s, yhat + nc * k, sbar);
            for (int j = 0; j < 3; ++j) {
                ctot[3 * k + j] = sbar[6 + j];
                cu[3 * k + j]   = ubar[1 + j];
                cw[3 * k + j]   = gu[1 + j];
            }
            for (int l = 0; l < 6; ++l)
                lam[l] = lam_all[k * nx + l] = sbar[l];
            for (int e = 0; e < 3; ++e) {
                double seed[nx] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, sb[nx], ub[nu];
                seed[6 + e] = 1.0;
                rk4_adj(s, uk, seed, sb, ub);
                for (int j = 0; j < 3; ++j) {
                    Mc[9 * k + 3 * e + j] = sb[6 + j];
                    Uc[9 * k + 3 * e + j] = ub[1 + j];
                }
            }
        }
        // yaw: lam_psi(k) = lam_psi(k+1) + c_k[psi]
        std::vector<double> inc(N), run(N + 1);
        for (int j = 0; j < N; ++j)
            inc[j] = ctot[3 * (N - 1 - j) + 2];
        prefix_scan(inc.data(), N, lam_all[N * nx + 8], run.data());
        for (int j = 0; j <= N; ++j)
            lam_all[(N - j) * nx + 8] = run[j];
        std::vector<Affine2> am(N);
        std::vector<double> y2(2 * (N + 1));
        for (int j = 0; j < N; ++j) {
            const int k      = N - 1 - j;
            const double *M  = &Mc[9 * k]; // M[3 e + i]: d lam_a(k)[i] / d lam_a(k+1)[e]
            const double lps = lam_all[(k + 1) * nx + 8];
            am[j]            = {{{M[0], M[3]}, {M[1], M[4]}},
                                {std::fma(M[6], lps, ctot[3 * k]), std::fma(M[7], lps, ctot[3 * k + 1])}};
        }
        const double lN[2] = {lam_all[N * nx + 6], lam_all[N * nx + 7]};
        affine_scan(am.data(), N, lN, y2.data());
        for (int j = 0; j <= N; ++j) {
            lam_all[(N - j) * nx + 6] = y2[2 * j];
            lam_all[(N - j) * nx + 7] = y2[2 * j + 1];
        }
        for (int k = 0; k < N; ++k) {
            const double *ln = &lam_all[(k + 1) * nx + 6];
            for (int i = 0; i < 3; ++i) {
                double t = cu[3 * k + i];
                for (int e = 0; e < 3; ++e)
                    t = std::fma(Uc[9 * k + 3 * e + i], ln[e], t);
                grad[nu * k + 1 + i] = t + cw[3 * k + i];
            }
        }
        if (LAM)
            std::copy(lam_all.begin(), lam_all.end(), LAM);
    }

    // ---- the second-order tape ------------------------------------------------------------------
    // Recorded by grad_L (the point PANTR's Hessian-vector products follow).  With z = (x_k, u_k),
    // x = (p, v, a) (position, velocity, attitude), u = (thrust, w) and the multiplier estimate
    // yhat_k(x_k) of the stage, a product needs
    //   J_k = d x_{k+1} / d z                and
    //   H_k = d/dz of the stage's adjoint outputs [lam_k ; grad_k] at fixed lam_{k+1}
    //       = sum_i lam_{k+1,i} hess F_i + hess l + d(dc' yhat)/dz.
    // Both have the block structure of the cascade attitude -> velocity -> position:
    //   p+ = p + cpv v + Jp (a, u)     v+ = v + Jv (a, u)     a+ = Ja (a, w)  (quadcopter: a + cw w)
    //   H  = blockdiag(Hpp, hvv I, Hau),  Hau over (a, u): 7 x 7
    // (f_c is linear in p and v and the cost separates; the constraints touch p and a apart).  Only
    // the nonzero blocks are recorded - one dual evaluation of the stage per direction of (a, u), a
    // dual evaluation of the position terms of cost and constraints per direction of p; cpv = c6 * 6
    // and hvv = 2 q_v are constants - and the product runs over these blocks, in the order written
    // in hess_prod_tape() (which is the specification the kernels follow).
    static constexpr int nz = nx + nu, nau = 3 + nu; ///< nau: directions (a_0..2, u_0..3)
    static constexpr bool angle_rates_are_inputs = !Model::realistic;
    struct StageTape {
        double Jp[3][nau], Jv[3][nau]; ///< d p+_i / d (a,u)_e, d v+_i / d (a,u)_e
        double Ja[3][6];               ///< d a+_i / d (a_0..2, w_0..2)
        double Hau[nau][nau];          ///< [l][e]: derivative along (a,u)_l of adjoint output (a,u)_e
        double Hpp[3][3];              ///< [l][e]: derivative along p_l of adjoint output p_e
    };
    mutable std::vector<StageTape> tape_;
    mutable double HNpp_[3][3], HNaa_[3][3];
    static constexpr double cpv = Model::c6 * 6.0; ///< d p+ / d v
    static constexpr double hvv = 2.0 * 1.0, hvvN = 2.0 * 25.0; ///< d2 l / d v2, stage and terminal cost
    mutable std::vector<double> tape_u_;
    mutable bool tape_valid_ = false;

    template <class T>
    void stage_multipliers(const T *s, int k, T *yk) const {
        T c[nc];
        Model::constr(s, c);
        for (int i = 0; i < nc; ++i) {
            int j  = nc * k + i;
            T zeta = c[i] + (y[j] / Sig[j]);
            T zhat = tmax(tmin(zeta, d_ub(j)), d_lb(j));
            T d    = zeta - zhat;
            yk[i]  = Sig[j] * d;
        }
    }
    /// Derivative along state direction d of the state-only adjoint terms of stage k (cost gradient
    /// and constraint adjoint with the multiplier estimate of the stage): row d of hess l + d(dc' yhat)/dx.
    void state_terms_dual(const double *s_, const double *uk, int k, int d, Dual *lt) const {
        Dual s[nx], yk[nc], gu[nu], ud[nu];
        for (int l = 0; l < nx; ++l) {
            s[l]  = {s_[l], l == d ? 1.0 : 0.0};
            lt[l] = Dual{};
        }
        stage_multipliers(s, k, yk);
        if (uk) {
            for (int c = 0; c < nu; ++c)
                ud[c] = {uk[c], 0.0};
            Model::cost_grad(s, ud, lt, gu);
        } else {
            Model::cost_grad(s, static_cast<const Dual *>(nullptr), lt, gu);
        }
        Model::constr_adj(s, yk, lt);
    }
    /// The adjoint sweep of reverse<double>() (same gradient, bit for bit) that also records the tape.
    void reverse_record(const double *u, const double *X, double *grad) const {
        tape_.resize(N);
        tape_u_.assign(u, u + n());
        tape_valid_ = true;
        double lam[nx];
        {
            double gu[nu];
            for (int l = 0; l < nx; ++l)
                lam[l] = 0.0;
            Model::cost_grad(X + N * nx, static_cast<const double *>(nullptr), lam, gu);
            std::vector<double> yN(nc);
            stage_multipliers(X + N * nx, N, yN.data());
            Model::constr_adj(X + N * nx, yN.data(), lam);
            Dual lt[nx];
            for (int d = 0; d < 3; ++d) {
                state_terms_dual(X + N * nx, nullptr, N, d, lt);
                for (int e = 0; e < 3; ++e)
                    HNpp_[d][e] = lt[e].t;
                state_terms_dual(X + N * nx, nullptr, N, 6 + d, lt);
                for (int e = 0; e < 3; ++e)
                    HNaa_[d][e] = lt[6 + e].t;
            }
        }
        // realistic_quadcopter: gradient and adjoints as reverse() forms them (reverse_affine); the dual
        // evaluations below then only supply the tangent parts
        std::vector<double> LAM;
        if (Model::realistic) {
            std::vector<double> yh(m());
            for (int k = 0; k <= N; ++k)
                stage_multipliers(X + k * nx, k, yh.data() + nc * k);
            LAM.resize((N + 1) * nx);
            reverse_affine(u, X, yh.data(), grad, LAM.data());
        }
        for (int k = N - 1; k >= 0; --k) {
            StageTape &tp = tape_[k];
            double lam_k[nx];
            if (Model::realistic)
                std::copy(&LAM[(k + 1) * nx], &LAM[(k + 1) * nx] + nx, lam);
            for (int l = 0; l < nau; ++l) { // direction (a,u)_l = z_{6 + l}
                const int d = 6 + l;
                Dual s[nx], ud[nu], xn[nx], yk[nc], mu[nx], sbar[nx], ubar[nu], gu[nu];
                for (int i = 0; i < nx; ++i) {
                    s[i]  = {X[k * nx + i], i == d ? 1.0 : 0.0};
                    mu[i] = {lam[i], 0.0};
                }
                for (int c = 0; c < nu; ++c)
                    ud[c] = {u[nu * k + c], nx + c == d ? 1.0 : 0.0};
                rk4(s, ud, xn);
                for (int i = 0; i < 3; ++i) {
                    tp.Jp[i][l] = xn[i].t;
                    tp.Jv[i][l] = xn[3 + i].t;
                    if (l != 3) // the attitude does not depend on the thrust
                        tp.Ja[i][l < 3 ? l : l - 1] = xn[6 + i].t;
                }
                stage_multipliers(s, k, yk);
                rk4_adj(s, ud, mu, sbar, ubar);
                Model::cost_grad(s, ud, sbar, gu);
                for (int c = 0; c < nu; ++c) {
                    Dual g           = ubar[c] + gu[c];
                    tp.Hau[l][3 + c] = g.t;
                    if (!Model::realistic)
                        grad[nu * k + c] = g.v;
                }
                Model::constr_adj(s, yk, sbar);
                for (int e = 0; e < 3; ++e)
                    tp.Hau[l][e] = sbar[6 + e].t;
                for (int i = 0; i < nx; ++i)
                    lam_k[i] = sbar[i].v;
            }
            Dual lt[nx];
            for (int d = 0; d < 3; ++d) {
                state_terms_dual(X + k * nx, u + nu * k, k, d, lt);
                for (int e = 0; e < 3; ++e)
                    tp.Hpp[d][e] = lt[e].t;
            }
            std::copy(lam_k, lam_k + nx, lam);
        }
    }
    /// c[0] z[0] + c[1] z[1] + ... as one fma chain in index order, on top of `acc` (CONT) or from the
    /// first product.
    template <int LEN, bool CONT = false>
    static double dot_seq(const double *c, const double *z, double acc = 0.0) {
        int e = 0;
        if (!CONT) {
            acc = c[0] * z[0];
            e   = 1;
        }
        for (; e < LEN; ++e)
            acc = std::fma(c[e], z[e], acc);
        return acc;
    }
    /// Running sums in the order the kernels' wavefront scan forms them.  excl[k] = init + r[0] + ... +
    /// r[k-1] for k = 0..n, with the additions grouped as follows: blocks of 64 consecutive entries (a
    /// short block is padded with +0.0); inside a block a Kogge-Stone scan at distances 1, 2, 4, 8 within
    /// groups of 16 (an entry without a partner adds +0.0), then the odd groups add the total of the
    /// group before them, then the upper half adds the total of the lower half; an entry's sum is
    /// (sum of the blocks before) + (sum of its predecessors in its block).
    static void prefix_scan(const double *r, int n, double init, double *excl /* n + 1 */) {
        double carry = init;
        for (int k0 = 0; k0 < n; k0 += 64) {
            double x[64], t[64];
            for (int i = 0; i < 64; ++i)
                x[i] = k0 + i < n ? r[k0 + i] : 0.0;
            for (int d = 1; d <= 8; d *= 2) {
                for (int i = 0; i < 64; ++i)
                    t[i] = x[i] + ((i & 15) >= d ? x[i - d] : 0.0);
                std::copy(t, t + 64, x);
            }
            for (int i = 0; i < 64; ++i)
                t[i] = x[i] + ((i & 16) ? x[(i & ~15) - 1] : 0.0);
            std::copy(t, t + 64, x);
            for (int i = 0; i < 64; ++i)
                t[i] = x[i] + ((i & 32) ? x[31] : 0.0);
            std::copy(t, t + 64, x);
            const int cnt = std::min(64, n - k0);
            for (int i = 0; i < cnt; ++i)
                excl[k0 + i] = carry + (i ? x[i - 1] : 0.0);
            carry = carry + x[cnt - 1];
        }
        excl[n] = carry;
    }
    /// An affine map y -> A y + b of a pair, and (later o earlier) with the products in this order.
    struct Affine2 {
        double A[2][2], b[2];
        static Affine2 identity() { return {{{1.0, 0.0}, {0.0, 1.0}}, {0.0, 0.0}}; }
        void apply(const double *y, double *out) const {
            for (int i = 0; i < 2; ++i)
                out[i] = std::fma(A[i][1], y[1], std::fma(A[i][0], y[0], b[i]));
        }
        static Affine2 compose(const Affine2 &later, const Affine2 &earlier) {
            Affine2 c;
            for (int i = 0; i < 2; ++i) {
                for (int j = 0; j < 2; ++j)
                    c.A[i][j] = std::fma(later.A[i][1], earlier.A[1][j], later.A[i][0] * earlier.A[0][j]);
                c.b[i] = std::fma(later.A[i][1], earlier.b[1], std::fma(later.A[i][0], earlier.b[0], later.b[i]));
            }
            return c;
        }
    };
    /// y(k+1) = m[k](y(k)) for a sequence of affine maps of a pair, with the compositions grouped like
    /// the additions of prefix_scan(): blocks of 64 consecutive maps (a short block is padded with the
    /// identity); inside a block Kogge-Stone at distances 1, 2, 4, 8 within groups of 16 (an entry
    /// without a partner composes with the identity), then the odd groups compose with the total of the
    /// group before them, then the upper half with the total of the lower half; y(k+1) = (composition
    /// of the block's maps up to k) applied to the block's initial value.  out: y(0..n), 2 values each.
    static void affine_scan(const Affine2 *m, int n, const double *y0, double *out /* 2 (n + 1) */) {
        double carry[2] = {y0[0], y0[1]};
        for (int k0 = 0; k0 < n; k0 += 64) {
            Affine2 x[64], t[64];
            const Affine2 id = Affine2::identity();
            for (int i = 0; i < 64; ++i)
                x[i] = k0 + i < n ? m[k0 + i] : id;
            for (int d = 1; d <= 8; d *= 2) {
                for (int i = 0; i < 64; ++i)
                    t[i] = Affine2::compose(x[i], (i & 15) >= d ? x[i - d] : id);
                std::copy(t, t + 64, x);
            }
            for (int i = 0; i < 64; ++i)
                t[i] = Affine2::compose(x[i], (i & 16) ? x[(i & ~15) - 1] : id);
            std::copy(t, t + 64, x);
            for (int i = 0; i < 64; ++i)
                t[i] = Affine2::compose(x[i], (i & 32) ? x[31] : id);
            std::copy(t, t + 64, x);
            const int cnt = std::min(64, n - k0);
            out[2 * k0] = carry[0], out[2 * k0 + 1] = carry[1];
            for (int i = 0; i < cnt; ++i)
                x[i].apply(carry, out + 2 * (k0 + i + 1));
            carry[0] = out[2 * (k0 + cnt)], carry[1] = out[2 * (k0 + cnt) + 1];
        }
        if (n == 0)
            out[0] = carry[0], out[1] = carry[1];
    }
    /// Hv over the tape.  Forward: the tangent trajectory x' = (p', v', a'); each state is its initial
    /// value plus the running sum of increments that do not involve it, so the increments are formed
    /// for all stages first and summed by prefix_scan (attitude of realistic_quadcopter: a 3 x 3
    /// recursion, stage by stage).  Backward: lam' = (lp, lv, la) and the products, the same way over
    /// the stages in reverse order.  Output l takes the row recorded along direction l as its
    /// coefficients (H is symmetric up to rounding).
    void hess_prod_tape(const double *v, double *Hv) const {
        std::vector<double> Xt((N + 1) * nx, 0.0), inc(N), run(N + 1);
        auto xt = [&](int k, int i) -> double & { return Xt[k * nx + i]; };
        // attitude
        for (int i = 0; i < 3; ++i) {
            if (angle_rates_are_inputs) {
                for (int k = 0; k < N; ++k)
                    inc[k] = tape_[k].Ja[i][3 + i] * v[nu * k + 1 + i];
                prefix_scan(inc.data(), N, 0.0, run.data());
                for (int k = 0; k <= N; ++k)
                    xt(k, 6 + i) = run[k];
            }
        }
        if (!angle_rates_are_inputs) {
            // (phi', theta') is an affine recursion (the yaw does not feed back): affine_scan; the yaw's
            // tangent is a running sum of increments that involve the other two
            std::vector<Affine2> am(N);
            std::vector<double> y2(2 * (N + 1));
            for (int k = 0; k < N; ++k) {
                const auto &Ja = tape_[k].Ja;
                am[k]          = {{{Ja[0][0], Ja[0][1]}, {Ja[1][0], Ja[1][1]}},
                                  {dot_seq<3>(Ja[0] + 3, v + nu * k + 1), dot_seq<3>(Ja[1] + 3, v + nu * k + 1)}};
            }
            const double zero2[2] = {0.0, 0.0};
            affine_scan(am.data(), N, zero2, y2.data());
            for (int k = 0; k <= N; ++k)
                xt(k, 6) = y2[2 * k], xt(k, 7) = y2[2 * k + 1];
            for (int k = 0; k < N; ++k) {
                const auto &Ja  = tape_[k].Ja;
                const double Sw = dot_seq<3>(Ja[2] + 3, v + nu * k + 1);
                inc[k]          = std::fma(Ja[2][1], xt(k, 7), std::fma(Ja[2][0], xt(k, 6), Sw));
            }
            prefix_scan(inc.data(), N, 0.0, run.data());
            for (int k = 0; k <= N; ++k)
                xt(k, 8) = run[k];
        }
        // velocity, position
        std::vector<double> Rp(3 * N);
        for (int i = 0; i < 3; ++i) {
            for (int k = 0; k < N; ++k) {
                const double *vk    = v + nu * k;
                const double z[nau] = {xt(k, 6), xt(k, 7), xt(k, 8), vk[0], vk[1], vk[2], vk[3]};
                inc[k]              = dot_seq<nau>(tape_[k].Jv[i], z);
                Rp[3 * k + i]       = dot_seq<nau>(tape_[k].Jp[i], z);
            }
            prefix_scan(inc.data(), N, 0.0, run.data());
            for (int k = 0; k <= N; ++k)
                xt(k, 3 + i) = run[k];
            for (int k = 0; k < N; ++k)
                inc[k] = std::fma(cpv, xt(k, 3 + i), Rp[3 * k + i]);
            prefix_scan(inc.data(), N, 0.0, run.data());
            for (int k = 0; k <= N; ++k)
                xt(k, i) = run[k];
        }
        // backward, over j = N - 1 - k; LM[k] = lam'_k
        std::vector<double> LM((N + 1) * nx, 0.0);
        auto lm = [&](int k, int i) -> double & { return LM[k * nx + i]; };
        auto suffix = [&](int comp) { // lm(k, comp) = lm(N, comp) + inc[N-1-k] + ... (inc in reversed order)
            prefix_scan(inc.data(), N, lm(N, comp), run.data());
            for (int j = 0; j <= N; ++j)
                lm(N - j, comp) = run[j];
        };
        for (int i = 0; i < 3; ++i) {
            lm(N, i)     = dot_seq<3>(HNpp_[i], &xt(N, 0));
            lm(N, 3 + i) = hvvN * xt(N, 3 + i);
            lm(N, 6 + i) = dot_seq<3>(HNaa_[i], &xt(N, 6));
        }
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < N; ++j)
                inc[j] = dot_seq<3>(tape_[N - 1 - j].Hpp[i], &xt(N - 1 - j, 0));
            suffix(i);
            for (int j = 0; j < N; ++j)
                inc[j] = std::fma(cpv, lm(N - j, i), hvv * xt(N - 1 - j, 3 + i));
            suffix(3 + i);
        }
        std::vector<double> T(nau * N);
        for (int k = 0; k < N; ++k) {
            const StageTape &tp = tape_[k];
            const double *vk    = v + nu * k;
            const double z[nau] = {xt(k, 6), xt(k, 7), xt(k, 8), vk[0], vk[1], vk[2], vk[3]};
            for (int l = 0; l < nau; ++l) {
                double acc = dot_seq<nau>(tp.Hau[l], z);
                for (int i = 0; i < 3; ++i)
                    acc = std::fma(tp.Jp[i][l], lm(k + 1, i), acc);
                for (int i = 0; i < 3; ++i)
                    acc = std::fma(tp.Jv[i][l], lm(k + 1, 3 + i), acc);
                T[nau * k + l] = acc;
            }
            Hv[nu * k] = T[nau * k + 3];
        }
        if (angle_rates_are_inputs) {
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < N; ++j)
                    inc[j] = T[nau * (N - 1 - j) + i];
                suffix(6 + i);
                for (int k = 0; k < N; ++k)
                    Hv[nu * k + 1 + i] = std::fma(tape_[k].Ja[i][3 + i], lm(k + 1, 6 + i), T[nau * k + 4 + i]);
            }
        } else {
            // yaw: lam_psi(k) = lam_psi(k+1) + T_psi(k); then (lam_phi, lam_theta) by the transposed affine
            // recursion over the reversed stages, its constant term holding the yaw's contribution
            for (int j = 0; j < N; ++j)
                inc[j] = T[nau * (N - 1 - j) + 2];
            suffix(8);
            std::vector<Affine2> am(N);
            std::vector<double> y2(2 * (N + 1));
            for (int j = 0; j < N; ++j) {
                const int k    = N - 1 - j;
                const auto &Ja = tape_[k].Ja;
                am[j]          = {{{Ja[0][0], Ja[1][0]}, {Ja[0][1], Ja[1][1]}},
                                  {std::fma(Ja[2][0], lm(k + 1, 8), T[nau * k]), std::fma(Ja[2][1], lm(k + 1, 8), T[nau * k + 1])}};
            }
            const double lN[2] = {lm(N, 6), lm(N, 7)};
            affine_scan(am.data(), N, lN, y2.data());
            for (int j = 0; j <= N; ++j)
                lm(N - j, 6) = y2[2 * j], lm(N - j, 7) = y2[2 * j + 1];
            for (int k = 0; k < N; ++k) {
                const StageTape &tp = tape_[k];
                for (int j = 0; j < 3; ++j) {
                    double acc = T[nau * k + 4 + j];
                    for (int i = 0; i < 3; ++i)
                        acc = std::fma(tp.Ja[i][3 + j], lm(k + 1, 6 + i), acc);
                    Hv[nu * k + 1 + j] = acc;
                }
            }
        }
    }

    double eval_psi(const double *u, double *yhat) const {
        ++cnt.psi;
        std::vector<double> X((N + 1) * nx);
        return forward(u, X.data(), yhat);
    }
    double psi_grad_impl(const double *u, double *grad) const {
        std::vector<double> X((N + 1) * nx), yh(m());
        double psi = forward(u, X.data(), yh.data());
        reverse(u, X.data(), yh.data(), grad);
        return psi;
    }
    double eval_psi_grad_psi(const double *u, double *grad) const {
        ++cnt.psi_grad_psi;
        return psi_grad_impl(u, grad);
    }
    void eval_grad_psi(const double *u, double *grad) const {
        ++cnt.grad_psi;
        psi_grad_impl(u, grad);
    }
    /// grad_L(xhat, yhat(xhat)) = grad psi(xhat)  (formulation.hpp:98-99 with y = yhat)
    /// (PANTR evaluates it at the point its Hessian-vector products follow, so this is where the
    /// second-order tape is recorded, as in the kernels.)
    void eval_grad_L(const double *u, double *grad, bool record = true) const {
        ++cnt.grad_L;
        std::vector<double> X((N + 1) * nx), yh(m());
        forward(u, X.data(), yh.data());
        if (record)
            reverse_record(u, X.data(), grad);
        else
            reverse(u, X.data(), yh.data(), grad);
    }
    /// The Hessian-vector product of the problem (formulation.hpp:93-97), over the tape.
    void eval_hess_psi_prod(const double *u, const double *v, double *Hv) const {
        ++cnt.hess_psi_prod;
        if (!tape_valid_ || tape_u_.size() != (size_t)n() || !std::equal(u, u + n(), tape_u_.begin())) {
            std::vector<double> X((N + 1) * nx), yh(m()), g(n()); // tape is for another point
            forward(u, X.data(), yh.data());
            reverse_record(u, X.data(), g.data());
        }
        hess_prod_tape(v, Hv);
    }
    /// Hessian-vector product of psi by forward-over-reverse on dual numbers through the whole
    /// horizon (formulation.hpp:93-97: gradient(jtimes(aug_lagr, u, v), u)).  Independent check of
    /// eval_hess_psi_prod (agrees to rounding, not bit for bit).
    void eval_hess_psi_prod_dual(const double *u, const double *v, double *Hv) const {
        std::vector<Dual> ud(n()), X((N + 1) * nx), yh(m()), g(n());
        for (int i = 0; i < n(); ++i)
            ud[i] = {u[i], v[i]};
        forward(ud.data(), X.data(), yh.data());
        reverse(ud.data(), X.data(), yh.data(), g.data());
        for (int i = 0; i < n(); ++i)
            Hv[i] = g[i].t;
    }
    double eval_f(const double *u) const { // formulation.hpp:57-60
        std::vector<double> X((N + 1) * nx), zero(m(), 0.0), one(m(), 1.0);
        SingleShootingAL tmp(*this);
        tmp.set_al(zero.data(), one.data());
        std::vector<double> s(nx), xn(nx);
        double acc = 0;
        std::copy(x0.begin(), x0.end(), s.begin());
        for (int k = 0; k < N; ++k) {
            acc = acc + Model::stage_cost(s.data(), u + nu * k);
            rk4(s.data(), u + nu * k, xn.data());
            s = xn;
        }
        return acc + Model::term_cost(s.data());
    }
    void eval_g(const double *u, double *g) const { // formulation.hpp:64-67
        std::vector<double> s(x0), xn(nx);
        for (int k = 0; k <= N; ++k) {
            Model::constr(s.data(), g + nc * k);
            if (k == N)
                break;
            rk4(s.data(), u + nu * k, xn.data());
            s = xn;
        }
    }
    void eval_prox_grad_step(double gamma, const double *x, const double *grad, double *xh,
                             double *p) const {
        ++cnt.prox_grad_step;
        for (int i = 0; i < n(); ++i) {
            double t = (-gamma) * grad[i];
            p[i]     = std::fmin(std::fmax(t, lb(i) - x[i]), ub(i) - x[i]);
            xh[i]    = x[i] + p[i];
        }
    }
    int eval_inactive_indices_res_lna(double gamma, const double *x, const double *grad,
                                      unsigned char *mask) const {
        int nJ = 0;
        for (int i = 0; i < n(); ++i) {
            double xfw = x[i] - gamma * grad[i];
            mask[i]    = lb(i) < xfw && xfw < ub(i);
            nJ += mask[i];
        }
        return nJ;
    }
    static void simulate_dynamics(const double *x, const double *u, double *xn) { rk4(x, u, xn); }
};

} // namespace oracle
