// Quadcopter and realistic_quadcopter stage models (gfx950), scalar form.
//
// What the reference computes (paths relative to /root/reference; `method=pantr` loads the CasADi
// problem generated from the Python models, problems/generate_problem.py:26-61):
//   quadcopter ............ python/alpaqa_mpc_benchmarks/problems/quadcopter.py:6-112
//   realistic_quadcopter .. python/alpaqa_mpc_benchmarks/problems/realistic_quadcopter.py:6-118
// The stage is scalar code, called with lane = stage, lane = (stage, direction) or - only in the attitude
// integration of realistic_quadcopter - with the same values on every lane (trig_uniform() then
// evaluates one angle per lane and broadcasts).  The pieces the cascade sweeps of pantr_alm.hip.h are
// built from are separate functions: accel, angle_rates, fc_adj_translation, angle_rates_adj_coef /
// _apply; fc and fc_adj are their compositions (the dual-number recording and the plant step use
// those).  sin/cos values come from wave_math's deterministic routine and are handed around as a Trig
// instead of being recomputed.  First derivatives are hand-written adjoints; second-order information
// comes from running the same templates on dual numbers whose tangent parts differ per lane.
#pragma once
#include "wave_math.hip.h"

namespace pantr {

template <bool REALISTIC>
struct QuadModelT {
    static constexpr int nx = 9, nu = 4, nc = 4;
    /// Both models are a cascade angles -> velocities -> positions (the attitude does not depend on the
    /// translation), which ScalarALProblem's sweeps use.  In quadcopter the angle rates are inputs, so
    /// the attitude is a plain sum over the stages; in realistic_quadcopter they depend on the angles
    /// and the attitude is integrated stage by stage.
    static constexpr bool angles_are_integrated_inputs = !REALISTIC;
    static constexpr double Ts = 0.1, hTs = 0.5 * Ts, g_grav = 9.81, c6 = Ts / 6;
    static constexpr double r2 = 0.1 * 0.1; ///< collision cylinder radius squared (quadcopter.py:70)
    __host__ __device__ static constexpr double pf(int i) { return i == 2 ? 0.5 : 0.25; } // quadcopter.py:90
    __host__ __device__ static const char *name() { return REALISTIC ? "realistic_quadcopter" : "quadcopter"; }

    __host__ __device__ static double u_lb(int c) { return c == 0 ? 0.0 : -0.1; }          // quadcopter.py:55-58
    __host__ __device__ static double u_ub(int c) { return c == 0 ? 9.81 * 5 : 0.1; }
    __host__ __device__ static double d_lb(int i) { // quadcopter.py:72-75
        return i < 2 ? -1.5707963267948966 : i == 2 ? 0.8525245220595057 /* cos(1.1/2) */ : -inf_();
    }
    __host__ __device__ static double d_ub(int i) { return i < 2 ? 1.5707963267948966 : i == 2 ? inf_() : 0.0; }
    __host__ __device__ static constexpr double inf_() { return __builtin_huge_val(); }
    __host__ __device__ static void initial_state(double *x0) { // quadcopter.py:80-88
        const double v[9] = {-0.2, -0.25, 0.5, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 9; ++i)
            x0[i] = v[i];
    }
    __host__ __device__ static double initial_guess(int c) { return c == 0 ? 9.81 : 0.0; } // quadcopter.py:111

    /// Third column of the body-to-world rotation and, for the realistic model, the Euler-rate
    /// map d_orient = Omega^{-1} omega, with the trigonometric values they are built from.
    template <class T>
    struct Trig {
        T s0, c0, s1, c1, s2, c2;
    };
    /// Every lane holds its own state (one instance per lane).
    template <class T>
    __device__ static __forceinline__ Trig<T> trig(const T *s) {
        Trig<T> t;
        sincos_t(s[6], t.s0, t.c0);
        sincos_t(s[7], t.s1, t.c1);
        sincos_t(s[8], t.s2, t.c2);
        return t;
    }
    /// The state's values are the same on all 64 lanes (all active): lane j < 3 of every 16-lane row
    /// evaluates angle j, the six results are broadcast inside the rows (v_mov_b64_dpp row_newbcast).  Same routine, same inputs: bit-identical to trig().
    /// `store` (LDS, or nullptr): the three (sin, cos) pairs are also kept for the sweeps that revisit
    /// this state (the adjoint sweep and the second-order recording read them instead of evaluating
    /// sin/cos again: TrigStore below).
    __device__ static __forceinline__ Trig<double> trig_uniform(double a0, double a1, double a2,
                                                                double2 *store = nullptr) {
        const int lane = threadIdx.x & 63, j = lane & 15; // lane j of every 16-lane row: angle j
        double sv, cv;
        sincos_det(j == 1 ? a1 : j == 2 ? a2 : a0, sv, cv);
        if (store) // every lane stores: lanes 3..15 of a row hold angle 0 again (same value, same address)
            store[j < 3 ? j : 0] = make_double2(sv, cv);
        return {row_bcast<0>(sv), row_bcast<0>(cv), row_bcast<1>(sv), row_bcast<1>(cv), row_bcast<2>(sv),
                row_bcast<2>(cv)};
    }
    __device__ static __forceinline__ Trig<double> trig_uniform(const double *s, double2 *store = nullptr) {
        return trig_uniform(s[6], s[7], s[8], store);
    }
    /// Trigonometric values of the four Runge-Kutta substage states of every stage, written by the
    /// forward sweep: entry (4 k + j) holds the three (sin, cos) pairs of substage j of stage k
    /// (j = 0: the stage state itself; stage N has only j = 0).
    static constexpr int TRIG_DOUBLES_PER_STAGE = 4 * 3 * 2;
    __device__ static __forceinline__ Trig<double> trig_load(const double2 *e) {
        const double2 a = e[0], b = e[1], c = e[2];
        return {a.x, a.y, b.x, b.y, c.x, c.y};
    }
    /// Dual numbers from stored values and the three angle tangents (same products as sincos_t(dual)).
    __device__ static __forceinline__ Trig<dual> trig_load(const double2 *e, double t0, double t1, double t2) {
        const Trig<double> v = trig_load(e);
        return {{v.s0, v.c0 * t0}, {v.c0, -(v.s0 * t0)}, {v.s1, v.c1 * t1},
                {v.c1, -(v.s1 * t1)}, {v.s2, v.c2 * t2}, {v.c2, -(v.s2 * t2)}};
    }
    /// Acceleration of the stage at thrust u0 for the attitude t: the expressions of fc() for ds[3..5].
    __device__ static __forceinline__ void accel(const Trig<double> &t, double u0, double (&a)[3]) {
        double r0, r1, r2_;
        if (!REALISTIC) {
            r0  = ((t.c2 * t.s1) * t.c0) + (t.s2 * t.s0);
            r1  = ((t.s2 * t.s1) * t.c0) - (t.c2 * t.s0);
            r2_ = t.c1 * t.c0;
        } else {
            r0  = (t.c2 * t.s1) + ((t.c1 * t.s0) * t.s2);
            r1  = (t.s2 * t.s1) - ((t.c2 * t.c1) * t.s0);
            r2_ = t.c0 * t.c1;
        }
        a[0] = r0 * u0;
        a[1] = r1 * u0;
        a[2] = (r2_ * u0) - g_grav;
    }

    /// f_c(state, input) (quadcopter.py:30-46, realistic_quadcopter.py:30-50)
    template <class T>
    __device__ static __forceinline__ void fc(const Trig<T> &t, const T *s, const T *u, T *ds) {
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
        angle_rates(t, u, ds + 6);
    }
    /// Rates of the three angles, ds[6..8] of f_c: the body rates themselves (quadcopter) or mapped
    /// through the inverse Euler-rate matrix (realistic_quadcopter).  They depend on the angles and the
    /// inputs only, which is what lets the sweeps integrate the attitude ahead of everything else.
    template <class T>
    __device__ static __forceinline__ void angle_rates(const Trig<T> &t, const T *u, T *da) {
        if (!REALISTIC) {
            da[0] = u[1];
            da[1] = u[2];
            da[2] = u[3];
        } else { // Omega [a b c]' = omega, Omega = [[c1,0,-c0 s1],[0,1,s0],[s1,0,c0 c1]], det = c0
            T a   = (t.c1 * u[1]) + (t.s1 * u[3]);
            T cc  = qdiv((t.c1 * u[3]) - (t.s1 * u[1]), t.c0);
            T b   = u[2] - (t.s0 * cc);
            da[0] = a;
            da[1] = b;
            da[2] = cc;
        }
    }

    /// Adjoint of f_c: sb = (df/ds)' kb, ub = (df/du)' kb.
    template <class T>
    __device__ static __forceinline__ void fc_adj(const Trig<T> &t, const T *u, const T *kb, T *sb, T *ub) {
        fc_adj_translation(t, u, kb, sb, ub);
        angle_rates_adj(t, u, kb + 6, sb + 6, ub + 1);
    }
    /// The part of fc_adj that the position and velocity rows of f_c contribute: all of sb, and ub[0]
    /// (the thrust); it reads kb[0..5] only.
    template <class T>
    __device__ static __forceinline__ void fc_adj_translation(const Trig<T> &t, const T *u, const T *kb, T *sb,
                                                              T *ub) {
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
    }
    /// Adjoint of angle_rates: with kba = kb[6..8], adds (d rates / d angles)' kba to sba = sb[6..8]
    /// and returns uba = ub[1..3] = (d rates / d u[1..3])' kba.  In two steps: what depends on the
    /// substage state and the inputs only (RateAdjCoef: the adjoint sweep evaluates it for all stages
    /// at once), and the products with the incoming adjoint.
    template <class T>
    struct RateAdjCoef {
        T ic, x, y, num, z;
    };
    template <class T>
    __device__ static __forceinline__ RateAdjCoef<T> angle_rates_adj_coef(const Trig<T> &t, const T *u) {
        // a = c1 u1 + s1 u3 ; cc = (c1 u3 - s1 u1)/c0 ; b = u2 - s0 cc
        RateAdjCoef<T> k;
        k.num      = (t.c1 * u[3]) - (t.s1 * u[1]);
        k.ic       = qrecip(t.c0); // one reciprocal serves the three quotients by cos(phi)
        const T cc = k.num * k.ic;
        k.x        = (cc * t.s0) * k.ic; // d cc / d phi = num s0 / c0^2
        k.y        = t.c0 * cc;          // -d b / d phi at fixed cc
        k.z        = (t.s1 * u[3]) + (t.c1 * u[1]); // -d num / d theta
        return k;
    }
    template <class T>
    __device__ static __forceinline__ void angle_rates_adj_apply(const T &s0, const T &s1, const T &c1,
                                                                 const RateAdjCoef<T> &k, const T *kba, T *sba,
                                                                 T *uba) {
        const T ka = kba[0], kb_ = kba[1], kc = kba[2];
        const T kcc = kc - (s0 * kb_); // total adjoint of cc
        const T q   = kcc * k.ic;      // adjoint of num
        uba[0] = (c1 * ka) - (s1 * q);
        uba[1] = kb_;
        uba[2] = (s1 * ka) + (c1 * q);
        sba[0] = sba[0] + ((k.x * kcc) - (k.y * kb_));
        sba[1] = sba[1] + ((k.num * ka) - (k.z * q)); // a -> num ka ; num -> -z q
    }
    template <class T>
    __device__ static __forceinline__ void angle_rates_adj(const Trig<T> &t, const T *u, const T *kba, T *sba,
                                                           T *uba) {
        if (!REALISTIC) {
            uba[0] = kba[0];
            uba[1] = kba[1];
            uba[2] = kba[2];
        } else {
            angle_rates_adj_apply(t.s0, t.s1, t.c1, angle_rates_adj_coef(t, u), kba, sba, uba);
        }
    }

    /// l(x,u) (quadcopter.py:93-106): lu + lx
    template <class T>
    __device__ static __forceinline__ T stage_cost(const T *s, const T *u) {
        T su  = (((u[0] * u[0]) + (u[1] * u[1])) + (u[2] * u[2])) + (u[3] * u[3]);
        T so  = ((u[1] * u[1]) + (u[2] * u[2])) + (u[3] * u[3]);
        T lu  = (1e-4 * su) + (10.0 * so);
        return lu + state_cost(s, 1.0, 10.0, 1.0);
    }
    /// l_N(x) (quadcopter.py:107-112)
    template <class T>
    __device__ static __forceinline__ T term_cost(const T *s) { return state_cost(s, 25.0, 250.0, 10.0); }
    template <class T>
    __device__ static __forceinline__ T state_cost(const T *s, double qv, double qp, double qth) {
        T d0 = s[0] - pf(0), d1 = s[1] - pf(1), d2 = s[2] - pf(2);
        T sp = ((d0 * d0) + (d1 * d1)) + (d2 * d2);
        T sv = ((s[3] * s[3]) + (s[4] * s[4])) + (s[5] * s[5]);
        T st = ((s[6] * s[6]) + (s[7] * s[7])) + (s[8] * s[8]);
        return ((qv * sv) + (qp * sp)) + (qth * st);
    }
    /// gs += dl/dx, gu = dl/du  (u == nullptr: terminal cost)
    template <class T>
    __device__ static __forceinline__ void cost_grad(const T *s, const T *u, T *gs, T *gu) {
        const double qv = u ? 1.0 : 25.0, qp = u ? 10.0 : 250.0, qth = u ? 1.0 : 10.0;
        for (int i = 0; i < 3; ++i) {
            gs[i]     = gs[i] + (2.0 * qp) * (s[i] - pf(i));
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
    __device__ static __forceinline__ void constr(const Trig<T> &t, const T *s, T *c) {
        c[0] = s[6];
        c[1] = s[7];
        c[2] = t.c0 * t.c1;
        c[3] = (r2 - (s[0] * s[0])) - (s[1] * s[1]);
    }
    /// sb += (dc/dx)' yb
    template <class T>
    __device__ static __forceinline__ void constr_adj(const Trig<T> &t, const T *s, const T *yb, T *sb) {
        sb[6] = sb[6] + (yb[0] - ((t.s0 * t.c1) * yb[2]));
        sb[7] = sb[7] + (yb[1] - ((t.c0 * t.s1) * yb[2]));
        sb[0] = sb[0] - ((2.0 * s[0]) * yb[3]);
        sb[1] = sb[1] - ((2.0 * s[1]) * yb[3]);
    }
};

} // namespace pantr
