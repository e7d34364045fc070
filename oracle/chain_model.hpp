// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/README.md).
//
// Plain scalar C++ restatement of the hanging-chain optimal-control problem of
// kul-optec/pantr-cdc2023-experiments, single-shooting formulation, with
// hand-written reverse-mode (gradient) and forward-over-reverse (Hessian-vector
// product) sweeps.  Nothing under pantr-cdc2023-experiments_amd/ may include
// this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg use it (as the checker / the CPU baseline).
//
// Reference text followed (paths relative to /root/reference):
//   model constants ........ problems/problems/hanging_chain.cpp:28-42
//   continuous dynamics .... problems/problems/hanging_chain.cpp:85-100
//   RK4 discretisation ..... problems/problems/formulation.hpp:21-28
//   stage / terminal cost .. problems/problems/hanging_chain.cpp:127-132
//   box on the inputs ...... problems/problems/hanging_chain.cpp:107-110
//   initial state/guess .... problems/problems/hanging_chain.cpp:121-124,135
//   single shooting ........ problems/problems/formulation.hpp:45-62 (mapaccum of f_d, sum of l + l_N)
//   psi / grad / Hv set .... problems/problems/formulation.hpp:79-97 (m = 0 here: psi == f)
//
// Arithmetic order.  The reference evaluates these expressions through CasADi
// (absent here), so the association order of its sums is not recoverable.  This
// file fixes one order (documented in DESIGN.md "Arithmetic specification") and
// the HIP kernels follow the same one, which makes GPU-vs-oracle parity exact.
// All reductions over a "lane" index l = 0..63 use the butterfly tree of
// wave_sum() below (the order a 64-wide wavefront xor-shuffle reduction gives).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace oracle {

// ---------------------------------------------------------------------------
// 64-lane butterfly reductions (the order of a wavefront xor-shuffle all-reduce)
// ---------------------------------------------------------------------------
inline double wave_sum(const double *part /*[64]*/) {
    double a[64], b[64];
    std::memcpy(a, part, sizeof(a));
    for (int m = 32; m >= 1; m >>= 1) {
        for (int i = 0; i < 64; ++i)
            b[i] = a[i] + a[i ^ m];
        std::memcpy(a, b, sizeof(a));
    }
    return a[0];
}

/// dot(a, b) over i < n: lane l accumulates i = l, l+64, ... in ascending order
/// (separate multiply and add, no FMA), then the 64-lane butterfly.
inline double dot(const double *a, const double *b, int n) {
    double part[64];
    for (int l = 0; l < 64; ++l) {
        double acc = 0.0;
        for (int i = l; i < n; i += 64)
            acc = acc + a[i] * b[i];
        part[l] = acc;
    }
    return wave_sum(part);
}

inline double norm_inf(const double *a, int n) {
    double r = 0.0;
    for (int i = 0; i < n; ++i)
        r = std::fmax(r, std::fabs(a[i]));
    return r;
}

// ---------------------------------------------------------------------------
// First-order dual number (value, tangent) with a fixed operation order
// ---------------------------------------------------------------------------
struct Dual {
    double v, t;
};
inline Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.t + b.t}; }
inline Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.t - b.t}; }
inline Dual operator*(Dual a, Dual b) { return {a.v * b.v, std::fma(a.t, b.v, a.v * b.t)}; }
inline Dual operator*(double c, Dual a) { return {c * a.v, c * a.t}; }
inline Dual operator+(Dual a, double c) { return {a.v + c, a.t}; }
inline Dual operator-(Dual a, double c) { return {a.v - c, a.t}; }
inline Dual operator-(double c, Dual a) { return {c - a.v, 0.0 - a.t}; }
/// c / b = q with tangent b.t * k, k = (-q) / b.v  (every division acts on primal values only,
/// so a kernel may record q and k once and replay tangents without dividing)
inline Dual operator/(double c, Dual b) {
    double q = c / b.v;
    double k = (-q) / b.v;
    return {q, b.t * k};
}
/// a / b = q with tangent (a.t - q b.t) / b.v
inline Dual operator/(Dual a, Dual b) {
    double q = a.v / b.v;
    return {q, (a.t - q * b.t) / b.v};
}
/// sqrt(a) = r with tangent a.t * h, h = 0.5 / r
inline Dual sqrt(Dual a) {
    double r = std::sqrt(a.v);
    double h = 0.5 / r;
    return {r, a.t * h};
}
inline double sqrt(double a) { return std::sqrt(a); }
inline double value(double a) { return a; }
inline double value(Dual a) { return a.v; }

// ---------------------------------------------------------------------------
// Hanging chain model
// ---------------------------------------------------------------------------
struct Chain {
    // hanging_chain.cpp:28-42 (HangingChainConfig defaults)
    static constexpr int NB = 9;              ///< number of free masses (N_balls)
    static constexpr int DIM = 3;             ///< spatial dimensions
    static constexpr int NP = NB + 1;         ///< points = masses + actuated end
    static constexpr int nx = DIM * NP + DIM * NB; ///< 57
    static constexpr int nu = DIM;                 ///< 3
    static constexpr int XOFF = 0;                 ///< positions at [0, 30)
    static constexpr int VOFF = DIM * NP;          ///< velocities at [30, 57)
    static constexpr double Ts = 0.1;
    static constexpr double alpha = 25, beta = 1, gamma = 0.01;
    static constexpr double mass = 0.03, Dspring = 0.1, Lspring = 0.033;
    static constexpr double v_max = 1, g_grav = 9.81;
    static constexpr double inv_m = 1.0 / mass;
    static constexpr double DL = Dspring * Lspring;
    static constexpr double c6 = Ts / 6;
    static constexpr double c2a = 2 * alpha, c2b = 2 * beta, c2g = 2 * gamma;
    static double x_end(int c) { return c == 0 ? 1.0 : 0.0; } // hanging_chain.cpp:83
    static double u_lb(int) { return -v_max; }                // hanging_chain.cpp:107-110
    static double u_ub(int) { return +v_max; }

    /// hanging_chain.cpp:121-124
    static void initial_state(double *x0) {
        std::fill(x0, x0 + nx, 0.0);
        for (int i = 0; i < NP; ++i)
            x0[i * DIM] = static_cast<double>(i + 1) / static_cast<double>(NB + 1);
    }

    /// Per-point spring quantities kept from the forward pass for the adjoint.
    template <class T>
    struct Springs {
        T d[NP][DIM]; ///< dist_vect
        T n2[NP];     ///< squared length
        T r[NP];      ///< length
        T c[NP];      ///< D (1 - L / r)
    };

    /// dist_vect, norms and spring coefficients (hanging_chain.cpp:87-94)
    template <class T>
    static void springs(const T *s, Springs<T> &sp) {
        for (int p = 0; p < NP; ++p) {
            for (int c = 0; c < DIM; ++c)
                sp.d[p][c] = p == 0 ? s[c] : s[DIM * p + c] - s[DIM * (p - 1) + c];
            sp.n2[p] = (sp.d[p][0] * sp.d[p][0] + sp.d[p][1] * sp.d[p][1]) +
                       sp.d[p][2] * sp.d[p][2];
            sp.r[p] = sqrt(sp.n2[p]);
            sp.c[p] = Dspring * (1.0 - Lspring / sp.r[p]);
        }
    }

    /// Continuous-time dynamics f_c (hanging_chain.cpp:85-100).
    template <class T>
    static void fc(const T *s, const T *u, T *ds, Springs<T> &sp) {
        springs(s, sp);
        T F[NP][DIM];
        for (int p = 0; p < NP; ++p)
            for (int c = 0; c < DIM; ++c)
                F[p][c] = sp.d[p][c] * sp.c[p];
        for (int l = 0; l < DIM * NB; ++l) // x_i' = v_i
            ds[l] = s[VOFF + l];
        for (int c = 0; c < DIM; ++c) // x_end' = u
            ds[DIM * NB + c] = u[c];
        for (int i = 0; i < NB; ++i)
            for (int c = 0; c < DIM; ++c) {
                T a = inv_m * (F[i + 1][c] - F[i][c]);
                if (c == DIM - 1)
                    a = a - g_grav;
                ds[VOFF + DIM * i + c] = a;
            }
    }

    /// Adjoint of f_c at the point whose spring data is sp: given kbar (adjoint
    /// of ds) returns abar (adjoint of s); the adjoint of u is kbar[27..29].
    template <class T>
    static void fc_adj(const Springs<T> &sp, const T *kbar, T *abar) {
        T ld[NP][DIM];
        for (int p = 0; p < NP; ++p) {
            T lF[DIM], pr[DIM];
            for (int c = 0; c < DIM; ++c) {
                // a_{p-1} depends on +F_p, a_p on -F_p
                if (p == 0)
                    lF[c] = inv_m * (0.0 - kbar[VOFF + DIM * p + c]);
                else if (p == NP - 1)
                    lF[c] = inv_m * (kbar[VOFF + DIM * (p - 1) + c] - 0.0);
                else
                    lF[c] = inv_m * (kbar[VOFF + DIM * (p - 1) + c] - kbar[VOFF + DIM * p + c]);
                pr[c] = sp.d[p][c] * lF[c];
            }
            T dt = (pr[0] + pr[1]) + pr[2];
            T w  = DL / (sp.n2[p] * sp.r[p]);
            T t  = dt * w;
            for (int c = 0; c < DIM; ++c)
                ld[p][c] = (sp.c[p] * lF[c]) + (t * sp.d[p][c]);
        }
        for (int p = 0; p < NP; ++p)
            for (int c = 0; c < DIM; ++c)
                abar[DIM * p + c] = p == NP - 1 ? ld[p][c] : ld[p][c] - ld[p + 1][c];
        for (int l = 0; l < DIM * NB; ++l)
            abar[VOFF + l] = kbar[l];
    }

    /// RK4 step (formulation.hpp:21-28):
    ///   state + (Ts/6) (k1 + 2 k2 + 2 k3 + k4),  k2 = f(state + Ts k1 / 2), ...
    template <class T>
    static void rk4(const T *s, const T *u, T *xn) {
        T k1[nx], k2[nx], k3[nx], k4[nx], w[nx];
        Springs<T> sp;
        fc(s, u, k1, sp);
        for (int l = 0; l < nx; ++l)
            w[l] = s[l] + 0.5 * (Ts * k1[l]);
        fc(w, u, k2, sp);
        for (int l = 0; l < nx; ++l)
            w[l] = s[l] + 0.5 * (Ts * k2[l]);
        fc(w, u, k3, sp);
        for (int l = 0; l < nx; ++l)
            w[l] = s[l] + Ts * k3[l];
        fc(w, u, k4, sp);
        for (int l = 0; l < nx; ++l)
            xn[l] = s[l] + c6 * (((k1[l] + 2.0 * k2[l]) + 2.0 * k3[l]) + k4[l]);
    }

    /// Adjoint of one RK4 step: given mu = adjoint of x_next, returns sbar =
    /// adjoint of s and ubar[3] = adjoint of u (dynamics part only).
    template <class T>
    static void rk4_adj(const T *s, const T *u, const T *mu, T *sbar, T *ubar) {
        T k[nx], w[nx];
        Springs<T> sp1, sp2, sp3, sp4;
        fc(s, u, k, sp1);
        for (int l = 0; l < nx; ++l)
            w[l] = s[l] + 0.5 * (Ts * k[l]);
        fc(w, u, k, sp2);
        for (int l = 0; l < nx; ++l)
            w[l] = s[l] + 0.5 * (Ts * k[l]);
        fc(w, u, k, sp3);
        for (int l = 0; l < nx; ++l)
            w[l] = s[l] + Ts * k[l];
        springs(w, sp4);

        T kb1[nx], kb2[nx], kb3[nx], kb4[nx], a[nx];
        for (int l = 0; l < nx; ++l) {
            T sg   = c6 * mu[l];
            kb1[l] = sg;
            kb2[l] = 2.0 * sg;
            kb3[l] = 2.0 * sg;
            kb4[l] = sg;
            sbar[l] = mu[l];
        }
        fc_adj(sp4, kb4, a);
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb3[l]  = kb3[l] + Ts * a[l];
        }
        fc_adj(sp3, kb3, a);
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb2[l]  = kb2[l] + 0.5 * (Ts * a[l]);
        }
        fc_adj(sp2, kb2, a);
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb1[l]  = kb1[l] + 0.5 * (Ts * a[l]);
        }
        fc_adj(sp1, kb1, a);
        for (int l = 0; l < nx; ++l)
            sbar[l] = sbar[l] + a[l];
        for (int c = 0; c < DIM; ++c) {
            int l   = DIM * NB + c;
            ubar[c] = ((kb4[l] + kb3[l]) + kb2[l]) + kb1[l];
        }
    }

    // ---- tape: second-order information of one RK4 substage --------------------------------
    // The Hessian-vector product the solver uses.  One gradient-like sweep at the current point
    // records, per substage and per spring, the 3x3 matrices
    //   K = dF/dd = c I + w d d^T                    (w = DL / r^3)
    //   M = d(K lF)/dd = w [ lF d^T + d lF^T + (d.lF) (I - (3/n2) d d^T) ]
    // (lF = adjoint force of the gradient sweep).  A product H v is then two linear sweeps over
    // the tape: tangent forward  F_t = K d_t,  second-order adjoint  ld_t = K lF_t + M d_t.
    // Association order and the explicit fma calls below are the specification the kernels
    // follow bit for bit; eval_hess_psi_prod_dual (dual numbers) is the independent check.
    static constexpr double hTs = 0.5 * Ts;
    struct TapeSub {
        double K[NP][DIM][DIM], M[NP][DIM][DIM];
    };
    struct TapeStage {
        TapeSub sub[4];
    };

    /// fc_adj on doubles that also records K and M of the substage.
    static void fc_adj_record(const Springs<double> &sp, const double *kbar, double *abar, TapeSub &ts) {
        double ld[NP][DIM];
        for (int p = 0; p < NP; ++p) {
            double lF[DIM], pr[DIM];
            for (int c = 0; c < DIM; ++c) {
                if (p == 0)
                    lF[c] = inv_m * (0.0 - kbar[VOFF + DIM * p + c]);
                else if (p == NP - 1)
                    lF[c] = inv_m * (kbar[VOFF + DIM * (p - 1) + c] - 0.0);
                else
                    lF[c] = inv_m * (kbar[VOFF + DIM * (p - 1) + c] - kbar[VOFF + DIM * p + c]);
                pr[c] = sp.d[p][c] * lF[c];
            }
            double dt  = (pr[0] + pr[1]) + pr[2];
            double n2r = sp.n2[p] * sp.r[p];
            double w   = DL / n2r;
            double t   = dt * w;
            for (int c = 0; c < DIM; ++c)
                ld[p][c] = (sp.c[p] * lF[c]) + (t * sp.d[p][c]);
            double e = 3.0 / sp.n2[p];
            for (int c = 0; c < DIM; ++c) {
                double wd = w * sp.d[p][c];
                double g  = e * sp.d[p][c];
                for (int c2 = 0; c2 < DIM; ++c2) {
                    ts.K[p][c][c2] = std::fma(wd, sp.d[p][c2], c2 == c ? sp.c[p] : 0.0);
                    double m       = lF[c] * sp.d[p][c2];
                    m              = std::fma(sp.d[p][c], lF[c2], m);
                    double kd      = std::fma(-g, sp.d[p][c2], c2 == c ? 1.0 : 0.0);
                    m              = std::fma(dt, kd, m);
                    ts.M[p][c][c2] = w * m;
                }
            }
            // K and M are symmetric in exact arithmetic; the tape stores their upper triangles
            for (int c = 0; c < DIM; ++c)
                for (int c2 = 0; c2 < c; ++c2) {
                    ts.K[p][c][c2] = ts.K[p][c2][c];
                    ts.M[p][c][c2] = ts.M[p][c2][c];
                }
        }
        for (int p = 0; p < NP; ++p)
            for (int c = 0; c < DIM; ++c)
                abar[DIM * p + c] = p == NP - 1 ? ld[p][c] : ld[p][c] - ld[p + 1][c];
        for (int l = 0; l < DIM * NB; ++l)
            abar[VOFF + l] = kbar[l];
    }

    /// rk4_adj on doubles, recording the stage's tape (same arithmetic as rk4_adj<double>).
    static void rk4_adj_record(const double *s, const double *u, const double *mu, double *sbar,
                               double *ubar, TapeStage &tp) {
        double k[nx], w[nx];
        Springs<double> sp1, sp2, sp3, sp4;
        fc(s, u, k, sp1);
        for (int l = 0; l < nx; ++l)
            w[l] = s[l] + 0.5 * (Ts * k[l]);
        fc(w, u, k, sp2);
        for (int l = 0; l < nx; ++l)
            w[l] = s[l] + 0.5 * (Ts * k[l]);
        fc(w, u, k, sp3);
        for (int l = 0; l < nx; ++l)
            w[l] = s[l] + Ts * k[l];
        springs(w, sp4);
        double kb1[nx], kb2[nx], kb3[nx], kb4[nx], a[nx];
        for (int l = 0; l < nx; ++l) {
            double sg = c6 * mu[l];
            kb1[l] = sg;
            kb2[l] = 2.0 * sg;
            kb3[l] = 2.0 * sg;
            kb4[l] = sg;
            sbar[l] = mu[l];
        }
        fc_adj_record(sp4, kb4, a, tp.sub[3]);
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb3[l]  = kb3[l] + Ts * a[l];
        }
        fc_adj_record(sp3, kb3, a, tp.sub[2]);
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb2[l]  = kb2[l] + 0.5 * (Ts * a[l]);
        }
        fc_adj_record(sp2, kb2, a, tp.sub[1]);
        for (int l = 0; l < nx; ++l) {
            sbar[l] = sbar[l] + a[l];
            kb1[l]  = kb1[l] + 0.5 * (Ts * a[l]);
        }
        fc_adj_record(sp1, kb1, a, tp.sub[0]);
        for (int l = 0; l < nx; ++l)
            sbar[l] = sbar[l] + a[l];
        for (int c = 0; c < DIM; ++c) {
            int l   = DIM * NB + c;
            ubar[c] = ((kb4[l] + kb3[l]) + kb2[l]) + kb1[l];
        }
    }

    /// Tangent of f_c over the tape: d_t (kept for the adjoint sweep) and ds_t.
    static void fc_tan(const TapeSub &g, const double *s_t, const double *u_t, double *ds_t,
                       double (&d_t)[NP][DIM]) {
        double F[NP][DIM];
        for (int p = 0; p < NP; ++p) {
            for (int c = 0; c < DIM; ++c)
                d_t[p][c] = s_t[DIM * p + c] - (p == 0 ? 0.0 : s_t[DIM * (p - 1) + c]);
            for (int c = 0; c < DIM; ++c) {
                double f = g.K[p][c][0] * d_t[p][0];
                f        = std::fma(g.K[p][c][1], d_t[p][1], f);
                f        = std::fma(g.K[p][c][2], d_t[p][2], f);
                F[p][c]  = f;
            }
        }
        for (int l = 0; l < DIM * NB; ++l)
            ds_t[l] = s_t[VOFF + l];
        for (int c = 0; c < DIM; ++c)
            ds_t[DIM * NB + c] = u_t[c];
        for (int i = 0; i < NB; ++i)
            for (int c = 0; c < DIM; ++c)
                ds_t[VOFF + DIM * i + c] = inv_m * (F[i + 1][c] - F[i][c]);
    }
    /// Tangent of the f_c adjoint over the tape.
    static void fc_adj_tan(const TapeSub &g, const double (&d_t)[NP][DIM], const double *kbar_t, double *abar_t) {
        double ld[NP][DIM];
        for (int p = 0; p < NP; ++p) {
            double lF[DIM];
            for (int c = 0; c < DIM; ++c) {
                double A = p == 0 ? 0.0 : kbar_t[VOFF + DIM * (p - 1) + c];
                double B = p == NP - 1 ? 0.0 : kbar_t[VOFF + DIM * p + c];
                lF[c]    = inv_m * (A - B);
            }
            for (int c = 0; c < DIM; ++c) {
                double v = g.K[p][c][0] * lF[0];
                v        = std::fma(g.K[p][c][1], lF[1], v);
                v        = std::fma(g.K[p][c][2], lF[2], v);
                v        = std::fma(g.M[p][c][0], d_t[p][0], v);
                v        = std::fma(g.M[p][c][1], d_t[p][1], v);
                v        = std::fma(g.M[p][c][2], d_t[p][2], v);
                ld[p][c] = v;
            }
        }
        for (int p = 0; p < NP; ++p)
            for (int c = 0; c < DIM; ++c)
                abar_t[DIM * p + c] = p == NP - 1 ? ld[p][c] : ld[p][c] - ld[p + 1][c];
        for (int l = 0; l < DIM * NB; ++l)
            abar_t[VOFF + l] = kbar_t[l];
    }
    struct StageTan {
        double d_t[4][NP][DIM];
    };
    static void rk4_tan(const TapeStage &tp, const double *s_t, const double *u_t, double *xn_t, StageTan &st) {
        double k1[nx], k2[nx], k3[nx], k4[nx], w[nx];
        fc_tan(tp.sub[0], s_t, u_t, k1, st.d_t[0]);
        for (int l = 0; l < nx; ++l)
            w[l] = std::fma(hTs, k1[l], s_t[l]);
        fc_tan(tp.sub[1], w, u_t, k2, st.d_t[1]);
        for (int l = 0; l < nx; ++l)
            w[l] = std::fma(hTs, k2[l], s_t[l]);
        fc_tan(tp.sub[2], w, u_t, k3, st.d_t[2]);
        for (int l = 0; l < nx; ++l)
            w[l] = std::fma(Ts, k3[l], s_t[l]);
        fc_tan(tp.sub[3], w, u_t, k4, st.d_t[3]);
        for (int l = 0; l < nx; ++l) {
            double t = std::fma(2.0, k2[l], k1[l]);
            t        = std::fma(2.0, k3[l], t);
            t        = t + k4[l];
            xn_t[l]  = std::fma(c6, t, s_t[l]);
        }
    }
    static void rk4_adj_tan(const TapeStage &tp, const StageTan &st, const double *mu_t, double *sbar_t,
                            double *ubar_t) {
        double kb1[nx], kb2[nx], kb3[nx], kb4[nx], a[nx];
        for (int l = 0; l < nx; ++l) {
            double sg = c6 * mu_t[l];
            kb1[l] = sg;
            kb2[l] = 2.0 * sg;
            kb3[l] = 2.0 * sg;
            kb4[l] = sg;
            sbar_t[l] = mu_t[l];
        }
        fc_adj_tan(tp.sub[3], st.d_t[3], kb4, a);
        for (int l = 0; l < nx; ++l) {
            sbar_t[l] = sbar_t[l] + a[l];
            kb3[l]    = std::fma(Ts, a[l], kb3[l]);
        }
        fc_adj_tan(tp.sub[2], st.d_t[2], kb3, a);
        for (int l = 0; l < nx; ++l) {
            sbar_t[l] = sbar_t[l] + a[l];
            kb2[l]    = std::fma(hTs, a[l], kb2[l]);
        }
        fc_adj_tan(tp.sub[1], st.d_t[1], kb2, a);
        for (int l = 0; l < nx; ++l) {
            sbar_t[l] = sbar_t[l] + a[l];
            kb1[l]    = std::fma(hTs, a[l], kb1[l]);
        }
        fc_adj_tan(tp.sub[0], st.d_t[0], kb1, a);
        for (int l = 0; l < nx; ++l)
            sbar_t[l] = sbar_t[l] + a[l];
        for (int c = 0; c < DIM; ++c) {
            int l     = DIM * NB + c;
            ubar_t[c] = ((kb4[l] + kb3[l]) + kb2[l]) + kb1[l];
        }
    }
    static void cost_grad_x_add_tan(const double *s_t, double *lam_t) {
        for (int c = 0; c < DIM; ++c) {
            int l    = DIM * NB + c;
            lam_t[l] = std::fma(c2a, s_t[l], lam_t[l]);
        }
        for (int l = VOFF; l < nx; ++l)
            lam_t[l] = std::fma(c2b, s_t[l], lam_t[l]);
    }

    /// Per-component cost contribution of one stage (hanging_chain.cpp:127-131): components
    /// 27..29 (end point): alpha (x_end - ref)^2 then + gamma u^2; components 30..56: beta v^2.
    template <class T>
    static void cost_accumulate(const T *s, const T *u /*nullptr for l_N*/, T *acc /*[64]*/) {
        for (int c = 0; c < DIM; ++c) {
            int l  = DIM * NB + c;
            T t    = s[l] - x_end(c);
            acc[l] = acc[l] + alpha * (t * t);
            if (u)
                acc[l] = acc[l] + gamma * (u[c] * u[c]);
        }
        for (int l = VOFF; l < nx; ++l)
            acc[l] = acc[l] + beta * (s[l] * s[l]);
    }

    /// Total of the per-component running sums: per chain point
    /// ((((x0 + x1) + x2) + v0) + v1) + v2, then the 64-lane butterfly with point p at lane 4p
    /// (the kernels hold point p's components on lanes 4p..4p+2).
    static double cost_total(const double *acc /*[64], indexed like the state vector*/) {
        double part[64] = {};
        for (int p = 0; p < NP; ++p) {
            double ax[DIM], av[DIM];
            for (int c = 0; c < DIM; ++c) {
                ax[c] = acc[DIM * p + c];
                av[c] = p < NB ? acc[VOFF + DIM * p + c] : 0.0;
            }
            part[4 * p] = ((((ax[0] + ax[1]) + ax[2]) + av[0]) + av[1]) + av[2]; // lane of the point's component 0
        }
        return wave_sum(part);
    }

    /// Gradient of the stage/terminal cost w.r.t. the state, added to lam.
    template <class T>
    static void cost_grad_x_add(const T *s, T *lam) {
        for (int c = 0; c < DIM; ++c) {
            int l  = DIM * NB + c;
            lam[l] = lam[l] + c2a * (s[l] - x_end(c));
        }
        for (int l = VOFF; l < nx; ++l)
            lam[l] = lam[l] + c2b * s[l];
    }
};

// ---------------------------------------------------------------------------
// Single-shooting problem functions (formulation.hpp:45-97 with m = 0)
// ---------------------------------------------------------------------------
struct EvalCounters {
    long psi = 0, grad_psi = 0, psi_grad_psi = 0, grad_L = 0, hess_psi_prod = 0,
         prox_grad_step = 0;
};

template <class Model>
struct SingleShooting {
    int N;
    std::vector<double> x0; ///< problem parameter (initial state), mpc-driver.cpp:126
    mutable EvalCounters cnt;
    mutable std::vector<typename Model::TapeStage> tape_; ///< second-order tape ...
    mutable std::vector<double> tape_u_;                  ///< ... and the point it was recorded at
    static constexpr int nx = Model::nx, nu = Model::nu;

    explicit SingleShooting(int N_) : N(N_), x0(nx) { Model::initial_state(x0.data()); }
    int n() const { return nu * N; }
    int m() const { return 0; }

    /// f(u; x0) = sum_k l(x_k, u_k) + l_N(x_N)   (formulation.hpp:57-60)
    double eval_f_impl(const double *u, double *X /*optional (N+1)*nx*/) const {
        double s[nx], xn[nx], acc[64] = {};
        std::copy(x0.begin(), x0.end(), s);
        for (int k = 0; k < N; ++k) {
            if (X)
                std::copy(s, s + nx, X + k * nx);
            Model::cost_accumulate(s, u + nu * k, acc);
            Model::rk4(s, u + nu * k, xn);
            std::copy(xn, xn + nx, s);
        }
        if (X)
            std::copy(s, s + nx, X + N * nx);
        Model::template cost_accumulate<double>(s, nullptr, acc);
        return Model::cost_total(acc);
    }

    double eval_psi(const double *u, double * /*yhat: m = 0*/ = nullptr) const {
        ++cnt.psi;
        return eval_f_impl(u, nullptr);
    }
    void set_al(const double *, const double *) {} ///< no general constraints

    /// psi and its gradient by a forward sweep + reverse (adjoint) sweep.  With record = true
    /// the adjoint sweep also records the second-order tape at u (same gradient arithmetic).
    double psi_grad_impl(const double *u, double *grad, bool record = false) const {
        std::vector<double> X((N + 1) * nx);
        double psi = eval_f_impl(u, X.data());
        if (record) {
            tape_.resize(N);
            tape_u_.assign(u, u + n());
        }
        double lam[nx] = {}, sbar[nx], ubar[nu];
        Model::cost_grad_x_add(X.data() + N * nx, lam);
        for (int k = N - 1; k >= 0; --k) {
            const double *s = X.data() + k * nx;
            if (record)
                Model::rk4_adj_record(s, u + nu * k, lam, sbar, ubar, tape_[k]);
            else
                Model::rk4_adj(s, u + nu * k, lam, sbar, ubar);
            for (int c = 0; c < nu; ++c)
                grad[nu * k + c] = ubar[c] + Model::c2g * u[nu * k + c];
            std::copy(sbar, sbar + nx, lam);
            Model::cost_grad_x_add(s, lam);
        }
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
    /// grad_L(x, y) with m = 0 is the gradient of f (formulation.hpp:98-99).
    /// (PANTR evaluates it at the point its Hessian-vector products follow, so this is where
    /// the second-order tape is recorded, as in the kernels.)
    void eval_grad_L(const double *u, double *grad, bool record = true) const {
        ++cnt.grad_L;
        psi_grad_impl(u, grad, record);
    }

    /// Hv = d/du (grad psi(u)^T v): forward-over-reverse with dual numbers
    /// (formulation.hpp:93-97 builds the same product symbolically).  Independent check of
    /// eval_hess_psi_prod below (agrees to rounding, not bit for bit).
    void eval_hess_psi_prod_dual(const double *u, const double *v, double *Hv) const {
        std::vector<Dual> X((N + 1) * nx);
        Dual s[nx], xn[nx], ud[nu];
        for (int l = 0; l < nx; ++l)
            s[l] = {x0[l], 0.0};
        for (int k = 0; k < N; ++k) {
            std::copy(s, s + nx, X.data() + k * nx);
            for (int c = 0; c < nu; ++c)
                ud[c] = {u[nu * k + c], v[nu * k + c]};
            Model::rk4(s, ud, xn);
            std::copy(xn, xn + nx, s);
        }
        std::copy(s, s + nx, X.data() + N * nx);
        Dual lam[nx], sbar[nx], ubar[nu];
        for (int l = 0; l < nx; ++l)
            lam[l] = {0.0, 0.0};
        Model::cost_grad_x_add(X.data() + N * nx, lam);
        for (int k = N - 1; k >= 0; --k) {
            const Dual *sk = X.data() + k * nx;
            for (int c = 0; c < nu; ++c)
                ud[c] = {u[nu * k + c], v[nu * k + c]};
            Model::rk4_adj(sk, ud, lam, sbar, ubar);
            for (int c = 0; c < nu; ++c)
                Hv[nu * k + c] = (ubar[c] + Model::c2g * ud[c]).t;
            std::copy(sbar, sbar + nx, lam);
            Model::cost_grad_x_add(sk, lam);
        }
    }

    /// H v over a recorded tape: tangent forward sweep, second-order adjoint sweep.
    void hess_prod_tape(const std::vector<typename Model::TapeStage> &tape, const double *v, double *Hv) const {
        std::vector<double> Xt((N + 1) * nx);
        std::vector<typename Model::StageTan> st(N);
        double s_t[nx] = {}, xn_t[nx];
        for (int k = 0; k < N; ++k) {
            std::copy(s_t, s_t + nx, Xt.data() + k * nx);
            Model::rk4_tan(tape[k], s_t, v + nu * k, xn_t, st[k]);
            std::copy(xn_t, xn_t + nx, s_t);
        }
        double lam_t[nx] = {}, sbar_t[nx], ubar_t[nu];
        Model::cost_grad_x_add_tan(s_t, lam_t);
        for (int k = N - 1; k >= 0; --k) {
            Model::rk4_adj_tan(tape[k], st[k], lam_t, sbar_t, ubar_t);
            for (int c = 0; c < nu; ++c)
                Hv[nu * k + c] = std::fma(Model::c2g, v[nu * k + c], ubar_t[c]);
            std::copy(sbar_t, sbar_t + nx, lam_t);
            Model::cost_grad_x_add_tan(Xt.data() + k * nx, lam_t);
        }
    }
    /// The Hessian-vector product of the problem (formulation.hpp:93-97, 163-168).
    void eval_hess_psi_prod(const double *u, const double *v, double *Hv) const {
        ++cnt.hess_psi_prod;
        if (tape_u_.size() != (size_t)n() || !std::equal(u, u + n(), tape_u_.begin())) {
            std::vector<double> g(n()); // tape is for another point: record it here
            psi_grad_impl(u, g.data(), /*record=*/true);
        }
        hess_prod_tape(tape_, v, Hv);
    }

    /// Box C = U^N (formulation.hpp:220-227).
    double lb(int i) const { return Model::u_lb(i % nu); }
    double ub(int i) const { return Model::u_ub(i % nu); }

    /// Proximal-gradient step for a box: p = min(max(-gamma grad, lb - x), ub - x),
    /// xhat = x + p; h(xhat) = 0.  (C signature: problems/problems/lasso.cpp:72-84;
    /// box semantics as used by results.hpp:268-271.)
    void eval_prox_grad_step(double gamma, const double *x, const double *grad, double *xh,
                             double *p) const {
        ++cnt.prox_grad_step;
        for (int i = 0; i < n(); ++i) {
            double t = (-gamma) * grad[i];
            p[i]     = std::fmin(std::fmax(t, lb(i) - x[i]), ub(i) - x[i]);
            xh[i]    = x[i] + p[i];
        }
    }

    /// J = { i : lb_i < x_i - gamma grad_i < ub_i }; returns |J|, mask[i] in {0,1}.
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

    /// Plant step x+ = f_d(x, u)  (formulation.hpp:258-261).
    static void simulate_dynamics(const double *x, const double *u, double *xn) {
        Model::rk4(x, u, xn);
    }
};

} // namespace oracle
