// Hanging-chain stage model for one 64-wide wavefront (gfx950), lane = chain point.
//
// What the reference computes (paths relative to /root/reference):
//   constants ............ problems/problems/hanging_chain.cpp:28-42
//   f_c (springs, gravity) problems/problems/hanging_chain.cpp:85-100
//   RK4 ................... problems/problems/formulation.hpp:21-28
//   l, l_N ................ problems/problems/hanging_chain.cpp:127-132
//   input box ............. problems/problems/hanging_chain.cpp:107-110
//   initial state ......... problems/problems/hanging_chain.cpp:121-124
// The reference turns these into straight-line scalar C through CasADi.  Here lane p (0..9)
// owns chain point p: its three position components and, for the nine free masses p < 9, its
// three velocity components (point 9 is the actuated end).  Consequences:
//   * the 3-term norms and dot products of a spring are in-lane (no cross-lane traffic),
//   * a neighbouring point is one lane away inside a 16-lane row: DPP row_shr:1 / row_shl:1
//     moves it at VALU speed (no LDS crossbar round trip),
//   * the three components give three independent fp64 chains per lane (ILP 3).
// Lanes 10..63 idle through the sweeps (they take part in the decision-vector algebra).
// The adjoint of f_c is hand-derived.  Second-order information: dual numbers on the same
// templates (validation path) and, in the solver, a recorded tape of the primal substage
// quantities replayed by division-free tangent sweeps (bit-identical to the dual path).
#pragma once
#include "wave_math.hip.h"

namespace pantr {

struct ChainModel {
    static constexpr int NB = 9, DIM = 3, NP = NB + 1;
    static constexpr int nx = DIM * NP + DIM * NB; // 57
    static constexpr int nu = DIM;                 // 3
    static constexpr int VOFF = DIM * NP;          // velocities start at x0[30]
    static constexpr double Ts = 0.1;
    static constexpr double alpha = 25, beta = 1, gamma = 0.01;
    static constexpr double mass = 0.03, Dspring = 0.1, Lspring = 0.033;
    static constexpr double v_max = 1, g_grav = 9.81;
    static constexpr double inv_m = 1.0 / mass;
    static constexpr double DL    = Dspring * Lspring;
    static constexpr double c6    = Ts / 6;
    static constexpr double c2a = 2 * alpha, c2b = 2 * beta, c2g = 2 * gamma;

    __host__ __device__ static double u_lb(int) { return -v_max; }
    __host__ __device__ static double u_ub(int) { return +v_max; }
    __host__ __device__ static double initial_state(int l) {
        return (l < VOFF && l % DIM == 0) ? static_cast<double>(l / DIM + 1) / static_cast<double>(NB + 1)
                                          : 0.0;
    }

    /// Per-lane role.
    struct Lane {
        int l;
        bool active, mass, end; ///< lane < 10, lane < 9 (has a velocity), lane == 9
        __device__ explicit Lane(int lane) : l(lane), active(lane < NP), mass(lane < NB), end(lane == NB) {}
    };

    /// State (or adjoint, or time derivative) of one chain point.
    template <class T>
    struct State {
        T x[DIM], v[DIM];
    };
    /// Stage input u_k, held by every lane (lane 9 uses it).
    template <class T>
    struct Input {
        T u[DIM];
    };
    template <class T>
    __device__ static __forceinline__ State<T> zero_state() {
        State<T> s;
#pragma unroll
        for (int c = 0; c < DIM; ++c)
            s.x[c] = s.v[c] = make_const<T>(0.0);
        return s;
    }

    // ---- memory <-> registers -------------------------------------------------------------------
    /// Initial state of this lane's point from the reference's state vector layout
    /// [vec(x) (30); vec(v) (27)]  (hanging_chain.cpp:70-73).
    __device__ static __forceinline__ State<double> load_x0(const Lane &L, const double *x0) {
        State<double> s = zero_state<double>();
        if (L.active)
#pragma unroll
            for (int c = 0; c < DIM; ++c)
                s.x[c] = x0[DIM * L.l + c];
        if (L.mass)
#pragma unroll
            for (int c = 0; c < DIM; ++c)
                s.v[c] = x0[VOFF + DIM * L.l + c];
        return s;
    }
    __device__ static __forceinline__ void store_x(const Lane &L, double *x, const State<double> &s) {
        if (L.active)
#pragma unroll
            for (int c = 0; c < DIM; ++c)
                x[DIM * L.l + c] = s.x[c];
        if (L.mass)
#pragma unroll
            for (int c = 0; c < DIM; ++c)
                x[VOFF + DIM * L.l + c] = s.v[c];
    }
    /// Trajectory rows in the HBM workspace: 6 components x 16 lanes of doubles per stage.
    static constexpr int TRAJ_LANES   = 16;
    static constexpr int TRAJ_DOUBLES = 2 * DIM * TRAJ_LANES; // 96 doubles = 768 B per stage
    __device__ static __forceinline__ void traj_store(const Lane &L, double *row, const State<double> &s) {
        if (L.l < TRAJ_LANES) {
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                row[c * TRAJ_LANES + L.l]         = s.x[c];
                row[(DIM + c) * TRAJ_LANES + L.l] = s.v[c];
            }
        }
    }
    __device__ static __forceinline__ State<double> traj_load(const Lane &L, const double *row) {
        State<double> s = zero_state<double>();
        if (L.l < TRAJ_LANES) {
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                s.x[c] = row[c * TRAJ_LANES + L.l];
                s.v[c] = row[(DIM + c) * TRAJ_LANES + L.l];
            }
        }
        return s;
    }
    /// u_k from a decision vector in LDS (same address in every lane: a broadcast read).
    __device__ static __forceinline__ Input<double> load_input(const double *u_k) {
        return {{u_k[0], u_k[1], u_k[2]}};
    }
    __device__ static __forceinline__ Input<dual> load_input(const double *u_k, const double *v_k) {
        return {{{u_k[0], v_k[0]}, {u_k[1], v_k[1]}, {u_k[2], v_k[2]}}};
    }
    /// Lane 9 owns the adjoint of the stage input.
    __device__ static __forceinline__ void store_input_grad(const Lane &L, double *g_k, const double (&g)[DIM]) {
        if (L.end)
#pragma unroll
            for (int c = 0; c < DIM; ++c)
                g_k[c] = g[c];
    }

    // ---- springs ---------------------------------------------------------------------------------
    template <class T>
    struct Springs {
        T d[DIM], n2, r, c;
    };

    /// dist_vect, its length and the spring coefficient of the spring ending at this point.
    template <class T>
    __device__ static __forceinline__ void springs(const State<T> &s, Springs<T> &sp) {
#pragma unroll
        for (int c = 0; c < DIM; ++c)
            sp.d[c] = s.x[c] - row_from_prev(s.x[c]); // point 0 is anchored at the origin: x - 0
        sp.n2 = (sp.d[0] * sp.d[0] + sp.d[1] * sp.d[1]) + sp.d[2] * sp.d[2];
        sp.r  = wsqrt(sp.n2);
        sp.c  = Dspring * (1.0 - Lspring / sp.r);
    }

    /// ds = f_c(s, u).
    template <class T>
    __device__ static __forceinline__ State<T> fc(const Lane &L, const State<T> &s, const Input<T> &in,
                                                  Springs<T> &sp) {
        const T zero = make_const<T>(0.0);
        springs(s, sp);
        State<T> ds;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            T F  = sp.d[c] * sp.c;
            T Fn = row_from_next(F); // force of the spring towards point p+1
            T a  = inv_m * (Fn - F);
            if (c == DIM - 1)
                a = a - g_grav;
            ds.x[c] = select(L.end, in.u[c], s.v[c]);
            ds.v[c] = select(L.mass, a, zero);
        }
        return ds;
    }

    /// Intermediates of the f_c adjoint that the tape records.
    template <class T>
    struct AdjAux {
        T lF[DIM], dt, n2r, w, t;
    };

    /// Adjoint of f_c: kbar = adjoint of ds, returns the adjoint of s; the adjoint of u is
    /// kbar.x on lane 9.
    template <class T>
    __device__ static __forceinline__ State<T> fc_adj(const Lane &L, const Springs<T> &sp,
                                                      const State<T> &kbar, AdjAux<T> &ax) {
        const T zero = make_const<T>(0.0);
        T pr[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            T A      = row_from_prev(kbar.v[c]);        // adjoint of a_{p-1} (0 for point 0)
            T B      = select(L.mass, kbar.v[c], zero); // adjoint of a_p (none for point 9)
            ax.lF[c] = inv_m * (A - B);
            pr[c]    = sp.d[c] * ax.lF[c];
        }
        ax.dt  = (pr[0] + pr[1]) + pr[2];
        ax.n2r = sp.n2 * sp.r;
        ax.w   = DL / ax.n2r;
        ax.t   = ax.dt * ax.w;
        State<T> ab;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            T ld    = (sp.c * ax.lF[c]) + (ax.t * sp.d[c]);
            T ldn   = row_from_next(ld);
            ab.x[c] = select(L.end, ld, ld - ldn);
            ab.v[c] = select(L.mass, kbar.x[c], zero);
        }
        return ab;
    }

    template <class T>
    __device__ static __forceinline__ State<T> axpy(double a, const State<T> &k, const State<T> &s) {
        State<T> w;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            w.x[c] = s.x[c] + a * k.x[c];
            w.v[c] = s.v[c] + a * k.v[c];
        }
        return w;
    }
    /// s + 0.5 (Ts k)   (formulation.hpp:24-25: state + Ts * k / 2)
    template <class T>
    __device__ static __forceinline__ State<T> half_step(const State<T> &s, const State<T> &k) {
        State<T> w;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            w.x[c] = s.x[c] + 0.5 * (Ts * k.x[c]);
            w.v[c] = s.v[c] + 0.5 * (Ts * k.v[c]);
        }
        return w;
    }

    /// One RK4 step of the plant (formulation.hpp:21-28).
    template <class T>
    __device__ static __forceinline__ State<T> rk4(const Lane &L, const State<T> &s, const Input<T> &in) {
        Springs<T> sp;
        State<T> k1 = fc(L, s, in, sp);
        State<T> k2 = fc(L, half_step(s, k1), in, sp);
        State<T> k3 = fc(L, half_step(s, k2), in, sp);
        State<T> k4 = fc(L, axpy(Ts, k3, s), in, sp);
        State<T> xn;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            xn.x[c] = s.x[c] + c6 * (((k1.x[c] + 2.0 * k2.x[c]) + 2.0 * k3.x[c]) + k4.x[c]);
            xn.v[c] = s.v[c] + c6 * (((k1.v[c] + 2.0 * k2.v[c]) + 2.0 * k3.v[c]) + k4.v[c]);
        }
        return xn;
    }

    // ---- the tape: primal quantities of one RK4 substage, recorded once per PANTR iteration ----
    // Per substage, four groups of four doubles per lane (16 lanes; lanes 0..9 matter):
    //   G0 = {d0, d1, d2, c}     dist_vect, c = D (1 - L/r)
    //   G1 = {h, kq, n2, r}      h = 0.5/r, kq = (-(L/r))/r, squared length, length
    //   G2 = {lF0, lF1, lF2, dt} adjoint force, dt = d.lF
    //   G3 = {w, kw, t, -}       w = DL/(n2 r), kw = (-w)/(n2 r), t = dt w
    // HBM layout: tape[((k*4 + j)*4 + g)*16 + lane] as double4: 512 B per group row, 2 KiB per
    // substage, 8 KiB per stage.  G0,G1 of a substage are one contiguous KiB (= one LDS-DMA
    // instruction), G2,G3 the next.
    // The tangent spring quantities of a Hessian-vector product's forward sweep go to a second,
    // per-product tape ("st"): T0 = {d0_t, d1_t, d2_t, n2_t}, T1 = {r_t, c_t, -, -}: 1 KiB per
    // substage, 4 KiB per stage; its adjoint sweep reads them back instead of recomputing.
    static constexpr int TAPE_GROUPS       = 4;
    static constexpr int TAPE_LANES        = 16;
    static constexpr int TAPE_D4_PER_STAGE = 4 * TAPE_GROUPS * TAPE_LANES; // 256 double4 = 8 KiB
    static constexpr int ST_GROUPS         = 2;
    static constexpr int ST_D4_PER_STAGE   = 4 * ST_GROUPS * TAPE_LANES;   // 128 double4 = 4 KiB
    static constexpr int TAPE_STAGE_BYTES  = TAPE_D4_PER_STAGE * 32;
    static constexpr int ST_STAGE_BYTES    = ST_D4_PER_STAGE * 32;
    static constexpr int LDS_STAGE_BYTES   = TAPE_STAGE_BYTES + ST_STAGE_BYTES; // one landing buffer

    struct TapeFwd {
        double d[DIM], c, h, kq;
    };
    struct TapeRev {
        double d[DIM], c, n2, r, lF[DIM], dt, w, kw, t;
    };
    struct SpringsTan {
        double d[DIM], n2, r, c;
    };

    __device__ static __forceinline__ void tape_store(const Lane &L, double4 *stage, int j,
                                                      const Springs<double> &sp, const AdjAux<double> &ax) {
        if (L.l < TAPE_LANES) {
            double q  = Lspring / sp.r; // the same quotient springs() forms
            double h  = 0.5 / sp.r;
            double kq = (-q) / sp.r;
            double kw = (-ax.w) / ax.n2r;
            double4 *p = stage + (j * TAPE_GROUPS) * TAPE_LANES + L.l;
            p[0]              = make_double4(sp.d[0], sp.d[1], sp.d[2], sp.c);
            p[TAPE_LANES]     = make_double4(h, kq, sp.n2, sp.r);
            p[2 * TAPE_LANES] = make_double4(ax.lF[0], ax.lF[1], ax.lF[2], ax.dt);
            p[3 * TAPE_LANES] = make_double4(ax.w, kw, ax.t, 0.0);
        }
    }
    __device__ static __forceinline__ void st_store(const Lane &L, double4 *stage, int j, const SpringsTan &st) {
        if (L.l < TAPE_LANES) {
            double4 *p    = stage + (j * ST_GROUPS) * TAPE_LANES + L.l;
            p[0]          = make_double4(st.d[0], st.d[1], st.d[2], st.n2);
            p[TAPE_LANES] = make_double4(st.r, st.c, 0.0, 0.0);
        }
    }

    /// Start the LDS-DMA of one stage into the landing buffer `buf`: the forward sweep needs the
    /// G0,G1 KiB of each substage (4 instructions); the adjoint sweep all 8 KiB of the tape plus
    /// the 4 KiB of the product's own tangent tape (12 instructions).
    __device__ static __forceinline__ void prefetch_fwd(const Lane &L, const double4 *tape_stage, char *buf) {
        const char *src = reinterpret_cast<const char *>(tape_stage) + L.l * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            lds_dma_row(src + j * 2048, buf + j * 2048);
    }
    __device__ static __forceinline__ void prefetch_rev(const Lane &L, const double4 *tape_stage,
                                                        const double4 *st_stage, char *buf) {
        const char *src = reinterpret_cast<const char *>(tape_stage) + L.l * 16;
#pragma unroll
        for (int i = 0; i < TAPE_STAGE_BYTES / 1024; ++i)
            lds_dma_row(src + i * 1024, buf + i * 1024);
        const char *src2 = reinterpret_cast<const char *>(st_stage) + L.l * 16;
#pragma unroll
        for (int i = 0; i < ST_STAGE_BYTES / 1024; ++i)
            lds_dma_row(src2 + i * 1024, buf + TAPE_STAGE_BYTES + i * 1024);
    }
    __device__ static __forceinline__ TapeFwd tape_fwd(const Lane &L, const char *buf, int j) {
        const double4 *p = reinterpret_cast<const double4 *>(buf) + (j * TAPE_GROUPS) * TAPE_LANES + (L.l & 15);
        double4 a = p[0], b = p[TAPE_LANES];
        return {{a.x, a.y, a.z}, a.w, b.x, b.y};
    }
    __device__ static __forceinline__ TapeRev tape_rev(const Lane &L, const char *buf, int j) {
        const double4 *p = reinterpret_cast<const double4 *>(buf) + (j * TAPE_GROUPS) * TAPE_LANES + (L.l & 15);
        double4 a = p[0], b = p[TAPE_LANES], c = p[2 * TAPE_LANES], d = p[3 * TAPE_LANES];
        return {{a.x, a.y, a.z}, a.w, b.z, b.w, {c.x, c.y, c.z}, c.w, d.x, d.y, d.z};
    }
    __device__ static __forceinline__ SpringsTan st_rev(const Lane &L, const char *buf, int j) {
        const double4 *p = reinterpret_cast<const double4 *>(buf + TAPE_STAGE_BYTES) +
                           (j * ST_GROUPS) * TAPE_LANES + (L.l & 15);
        double4 a = p[0], b = p[TAPE_LANES];
        return {{a.x, a.y, a.z}, a.w, b.x, b.y};
    }

    /// Tangent of springs(): the .t parts of the dual evaluation, primal values from the tape.
    __device__ static __forceinline__ void springs_tan(const TapeFwd &g, const State<double> &s_t, SpringsTan &st) {
        double sq[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            st.d[c] = s_t.x[c] - row_from_prev(s_t.x[c]);
            sq[c]   = (g.d[c] * st.d[c]) + (st.d[c] * g.d[c]);
        }
        st.n2       = (sq[0] + sq[1]) + sq[2];
        st.r        = st.n2 * g.h;
        double q_t  = st.r * g.kq;
        double om_t = 0.0 - q_t;
        st.c        = Dspring * om_t;
    }

    /// Tangent of f_c.
    __device__ static __forceinline__ State<double> fc_tan(const Lane &L, const TapeFwd &g, const State<double> &s_t,
                                                           const Input<double> &in_t, SpringsTan &st) {
        springs_tan(g, s_t, st);
        State<double> ds;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            double F  = (g.d[c] * st.c) + (st.d[c] * g.c);
            double Fn = row_from_next(F);
            double a  = inv_m * (Fn - F);
            ds.x[c]   = L.end ? in_t.u[c] : s_t.v[c];
            ds.v[c]   = L.mass ? a : 0.0;
        }
        return ds;
    }

    /// Tangent of the f_c adjoint.
    __device__ static __forceinline__ State<double> fc_adj_tan(const Lane &L, const TapeRev &g, const SpringsTan &st,
                                                               const State<double> &kbar_t) {
        double lF[DIM], pr[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            double A = row_from_prev(kbar_t.v[c]);
            double B = L.mass ? kbar_t.v[c] : 0.0;
            lF[c]    = inv_m * (A - B);
            pr[c]    = (g.d[c] * lF[c]) + (st.d[c] * g.lF[c]);
        }
        double dt  = (pr[0] + pr[1]) + pr[2];
        double n2r = (g.n2 * st.r) + (st.n2 * g.r);
        double w   = n2r * g.kw;
        double t   = (g.dt * w) + (dt * g.w);
        State<double> ab;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            double ld  = ((g.c * lF[c]) + (st.c * g.lF[c])) + ((g.t * st.d[c]) + (t * g.d[c]));
            double ldn = row_from_next(ld);
            ab.x[c]    = L.end ? ld : ld - ldn;
            ab.v[c]    = L.mass ? kbar_t.x[c] : 0.0;
        }
        return ab;
    }

    /// Tangent of one RK4 step; the stage's tape rows are in LDS at `buf`; the tangent spring
    /// quantities of the four substages are written to `st_stage` for the adjoint sweep.
    __device__ static __forceinline__ State<double> rk4_tan(const Lane &L, const char *buf, const State<double> &s_t,
                                                            const Input<double> &in_t, double4 *st_stage) {
        SpringsTan st;
        State<double> k1 = fc_tan(L, tape_fwd(L, buf, 0), s_t, in_t, st);
        st_store(L, st_stage, 0, st);
        State<double> k2 = fc_tan(L, tape_fwd(L, buf, 1), half_step(s_t, k1), in_t, st);
        st_store(L, st_stage, 1, st);
        State<double> k3 = fc_tan(L, tape_fwd(L, buf, 2), half_step(s_t, k2), in_t, st);
        st_store(L, st_stage, 2, st);
        State<double> k4 = fc_tan(L, tape_fwd(L, buf, 3), axpy(Ts, k3, s_t), in_t, st);
        st_store(L, st_stage, 3, st);
        State<double> xn;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            xn.x[c] = s_t.x[c] + c6 * (((k1.x[c] + 2.0 * k2.x[c]) + 2.0 * k3.x[c]) + k4.x[c]);
            xn.v[c] = s_t.v[c] + c6 * (((k1.v[c] + 2.0 * k2.v[c]) + 2.0 * k3.v[c]) + k4.v[c]);
        }
        return xn;
    }

    template <class T>
    __device__ static __forceinline__ void adj_chain_init(const State<T> &mu, State<T> &kb1, State<T> &kb2,
                                                          State<T> &kb3, State<T> &kb4, State<T> &sbar) {
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            T sx = c6 * mu.x[c], sv = c6 * mu.v[c];
            kb1.x[c] = sx; kb2.x[c] = 2.0 * sx; kb3.x[c] = 2.0 * sx; kb4.x[c] = sx;
            kb1.v[c] = sv; kb2.v[c] = 2.0 * sv; kb3.v[c] = 2.0 * sv; kb4.v[c] = sv;
            sbar.x[c] = mu.x[c];
            sbar.v[c] = mu.v[c];
        }
    }
    template <class T>
    __device__ static __forceinline__ void add_to(State<T> &s, const State<T> &a) {
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            s.x[c] = s.x[c] + a.x[c];
            s.v[c] = s.v[c] + a.v[c];
        }
    }
    template <class T>
    __device__ static __forceinline__ void add_scaled_full(State<T> &kb, const State<T> &a) { // kb + Ts a
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            kb.x[c] = kb.x[c] + Ts * a.x[c];
            kb.v[c] = kb.v[c] + Ts * a.v[c];
        }
    }
    template <class T>
    __device__ static __forceinline__ void add_scaled_half(State<T> &kb, const State<T> &a) { // kb + 0.5 (Ts a)
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            kb.x[c] = kb.x[c] + 0.5 * (Ts * a.x[c]);
            kb.v[c] = kb.v[c] + 0.5 * (Ts * a.v[c]);
        }
    }

    /// Tangent of the RK4 adjoint (second-order adjoint): tape and tangent springs from LDS.
    /// ubar_t (valid on lane 9) = tangent of the adjoint of u through the dynamics.
    __device__ static __forceinline__ void rk4_adj_tan(const Lane &L, const char *buf, const State<double> &mu_t,
                                                       State<double> &sbar_t, double (&ubar_t)[DIM]) {
        State<double> kb1, kb2, kb3, kb4;
        adj_chain_init(mu_t, kb1, kb2, kb3, kb4, sbar_t);
        State<double> a = fc_adj_tan(L, tape_rev(L, buf, 3), st_rev(L, buf, 3), kb4);
        add_to(sbar_t, a);
        add_scaled_full(kb3, a);
        a = fc_adj_tan(L, tape_rev(L, buf, 2), st_rev(L, buf, 2), kb3);
        add_to(sbar_t, a);
        add_scaled_half(kb2, a);
        a = fc_adj_tan(L, tape_rev(L, buf, 1), st_rev(L, buf, 1), kb2);
        add_to(sbar_t, a);
        add_scaled_half(kb1, a);
        a = fc_adj_tan(L, tape_rev(L, buf, 0), st_rev(L, buf, 0), kb1);
        add_to(sbar_t, a);
#pragma unroll
        for (int c = 0; c < DIM; ++c)
            ubar_t[c] = ((kb4.x[c] + kb3.x[c]) + kb2.x[c]) + kb1.x[c];
    }

    /// Adjoint of one RK4 step.  mu = adjoint of x_next; sbar = adjoint of s; ubar (valid on
    /// lane 9) = adjoint of u through the dynamics.  With tape_stage != nullptr (T = double)
    /// the substage quantities are recorded.
    template <class T>
    __device__ static __forceinline__ void rk4_adj(const Lane &L, const State<T> &s, const Input<T> &in,
                                                   const State<T> &mu, State<T> &sbar, T (&ubar)[DIM],
                                                   double4 *tape_stage = nullptr) {
        Springs<T> sp1, sp2, sp3, sp4;
        AdjAux<T> ax;
        State<T> k = fc(L, s, in, sp1);
        k          = fc(L, half_step(s, k), in, sp2);
        k          = fc(L, half_step(s, k), in, sp3);
        springs(axpy(Ts, k, s), sp4);

        State<T> kb1, kb2, kb3, kb4;
        adj_chain_init(mu, kb1, kb2, kb3, kb4, sbar);
        State<T> a = fc_adj(L, sp4, kb4, ax);
        record(L, tape_stage, 3, sp4, ax);
        add_to(sbar, a);
        add_scaled_full(kb3, a);
        a = fc_adj(L, sp3, kb3, ax);
        record(L, tape_stage, 2, sp3, ax);
        add_to(sbar, a);
        add_scaled_half(kb2, a);
        a = fc_adj(L, sp2, kb2, ax);
        record(L, tape_stage, 1, sp2, ax);
        add_to(sbar, a);
        add_scaled_half(kb1, a);
        a = fc_adj(L, sp1, kb1, ax);
        record(L, tape_stage, 0, sp1, ax);
        add_to(sbar, a);
#pragma unroll
        for (int c = 0; c < DIM; ++c)
            ubar[c] = ((kb4.x[c] + kb3.x[c]) + kb2.x[c]) + kb1.x[c];
    }
    __device__ static __forceinline__ void record(const Lane &L, double4 *stage, int j,
                                                  const Springs<double> &sp, const AdjAux<double> &ax) {
        if (stage)
            tape_store(L, stage, j, sp, ax);
    }
    __device__ static __forceinline__ void record(const Lane &, double4 *, int, const Springs<dual> &,
                                                  const AdjAux<dual> &) {}

    // ---- cost (hanging_chain.cpp:127-131) -----------------------------------------------------------
    /// Per-component running sums: lane 9 adds alpha (x_end - ref)^2 then gamma u^2 to acc.x,
    /// lanes 0..8 add beta v^2 to acc.v.
    template <class T>
    __device__ static __forceinline__ void cost_accumulate(const Lane &L, const State<T> &s, const Input<T> &in,
                                                           bool has_u, State<T> &acc) {
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            T t      = s.x[c] - (c == 0 ? 1.0 : 0.0); // reference position (1,0,0) of the end point
            T a1     = alpha * (t * t);
            T b1     = beta * (s.v[c] * s.v[c]);
            acc.x[c] = select(L.end, acc.x[c] + a1, acc.x[c]);
            acc.v[c] = select(L.mass, acc.v[c] + b1, acc.v[c]);
            if (has_u) {
                T a2     = gamma * (in.u[c] * in.u[c]);
                acc.x[c] = select(L.end, acc.x[c] + a2, acc.x[c]);
            }
        }
    }
    /// Total cost: per point ((((x0 + x1) + x2) + v0) + v1) + v2, then the 64-lane butterfly.
    __device__ static __forceinline__ double cost_total(const Lane &L, const State<double> &acc) {
        double p = ((((acc.x[0] + acc.x[1]) + acc.x[2]) + acc.v[0]) + acc.v[1]) + acc.v[2];
        return uniform(wave_sum(L.active ? p : 0.0));
    }
    /// lam + d(l or l_N)/dx.
    template <class T>
    __device__ static __forceinline__ void cost_grad_x_add(const Lane &L, const State<T> &s, State<T> &lam) {
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            T ge     = c2a * (s.x[c] - (c == 0 ? 1.0 : 0.0));
            T gv     = c2b * s.v[c];
            lam.x[c] = select(L.end, lam.x[c] + ge, lam.x[c]);
            lam.v[c] = select(L.mass, lam.v[c] + gv, lam.v[c]);
        }
    }
    /// Tangent of cost_grad_x_add: lam_t + d2(l)/dx2 s_t.
    __device__ static __forceinline__ void cost_grad_x_add_tan(const Lane &L, const State<double> &s_t,
                                                               State<double> &lam_t) {
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            lam_t.x[c] = L.end ? lam_t.x[c] + c2a * s_t.x[c] : lam_t.x[c];
            lam_t.v[c] = L.mass ? lam_t.v[c] + c2b * s_t.v[c] : lam_t.v[c];
        }
    }
};

} // namespace pantr
