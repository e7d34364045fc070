// Hanging-chain stage model, lane-mapped for one 64-wide wavefront (gfx950).
//
// What the reference computes (paths relative to /root/reference):
//   constants ............ problems/problems/hanging_chain.cpp:28-42
//   f_c (springs, gravity) problems/problems/hanging_chain.cpp:85-100
//   RK4 ................... problems/problems/formulation.hpp:21-28
//   l, l_N ................ problems/problems/hanging_chain.cpp:127-132
//   input box ............. problems/problems/hanging_chain.cpp:107-110
//   initial state ......... problems/problems/hanging_chain.cpp:121-124
// The reference turns these into straight-line scalar C through CasADi; here the 57 state
// components live one per lane (lanes 57..63 idle):
//   lane l in [0,30)  : position component c = l%3 of point p = l/3 (points 0..8 are the
//                       free masses, point 9 the actuated end)
//   lane l in [30,57) : velocity component c = l%3 of mass (l-30)/3
// The adjoint of f_c is hand-derived; second-order information comes from running the same
// templates on dual numbers (forward-over-reverse).
#pragma once
#include "wave_math.hip.h"

namespace pantr {

struct ChainModel {
    static constexpr int NB = 9, DIM = 3, NP = NB + 1;
    static constexpr int nx = DIM * NP + DIM * NB; // 57
    static constexpr int nu = DIM;                 // 3
    static constexpr int VOFF   = DIM * NP;        // 30
    static constexpr int U_LANE = DIM * NB;        // 27: lanes 27..29 carry x_end' = u
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

    /// Per-lane role, computed once per kernel.
    struct Lane {
        int l, p, tri, uidx;
        bool isX, isV, isE, first, last, grav;
        double xe;
        __device__ explicit Lane(int lane) {
            l     = lane;
            p     = lane / DIM;
            tri   = DIM * p;
            uidx  = lane % DIM;
            isX   = lane < VOFF;
            isV   = lane >= VOFF && lane < nx;
            isE   = lane >= U_LANE && lane < VOFF;
            first = lane < DIM;           // point 0: anchored to the origin
            last  = lane >= U_LANE;       // point 9 (only meaningful when isX)
            grav  = isV && uidx == DIM - 1;
            xe    = lane == U_LANE ? 1.0 : 0.0; // reference position (1,0,0) of the end point
        }
    };

    template <class T>
    struct Springs {
        T d, n2, r, c;
    };

    template <class T>
    __device__ static __forceinline__ void springs(const Lane &L, T s, Springs<T> &sp) {
        const T zero = make_const<T>(0.0);
        T sprev      = lane_read(s, L.l - DIM);
        sp.d         = select(L.isX, select(L.first, s, s - sprev), zero);
        T sq         = sp.d * sp.d;
        sp.n2        = (lane_read(sq, L.tri) + lane_read(sq, L.tri + 1)) + lane_read(sq, L.tri + 2);
        sp.r         = wsqrt(sp.n2);
        sp.c         = Dspring * (1.0 - Lspring / sp.r);
    }

    /// ds = f_c(s, u): per-lane time derivative.  uc = u[lane % 3] (used by lanes 27..29).
    template <class T>
    __device__ static __forceinline__ T fc(const Lane &L, T s, T uc, Springs<T> &sp) {
        const T zero = make_const<T>(0.0);
        springs(L, s, sp);
        T F   = sp.d * sp.c;
        T Fhi = lane_read(F, L.l - (VOFF - DIM)); // F of point i+1, same component
        T Flo = lane_read(F, L.l - VOFF);         // F of point i
        T a   = inv_m * (Fhi - Flo);
        a     = select(L.grav, a - g_grav, a);
        T vel = lane_read(s, L.l + VOFF);
        return select(L.l < U_LANE, vel, select(L.isE, uc, select(L.isV, a, zero)));
    }

    /// Intermediates of the f_c adjoint that the tape path records.
    template <class T>
    struct AdjAux {
        T lF, dt, n2r, w, t;
    };

    /// Adjoint of f_c: kbar = adjoint of ds, returns adjoint of s (adjoint of u = kbar on 27..29).
    template <class T>
    __device__ static __forceinline__ T fc_adj(const Lane &L, const Springs<T> &sp, T kbar, AdjAux<T> &ax) {
        const T zero = make_const<T>(0.0);
        T A   = select(L.isX && !L.first, lane_read(kbar, L.l + (VOFF - DIM)), zero);
        T B   = select(L.isX && !L.last, lane_read(kbar, L.l + VOFF), zero);
        ax.lF = inv_m * (A - B);
        T pr  = sp.d * ax.lF;
        ax.dt = (lane_read(pr, L.tri) + lane_read(pr, L.tri + 1)) + lane_read(pr, L.tri + 2);
        ax.n2r = sp.n2 * sp.r;
        ax.w   = DL / ax.n2r;
        ax.t   = ax.dt * ax.w;
        T ld   = (sp.c * ax.lF) + (ax.t * sp.d);
        T ldn  = lane_read(ld, L.l + DIM);
        T axx  = select(L.last, ld, ld - ldn);
        T av   = lane_read(kbar, L.l - VOFF);
        return select(L.isX, axx, select(L.isV, av, zero));
    }

    // ---- the tape: primal quantities of one RK4 substage, recorded once per PANTR iteration ----
    // Three groups of four doubles per lane (lanes 0..31; only the 30 position lanes matter):
    //   G0 = {d, c, h, kq}      d = dist_vect, c = D(1 - L/r), h = 0.5/r, kq = (-(L/r))/r
    //   G1 = {n2, r, lF, dt}    squared length, length, adjoint force, d.lF
    //   G2 = {w, kw, t, -}      w = DL/(n2 r), kw = (-w)/(n2 r), t = dt w
    // Row layout in HBM: tape[((k*4 + j)*3 + g)*32 + lane] as double4 (32 B per lane, 1 KiB per
    // group row, one coalesced access per wavefront).
    static constexpr int TAPE_GROUPS      = 3;
    static constexpr int TAPE_LANES       = 32;
    static constexpr int TAPE_D4_PER_STAGE = 4 * TAPE_GROUPS * TAPE_LANES; // double4 elements

    struct TapeFwd { // what the tangent forward sweep needs of one substage
        double d, c, h, kq;
    };
    struct TapeRev { // what the tangent adjoint sweep needs of one substage
        double d, c, h, kq, n2, r, lF, dt, w, kw, t;
    };
    struct SpringsTan {
        double d, n2, r, c; // tangents of the spring quantities
    };

    __device__ static __forceinline__ void tape_store(const Lane &L, double4 *stage, int j,
                                                      const Springs<double> &sp, const AdjAux<double> &ax) {
        if (L.l < TAPE_LANES) {
            double q  = Lspring / sp.r; // the same quotient springs() forms
            double h  = 0.5 / sp.r;
            double kq = (-q) / sp.r;
            double kw = (-ax.w) / ax.n2r;
            double4 *p = stage + (j * TAPE_GROUPS) * TAPE_LANES + L.l;
            p[0]              = make_double4(sp.d, sp.c, h, kq);
            p[TAPE_LANES]     = make_double4(sp.n2, sp.r, ax.lF, ax.dt);
            p[2 * TAPE_LANES] = make_double4(ax.w, kw, ax.t, 0.0);
        }
    }
    static constexpr int TAPE_STAGE_BYTES = TAPE_D4_PER_STAGE * 32; // 12 KiB
    static constexpr int TAPE_ROW_BYTES   = TAPE_LANES * 32;        // 1 KiB

    /// Start the LDS-DMA of one stage's tape rows into `buf` (12 KiB of LDS): all 12 rows for
    /// the adjoint sweep, only the four G0 rows for the forward sweep.  One instruction per row.
    template <bool ALL_GROUPS>
    __device__ static __forceinline__ void tape_prefetch(const Lane &L, const double4 *stage, char *buf) {
        const char *src = reinterpret_cast<const char *>(stage) + L.l * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < (ALL_GROUPS ? TAPE_GROUPS : 1); ++g) {
                int row = j * TAPE_GROUPS + g;
                lds_dma_row(src + row * TAPE_ROW_BYTES, buf + row * TAPE_ROW_BYTES);
            }
    }
    __device__ static __forceinline__ TapeFwd tape_fwd(const Lane &L, const char *buf, int j) {
        const double4 *p = reinterpret_cast<const double4 *>(buf) + (j * TAPE_GROUPS) * TAPE_LANES + (L.l & 31);
        double4 a = p[0];
        return {a.x, a.y, a.z, a.w};
    }
    __device__ static __forceinline__ TapeRev tape_rev(const Lane &L, const char *buf, int j) {
        const double4 *p = reinterpret_cast<const double4 *>(buf) + (j * TAPE_GROUPS) * TAPE_LANES + (L.l & 31);
        double4 a = p[0], b = p[TAPE_LANES], c = p[2 * TAPE_LANES];
        return {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z};
    }

    /// Tangent of springs(): the .t parts of the dual evaluation, primal values from the tape.
    template <class G>
    __device__ static __forceinline__ void springs_tan(const Lane &L, const G &g, double s_t, SpringsTan &st) {
        double sprev = lane_read(s_t, L.l - DIM);
        st.d         = L.first ? s_t : s_t - sprev;
        double sq    = (g.d * st.d) + (st.d * g.d);
        st.n2        = (lane_read(sq, L.tri) + lane_read(sq, L.tri + 1)) + lane_read(sq, L.tri + 2);
        st.r         = st.n2 * g.h;
        double q_t   = st.r * g.kq;
        double om_t  = 0.0 - q_t;
        st.c         = Dspring * om_t;
    }

    /// Tangent of f_c.
    template <class G>
    __device__ static __forceinline__ double fc_tan(const Lane &L, const G &g, double s_t, double uc_t,
                                                    SpringsTan &st) {
        springs_tan(L, g, s_t, st);
        double F   = (g.d * st.c) + (st.d * g.c);
        double Fhi = lane_read(F, L.l - (VOFF - DIM));
        double Flo = lane_read(F, L.l - VOFF);
        double a   = inv_m * (Fhi - Flo);
        double vel = lane_read(s_t, L.l + VOFF);
        return L.l < U_LANE ? vel : (L.isE ? uc_t : (L.isV ? a : 0.0));
    }

    /// Tangent of the f_c adjoint.
    __device__ static __forceinline__ double fc_adj_tan(const Lane &L, const TapeRev &g, const SpringsTan &st,
                                                        double kbar_t) {
        // cross-lane reads run with every lane active (a masked-off source lane reads as 0),
        // so they are hoisted out of the selects
        double Ar  = lane_read(kbar_t, L.l + (VOFF - DIM));
        double Br  = lane_read(kbar_t, L.l + VOFF);
        double A   = L.first ? 0.0 : Ar;
        double B   = L.last ? 0.0 : Br;
        double lF  = inv_m * (A - B);
        double pr  = (g.d * lF) + (st.d * g.lF);
        double dt  = (lane_read(pr, L.tri) + lane_read(pr, L.tri + 1)) + lane_read(pr, L.tri + 2);
        double n2r = (g.n2 * st.r) + (st.n2 * g.r);
        double w   = n2r * g.kw;
        double t   = (g.dt * w) + (dt * g.w);
        double ld  = ((g.c * lF) + (st.c * g.lF)) + ((g.t * st.d) + (t * g.d));
        double ldn = lane_read(ld, L.l + DIM);
        double axx = L.last ? ld : ld - ldn;
        double av  = lane_read(kbar_t, L.l - VOFF);
        return L.isX ? axx : (L.isV ? av : 0.0);
    }

    /// Tangent of one RK4 step; the stage's tape rows are in LDS at `buf`.
    __device__ static __forceinline__ double rk4_tan(const Lane &L, const char *buf, double s_t, double uc_t) {
        SpringsTan st;
        double k1 = fc_tan(L, tape_fwd(L, buf, 0), s_t, uc_t, st);
        double w  = s_t + 0.5 * (Ts * k1);
        double k2 = fc_tan(L, tape_fwd(L, buf, 1), w, uc_t, st);
        w         = s_t + 0.5 * (Ts * k2);
        double k3 = fc_tan(L, tape_fwd(L, buf, 2), w, uc_t, st);
        w         = s_t + Ts * k3;
        double k4 = fc_tan(L, tape_fwd(L, buf, 3), w, uc_t, st);
        return s_t + c6 * (((k1 + 2.0 * k2) + 2.0 * k3) + k4);
    }

    /// Tangent of the RK4 adjoint (second-order adjoint): mu_t = tangent of the incoming adjoint.
    __device__ static __forceinline__ void rk4_adj_tan(const Lane &L, const char *buf, double s_t,
                                                       double uc_t, double mu_t, double &sbar_t,
                                                       double &ubar_t) {
        SpringsTan st1, st2, st3, st4;
        double k = fc_tan(L, tape_fwd(L, buf, 0), s_t, uc_t, st1);
        double w = s_t + 0.5 * (Ts * k);
        k        = fc_tan(L, tape_fwd(L, buf, 1), w, uc_t, st2);
        w        = s_t + 0.5 * (Ts * k);
        k        = fc_tan(L, tape_fwd(L, buf, 2), w, uc_t, st3);
        w        = s_t + Ts * k;
        springs_tan(L, tape_fwd(L, buf, 3), w, st4);

        double sg  = c6 * mu_t;
        double kb1 = sg, kb2 = 2.0 * sg, kb3 = 2.0 * sg, kb4 = sg;
        sbar_t     = mu_t;
        double a   = fc_adj_tan(L, tape_rev(L, buf, 3), st4, kb4);
        sbar_t     = sbar_t + a;
        kb3        = kb3 + Ts * a;
        a          = fc_adj_tan(L, tape_rev(L, buf, 2), st3, kb3);
        sbar_t     = sbar_t + a;
        kb2        = kb2 + 0.5 * (Ts * a);
        a          = fc_adj_tan(L, tape_rev(L, buf, 1), st2, kb2);
        sbar_t     = sbar_t + a;
        kb1        = kb1 + 0.5 * (Ts * a);
        a          = fc_adj_tan(L, tape_rev(L, buf, 0), st1, kb1);
        sbar_t     = sbar_t + a;
        ubar_t     = ((kb4 + kb3) + kb2) + kb1;
    }

    /// Tangent of cost_grad_x_add: lam_t + d2(l)/dx2 s_t.
    __device__ static __forceinline__ double cost_grad_x_add_tan(const Lane &L, double s_t, double lam_t) {
        return L.isE ? lam_t + c2a * s_t : (L.isV ? lam_t + c2b * s_t : lam_t);
    }

    /// One RK4 step of the plant (formulation.hpp:21-28).
    template <class T>
    __device__ static __forceinline__ T rk4(const Lane &L, T s, T uc) {
        Springs<T> sp;
        T k1 = fc(L, s, uc, sp);
        T w  = s + 0.5 * (Ts * k1);
        T k2 = fc(L, w, uc, sp);
        w    = s + 0.5 * (Ts * k2);
        T k3 = fc(L, w, uc, sp);
        w    = s + Ts * k3;
        T k4 = fc(L, w, uc, sp);
        return s + c6 * (((k1 + 2.0 * k2) + 2.0 * k3) + k4);
    }

    /// Adjoint of one RK4 step.  mu = adjoint of x_next.  sbar = adjoint of s,
    /// ubar = adjoint of u through the dynamics (valid on lanes 27..29).
    /// With `tape_stage` != nullptr (T = double only) the substage quantities are recorded.
    template <class T>
    __device__ static __forceinline__ void rk4_adj(const Lane &L, T s, T uc, T mu, T &sbar, T &ubar,
                                                   double4 *tape_stage = nullptr) {
        Springs<T> sp1, sp2, sp3, sp4;
        AdjAux<T> ax;
        T k = fc(L, s, uc, sp1);
        T w = s + 0.5 * (Ts * k);
        k   = fc(L, w, uc, sp2);
        w   = s + 0.5 * (Ts * k);
        k   = fc(L, w, uc, sp3);
        w   = s + Ts * k;
        springs(L, w, sp4);

        T sg  = c6 * mu;
        T kb1 = sg, kb2 = 2.0 * sg, kb3 = 2.0 * sg, kb4 = sg;
        sbar  = mu;
        T a   = fc_adj(L, sp4, kb4, ax);
        record(L, tape_stage, 3, sp4, ax);
        sbar  = sbar + a;
        kb3   = kb3 + Ts * a;
        a     = fc_adj(L, sp3, kb3, ax);
        record(L, tape_stage, 2, sp3, ax);
        sbar  = sbar + a;
        kb2   = kb2 + 0.5 * (Ts * a);
        a     = fc_adj(L, sp2, kb2, ax);
        record(L, tape_stage, 1, sp2, ax);
        sbar  = sbar + a;
        kb1   = kb1 + 0.5 * (Ts * a);
        a     = fc_adj(L, sp1, kb1, ax);
        record(L, tape_stage, 0, sp1, ax);
        sbar  = sbar + a;
        ubar  = ((kb4 + kb3) + kb2) + kb1;
    }
    __device__ static __forceinline__ void record(const Lane &L, double4 *stage, int j,
                                                  const Springs<double> &sp, const AdjAux<double> &ax) {
        if (stage)
            tape_store(L, stage, j, sp, ax);
    }
    __device__ static __forceinline__ void record(const Lane &, double4 *, int, const Springs<dual> &,
                                                  const AdjAux<dual> &) {}

    /// Per-lane running cost: lanes 27..29 add alpha (x_end - ref)^2 then gamma u^2,
    /// lanes 30..56 add beta v^2  (hanging_chain.cpp:127-131).
    template <class T>
    __device__ static __forceinline__ T cost_accumulate(const Lane &L, T s, T uc, bool has_u, T acc) {
        T t  = s - L.xe;
        T a1 = alpha * (t * t);
        T b1 = beta * (s * s);
        acc  = select(L.isE, acc + a1, select(L.isV, acc + b1, acc));
        if (has_u) {
            T a2 = gamma * (uc * uc);
            acc  = select(L.isE, acc + a2, acc);
        }
        return acc;
    }

    /// lam + d(l or l_N)/dx, per lane.
    template <class T>
    __device__ static __forceinline__ T cost_grad_x_add(const Lane &L, T s, T lam) {
        T ge = c2a * (s - L.xe);
        T gv = c2b * s;
        return select(L.isE, lam + ge, select(L.isV, lam + gv, lam));
    }
};

} // namespace pantr
