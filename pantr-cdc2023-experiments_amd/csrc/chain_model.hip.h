// Hanging-chain stage model for one 64-wide wavefront (gfx950), lane = (substage chain, component, chain point).
//
// What the reference computes (paths relative to /root/reference):
//   constants ............ problems/problems/hanging_chain.cpp:28-42
//   f_c (springs, gravity) problems/problems/hanging_chain.cpp:85-100
//   RK4 ................... problems/problems/formulation.hpp:21-28
//   l, l_N ................ problems/problems/hanging_chain.cpp:127-132
//   input box ............. problems/problems/hanging_chain.cpp:107-110
//   initial state ......... problems/problems/hanging_chain.cpp:121-124
// The reference turns these into straight-line scalar C through CasADi.  A lone wavefront pays for every
// instruction it issues (5-6 cycles each, 8-12 in a dependent fp64 chain, 70 for an LDS round trip; DESIGN.md
// section 3), so the layout is chosen to issue few of them.
//
// The four f_c evaluations of a Runge-Kutta stage form TWO INDEPENDENT CHAINS OF TWO: the position substage 2
// evaluates the springs at is x + h/2 v - it needs no force -, substage 3 needs a_1 (through k2.x = v + h/2 a_1),
// substage 4 needs a_2.  The same holds for the adjoint substages (4 -> 2 and 3 -> 1) and for the tangent sweeps of
// the Hessian-vector products.  The two chains are the same program on different data, so the lower half of the
// wavefront evaluates one and the upper half the other WITH THE SAME INSTRUCTIONS:
//     lane = 32 chain + 10 component + point        (lanes 30, 31 of a half idle)
//     forward:  lower half substages (1, 3), upper half (2, 4);   adjoint: lower half (3, 1), upper half (4, 2)
// Both halves carry the full state (position and, for the nine free masses, velocity of component c of point p; the
// end point's lanes carry the stage input in the velocity slot); once per stage they exchange their two accelerations
// (adjoint: their two position adjoints) with v_permlane32_swap and form the Runge-Kutta combination by the expressions
// of the stage-by-stage form - same operations on the same operands, same bits.
//   * the neighbouring point is the neighbouring lane of the wavefront: DPP wave_shr:1 / wave_shl:1.  What lies beyond
//     a component's ends - the anchor before point 0, the lane after the end point - is masked: by a lane constant
//     inside an fma in the tangent sweeps (fma(-1, b, a) is a - b in one rounding, fma(-0, b, a) is a for finite b),
//     by a select in the first-order sweeps (whose operands may be non-finite);
//   * the three components of a point are 10 lanes apart: gathered through an LDS row (one write, three reads; a lone
//     chain has nothing to fill a gather's latency with, and ds_bpermute is no faster).  A 3-term sum over the
//     components is formed from the gathered values on every lane;
//   * position and velocity of a point share a lane.
// Rounds 2-4 ran both chains interleaved on one instruction stream with lane = (component, point) - 30 of 64 lanes
// busy; that form lives on in scripts/microbench/chain_row_layout.hip.h, where hv_pair_ab.hip measures one against the
// other (tangent sweeps: 528 + 750 cycles per stage against 711 + 949; two WAVEFRONTS with a chain each lose: the
// exchange through LDS and s_barrier costs 200 cycles a stage).
// Second-order information (Hessian-vector products): the gradient sweep at the current point records, per RK4
// substage and spring, the 3x3 matrices K = dF/dd and M = d(K lF)/dd; a product is then two linear sweeps of a few
// FMAs per substage over that tape.
#pragma once
#include "wave_math.hip.h"

namespace pantr {

struct ChainModel {
    static constexpr int NB = 9, DIM = 3, NP = NB + 1;
    static constexpr int nx = DIM * NP + DIM * NB; // 57
    static constexpr int nu = DIM;                 // 3
    static constexpr int VOFF = DIM * NP;          // velocities start at x0[30]
    static constexpr double Ts = 0.1, hTs = 0.5 * Ts;
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

    /// Index of entry (c, c2) in the upper-triangle order.
    __host__ __device__ static constexpr int sym(int c, int c2) {
        return c <= c2 ? (c == 0 ? c2 : c == 1 ? 2 + c2 : 5) : sym(c2, c);
    }

    /// A tape row is 64 doubles: ten springs x six entries and four unused slots; nobody reads the last one.
    static constexpr int TAPE_ROW_DUMP = 63;
    /// First slot of spring p's six entries in a row: 6p, with one slot left free before spring 8 (at a stride of 48
    /// bytes springs 8 and 9 would fall on the LDS banks of springs 0 and 1).
    __host__ __device__ static constexpr int spring_base(int p) { return 6 * p + (p >= 8 ? 1 : 0); }

    static constexpr int TAPE_STAGE_BYTES  = 4096;
    static constexpr int ST_STAGE_BYTES    = 1024;
    static constexpr int LDS_STAGE_BYTES   = TAPE_STAGE_BYTES + ST_STAGE_BYTES;
    static constexpr int TAPE_D4_PER_STAGE = TAPE_STAGE_BYTES / 32;
    static constexpr int ST_D4_PER_STAGE   = ST_STAGE_BYTES / 32;
    static constexpr int XT_DOUBLES        = 32; ///< tangent-state row per stage (compact slots)

    /// Per-lane role.
    struct Lane {
        int l, h, c, p, q; ///< lane, chain (half of the wavefront), component, point, compact slot 3p+c (idle lanes: 31)
        bool active, mass, end, hi, grav, lead, nfirst;
        double inv_m_l, mass_l, nend_l, nfirst_l; ///< lane constants of the tangent sweeps; nfirst_l: 0 on point 0 (and idle lanes)
        int k_off;  ///< byte offset in a landing buffer of this lane's spring block in the K row of the chain's FIRST forward substage
        int e8[DIM]; ///< byte offsets of the three entries of row c inside a spring block
        int st_off; ///< byte offset in a tangent-tape stage of this lane's d_t slot, first forward substage (the second: + 512)
        int dt_off; ///< byte offset in a landing buffer of this point's d_t triple, first ADJOINT substage (the second: - 512)
        int xs_w, xs_r; ///< doubles: own slot / first component of the point in the exchange row of this half
        __device__ explicit Lane(int lane) {
            l  = lane;
            h  = lane >> 5;
            hi = h != 0;
            const int r = lane & 31;
            active = r < DIM * NP;
            c      = active ? r / NP : 0;
            p      = active ? r % NP : NB;
            mass   = active && p < NB;
            end    = active && p == NB;
            grav   = mass && c == DIM - 1;
            lead   = active && c == 0 && !hi; // lanes 0..9: one per point (cost_total)
            nfirst = active && p > 0;
            inv_m_l  = mass ? inv_m : 0.0;
            mass_l   = mass ? 1.0 : 0.0;
            nend_l   = end ? 0.0 : 1.0;
            nfirst_l = nfirst ? 1.0 : 0.0;
            q        = active ? DIM * p + c : 31;
            // forward substages (0-based): first = h, second = 2 + h; adjoint: first = 2 + h, second = h
            k_off = h * 512 + spring_base(p) * 8;
            for (int k = 0; k < DIM; ++k)
                e8[k] = sym(c, k) * 8;
            st_off = h * 256 + q * 8;
            dt_off = TAPE_STAGE_BYTES + (2 + h) * 256 + DIM * p * 8;
            xs_w   = 32 * h + q;
            xs_r   = 32 * h + DIM * p;
        }
    };
    using PairLane = Lane;

    /// Component of the state (or adjoint, or time derivative) of one chain point.
    struct State {
        double x, v;
    };

    // ---- cross-lane ----------------------------------------------------------------------------
    static constexpr int PAIR_XS_DOUBLES = 64; ///< exchange rows of the component gather: 32 doubles per half
    /// Same component of the previous / next point: the neighbouring lane (wave_shr:1: lane_below, wave_math.hip.h;
    /// wave_shl:1).  What a lane at an end of its component receives belongs to another component: the callers mask it.
    __device__ static __forceinline__ double lane_above(double v) { return dpp_move<0x130>(v); } // wave_shl:1
    /// The three components of this lane's point: one 8-byte write to the point's compact slot in this half's exchange
    /// row, three reads.  xs: PAIR_XS_DOUBLES doubles of this wavefront's dynamic LDS.  LDS executes a wavefront's
    /// accesses in order: the fences are for the compiler only.
    struct Comps {
        double c0, c1, c2;
    };
    __device__ static __forceinline__ Comps comps(const Lane &L, double v, double *xs) {
        xs[L.xs_w] = v;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double *s = xs + L.xs_r;
        return {s[0], s[1], s[2]};
    }
    /// The lower half's and the upper half's value, on every lane (one v_permlane32_swap per 32-bit half).
    __device__ static __forceinline__ void halves(double v, double &lower, double &upper) { row_partner<true>(v, lower, upper); }

    // ---- memory <-> registers ------------------------------------------------------------------
    /// Initial state from the reference's state vector layout [vec(x) (30); vec(v) (27)]
    /// (hanging_chain.cpp:70-73).
    __device__ static __forceinline__ State load_x0(const Lane &L, const double *x0) {
        State s{0.0, 0.0};
        if (L.active)
            s.x = x0[L.q];
        if (L.mass)
            s.v = x0[VOFF + L.q];
        return s;
    }
    __device__ static __forceinline__ void store_x(const Lane &L, double *x, const State &s) {
        if (L.active && !L.hi)
            x[L.q] = s.x;
        if (L.mass && !L.hi)
            x[VOFF + L.q] = s.v;
    }
    /// Trajectory rows in the HBM workspace: {x, v} per lane, 1 KiB per stage (both halves hold the same values).
    static constexpr int TRAJ_DOUBLES = 2 * WAVE;
    __device__ static __forceinline__ void traj_store(const Lane &L, double *row, const State &s) {
        reinterpret_cast<double2 *>(row)[L.l] = make_double2(s.x, s.v);
    }
    __device__ static __forceinline__ State traj_load(const Lane &L, const double *row) {
        double2 a = reinterpret_cast<const double2 *>(row)[L.l];
        return {a.x, a.y};
    }
    /// Tangent-state row of the Hv sweeps: one double per compact slot, the component the cost
    /// Hessian multiplies (x of the end point, v of the masses).  (Both halves store the same value to the same slot.)
    __device__ static __forceinline__ void xt_store(const Lane &L, double *row, const State &s_t) {
        row[L.q] = L.end ? s_t.x : s_t.v; // idle lanes: slot 31
    }
    __device__ static __forceinline__ double xt_load(const Lane &L, const double *row) { return row[L.q]; }
    /// This lane's component of u_k from a decision vector in LDS.
    __device__ static __forceinline__ double load_input(const Lane &L, const double *u_k) { return u_k[L.c]; }
    /// The end point's lanes (of the lower half) own the adjoint of the stage input.
    __device__ static __forceinline__ void store_input_grad(const Lane &L, double *g_k, double g) {
        if (L.end && !L.hi)
            g_k[L.c] = g;
    }

    // ---- springs (primal) -------------------------------------------------------------------------
    struct Springs {
        double d, n2, r, c;
        Comps dq; ///< d of all three components of the point (the gather that forms n2 serves the tape recording too)
    };
    __device__ static __forceinline__ void springs(const Lane &L, double x, Springs &sp, double *xs) {
        const double pv = lane_below(x);
        sp.d  = x - (L.nfirst ? pv : 0.0); // point 0 is anchored at the origin: x - 0
        sp.dq = comps(L, sp.d, xs);
        // (d0^2 + d1^2) + d2^2: the sum of the squares over the components, formed on every lane from the gathered values
        sp.n2 = (sp.dq.c0 * sp.dq.c0 + sp.dq.c1 * sp.dq.c1) + sp.dq.c2 * sp.dq.c2;
        sp.r  = sqrt_inrange(sp.n2); // = sqrt(n2), L / r: wave_math.hip.h
        sp.c  = Dspring * (1.0 - div_inrange(Lspring, sp.r));
    }
    /// The end point has no velocity state: its lanes carry the stage input there instead, so the
    /// time derivative of every position is just "the v slot" (with_input() once per stage; the
    /// slot's own derivative is 0 off the masses, so the input stays put through the Runge-Kutta
    /// combinations).
    __device__ static __forceinline__ State with_input(const Lane &L, const State &s, double uc) {
        return {s.x, L.end ? uc : s.v};
    }
    /// The acceleration part of f_c at position x (0 off the masses).
    __device__ static __forceinline__ double accel(const Lane &L, double x, Springs &sp, double *xs) {
        springs(L, x, sp, xs);
        double F  = sp.d * sp.c;
        double Fn = lane_above(F); // force of the spring towards point p+1
        double a  = inv_m * (Fn - F);
        a         = L.grav ? a - g_grav : a;
        return L.mass ? a : 0.0; // a select, not inv_m_l: lanes beyond the chain hold 0 * inf
    }
    struct AdjAux {
        double lF, dt, w;
        Comps lq; ///< lF of all three components of the point
    };
    /// Adjoint of f_c at the springs `sp`: kv = the velocity part of the incoming adjoint; returns the position part
    /// of the adjoint of s (the velocity part is `mass ? kbar.x : 0`, formed by the caller).
    __device__ static __forceinline__ double fc_adj(const Lane &L, const Springs &sp, double kv, AdjAux &ax, double *xs) {
        const double pk = lane_below(kv);
        double A   = L.nfirst ? pk : 0.0; // adjoint of a_{p-1} (0 for point 0)
        double B   = kv;                  // adjoint of a_p (stays 0 off the masses)
        ax.lF      = inv_m * (A - B);
        ax.lq      = comps(L, ax.lF, xs);
        ax.dt      = (sp.dq.c0 * ax.lq.c0 + sp.dq.c1 * ax.lq.c1) + sp.dq.c2 * ax.lq.c2; // sum over the components of d lF
        double n2r = sp.n2 * sp.r;
        ax.w       = div_inrange(DL, n2r);
        double t   = ax.dt * ax.w;
        double ld  = (sp.c * ax.lF) + (t * sp.d);
        double ldn = lane_above(ld);
        return L.end ? ld : ld - ldn;
    }

    /// One RK4 step of the plant (formulation.hpp:21-28): the lower half evaluates substages (1, 3), the upper (2, 4).
    /// spf / sps: the springs of this half's first / second substage (the adjoint sweep meets them again).
    __device__ static __forceinline__ State rk4(const Lane &L, const State &s_in, double uc, Springs &spf, Springs &sps,
                                                double *xs) {
        const State s = with_input(L, s_in, uc);
        // k_j = {kx_j, a_j}: kx_1 = v, kx_{j+1} = v + c_j a_j; a_j = a(x + c_{j-1} kx_{j-1})
        const double x2 = s.x + 0.5 * (Ts * s.v);
        const double af = accel(L, L.hi ? x2 : s.x, spf, xs);             // a1 | a2
        const double kn = s.v + 0.5 * (Ts * af);                           // kx2 | kx3
        const double t  = Ts * kn;
        const double as = accel(L, s.x + (L.hi ? t : 0.5 * t), sps, xs);   // a3 (at x + h/2 kx2) | a4 (at x + h kx3)
        double a1, a2, a3, a4;
        halves(af, a1, a2);
        halves(as, a3, a4);
        const double kx2 = s.v + 0.5 * (Ts * a1), kx3 = s.v + 0.5 * (Ts * a2), kx4 = s.v + Ts * a3;
        return {s.x + c6 * (((s.v + 2.0 * kx2) + 2.0 * kx3) + kx4), s.v + c6 * (((a1 + 2.0 * a2) + 2.0 * a3) + a4)};
    }
    __device__ static __forceinline__ State rk4(const Lane &L, const State &s_in, double uc, double *xs) {
        Springs spf, sps;
        return rk4(L, s_in, uc, spf, sps, xs);
    }

    // ---- the tape ------------------------------------------------------------------------------
    // Per substage j and spring p: the upper triangles of K = c I + w d d^T and of
    // M = w [ lF d^T + d lF^T + dt (I - (3/n2) d d^T) ] (both symmetric), six doubles each in the
    // order (0,0) (0,1) (0,2) (1,1) (1,2) (2,2): spring p's block at doubles [spring_base(p), +6) of a
    // 512-byte row.  Stage layout in HBM (4 KiB): [K rows of j = 0..3] (2 KiB: all the forward sweep
    // needs) then [M rows of j = 0..3].  Lane (c, p) of a half reads row c of the full matrices of its substage.
    // The forward sweep of each product writes its tangent spring vectors d_t to a second tape
    // ("st": [j][32 slots] doubles, 1 KiB per stage) that its adjoint sweep streams back, and one
    // 256-byte row per stage with the tangent-state components the cost Hessian needs.

    /// The K and M rows of substage j (each half its own) from the springs and the adjoint quantities of that substage.
    __device__ static __forceinline__ void tape_store(const Lane &L, double4 *stage, int j, const Springs &sp, const AdjAux &ax) {
        const double dq[DIM] = {sp.dq.c0, sp.dq.c1, sp.dq.c2}, lq[DIM] = {ax.lq.c0, ax.lq.c1, ax.lq.c2};
        const double e  = div_inrange(3.0, sp.n2);
        const double wd = ax.w * sp.d;
        const double g  = e * sp.d;
        // Every lane computes its three entries of the row and stores them - the lanes that own no entry
        // of the upper triangle (c > c2, idle lanes) into the row's unread last slot: stores under an exec
        // mask would put each entry's arithmetic into a basic block of its own, which the scheduler cannot
        // overlap with the adjoint recursion.
        double *Krow = reinterpret_cast<double *>(reinterpret_cast<char *>(stage) + j * 512);
        double *Mrow = reinterpret_cast<double *>(reinterpret_cast<char *>(stage) + 2048 + j * 512);
#pragma unroll
        for (int c2 = 0; c2 < DIM; ++c2) {
            double K  = __builtin_fma(wd, dq[c2], L.c == c2 ? sp.c : 0.0);
            double m  = ax.lF * dq[c2];
            m         = __builtin_fma(sp.d, lq[c2], m);
            double kd = __builtin_fma(-g, dq[c2], L.c == c2 ? 1.0 : 0.0);
            m         = __builtin_fma(ax.dt, kd, m);
            double M  = ax.w * m;
            // slot of entry (c, c2): recomputed here (three integer operations) rather than kept in the
            // lane's registers for the whole solve
            const int slot = (L.active && L.c <= c2) ? spring_base(L.p) + sym(L.c, c2) : TAPE_ROW_DUMP;
            Krow[slot] = K;
            Mrow[slot] = M;
        }
    }
    /// Start the LDS-DMA of one stage into the landing buffer `buf`: 2 KiB (K rows) for the
    /// forward sweep; 4 KiB of tape + 1 KiB of tangent tape for the adjoint sweep.
    /// The transfers of a stage: immediate offsets on one address register pair.  GROUP: as asm groups with one M0
    /// write each (wave_math.hip.h, lds_dma_rows) - the sweeps that wait for their rows with explicit vmcnt counts
    /// (two stages ahead) use these; the one-stage-ahead sweeps keep the builtin form, whose transfers the
    /// compiler schedules among the stage's other instructions.
    static constexpr int FWD_ROW_TRANSFERS = 2; ///< vector-memory operations of prefetch_fwd (the K rows: 2 KiB)
    static constexpr int REV_ROW_TRANSFERS = 6; ///< of prefetch_rev_xt (K, M: 4; d_t rows: 1; tangent-state row: 1)
    template <bool GROUP = false>
    __device__ static __forceinline__ void prefetch_fwd(const Lane &L, const double4 *tape_stage, char *buf) {
        const char *src = reinterpret_cast<const char *>(tape_stage) + L.l * 16;
        if constexpr (GROUP) {
            lds_dma_rows<FWD_ROW_TRANSFERS>(src, buf);
        } else {
            lds_dma_row<0>(src, buf);
            lds_dma_row<1024>(src, buf);
        }
    }
    /// The M rows of a stage alone, to their place in the landing buffer (WaveProblem::prime_d1, prime).
    template <bool GROUP = false>
    __device__ static __forceinline__ void prefetch_m(const Lane &L, const double4 *tape_stage, char *buf) {
        const char *src = reinterpret_cast<const char *>(tape_stage) + L.l * 16;
        if constexpr (GROUP) {
            lds_dma_rows<2>(src + 2048, buf + 2048);
        } else {
            lds_dma_row<2048>(src, buf);
            lds_dma_row<3072>(src, buf);
        }
    }
    /// The tangent tape and the tangent-state row of a stage alone, to their places (two transfers, as asm groups).
    __device__ static __forceinline__ void prefetch_st_xt(const Lane &L, const double4 *st_stage, const double *xt_row, char *buf) {
        lds_dma_rows<1>(reinterpret_cast<const char *>(st_stage) + L.l * 16, buf + TAPE_STAGE_BYTES);
        lds_dma_256(reinterpret_cast<const char *>(xt_row) + L.l * 4, buf + LDS_STAGE_BYTES);
    }
    template <bool GROUP = false>
    __device__ static __forceinline__ void prefetch_rev(const Lane &L, const double4 *tape_stage,
                                                        const double4 *st_stage, char *buf) {
        static_assert(TAPE_STAGE_BYTES == 4096 && ST_STAGE_BYTES == 1024, "four + one 1 KiB transfers");
        const char *src = reinterpret_cast<const char *>(tape_stage) + L.l * 16;
        if constexpr (GROUP) {
            lds_dma_rows<4>(src, buf);
            lds_dma_rows<1>(reinterpret_cast<const char *>(st_stage) + L.l * 16, buf + TAPE_STAGE_BYTES);
        } else {
            lds_dma_row<0>(src, buf);
            lds_dma_row<1024>(src, buf);
            lds_dma_row<2048>(src, buf);
            lds_dma_row<3072>(src, buf);
            lds_dma_row<0>(reinterpret_cast<const char *>(st_stage) + L.l * 16, buf + TAPE_STAGE_BYTES);
        }
    }
    /// The same plus the stage's tangent-state row (256 bytes, one 4-byte transfer per lane) behind the tangent
    /// tape in the landing buffer: sweeps that run two stages ahead read it from LDS (xt_lds) instead of waiting
    /// for a register load among the transfers in flight.  Six transfers.
    static constexpr int LDS_STAGE_BYTES_XT = LDS_STAGE_BYTES + XT_DOUBLES * 8;
    __device__ static __forceinline__ void prefetch_rev_xt(const Lane &L, const double4 *tape_stage, const double4 *st_stage,
                                                           const double *xt_row, char *buf) {
        static_assert(REV_ROW_TRANSFERS == 4 + 1 + 1, "prefetch_rev<true>: lds_dma_rows<4> + lds_dma_rows<1>; then one lds_dma_256");
        prefetch_rev<true>(L, tape_stage, st_stage, buf);
        lds_dma_256(reinterpret_cast<const char *>(xt_row) + L.l * 4, buf + LDS_STAGE_BYTES);
    }
    __device__ static __forceinline__ double xt_lds(const Lane &L, const char *buf) {
        return reinterpret_cast<const double *>(buf + LDS_STAGE_BYTES)[L.q];
    }
    // ---- products that keep the tangent STATE of every stage instead of its tangent spring vectors (large batches) ----
    // Where a launch is bound by the memory system, bytes are what a product costs: the tangent tape (1 KiB per stage
    // written, 1 KiB read back) and the tangent-state row (256 B each way) are 2.5 of the 9.25 KiB a stage moves.  The
    // forward sweep can store the tangent state at the stage's start instead - 57 numbers: one 512-byte row, positions from
    // the lower half of the wavefront, velocities from the upper - and the adjoint sweep run the stage's forward tangent
    // once more from it (86 instructions; the K rows are in the landing buffer anyway), which puts the d_t where the
    // tape would have landed them: 7 KiB per stage.  Same expressions on the same operands: same bits.
    static constexpr int S_ROW_DOUBLES = 64; ///< state row per stage: x at slots 0..31 (lane & 31), v at 32..63
    __device__ static __forceinline__ void s_store(const Lane &L, double *row, const State &s_t) {
        row[(L.l & 31) + (L.hi ? 32 : 0)] = L.hi ? s_t.v : s_t.x;
    }
    /// The state row of a stage (two 256-byte transfers) to the start of the landing buffer's tangent-tape area: it is
    /// read into registers before the stage's recomputed d_t are written over it.
    __device__ static __forceinline__ void prefetch_s(const Lane &L, const double *srow, char *buf) {
        lds_dma_256(reinterpret_cast<const char *>(srow) + L.l * 4, buf + TAPE_STAGE_BYTES);
        lds_dma_256(reinterpret_cast<const char *>(srow) + 256 + L.l * 4, buf + TAPE_STAGE_BYTES + 256);
    }
    /// K and M rows and the state row of a stage: REV_ROW_TRANSFERS operations, like prefetch_rev_xt.
    __device__ static __forceinline__ void prefetch_rev_s(const Lane &L, const double4 *tape_stage, const double *srow, char *buf) {
        static_assert(REV_ROW_TRANSFERS == 4 + 2, "lds_dma_rows<4>, then two lds_dma_256");
        lds_dma_rows<4>(reinterpret_cast<const char *>(tape_stage) + L.l * 16, buf);
        prefetch_s(L, srow, buf);
    }
    __device__ static __forceinline__ State s_lds(const Lane &L, const char *buf) {
        const double *s = reinterpret_cast<const double *>(buf + TAPE_STAGE_BYTES) + (L.l & 31);
        return {s[0], s[32]};
    }

    // ---- tangent sweeps over the tape ---------------------------------------------------------------
    /// st stores of a forward stage: one per substage PAIR (each stores both chains' d_t)
    static constexpr int PAIR_ST_STORES_PER_STAGE = 2;
    __device__ static __forceinline__ double lds_at(const char *base, int off) {
        return *reinterpret_cast<const double *>(base + off);
    }
    /// Tangent of f_c's acceleration, one substage of each chain: a = inv_m (F_{p+1} - F_p), F = K d_t; K at
    /// buf + koff (+ e8[.]); stores d_t to st_stage + st_off.
    __device__ static __forceinline__ double fc_tan_pair(const Lane &L, const char *buf, int koff, int st_off, double x,
                                                         char *st_stage, double *xs) {
        const double d = __builtin_fma(-L.nfirst_l, lane_below(x), x); // x - prev (point 0: the anchor at the origin)
        *reinterpret_cast<double *>(st_stage + st_off) = d;            // idle lanes: slot 31
        const char *kb = buf + koff;
        const double K0 = lds_at(kb, L.e8[0]), K1 = lds_at(kb, L.e8[1]), K2 = lds_at(kb, L.e8[2]);
        const Comps c = comps(L, d, xs);
        double F = K0 * c.c0;
        F        = __builtin_fma(K1, c.c1, F);
        F        = __builtin_fma(K2, c.c2, F);
        return L.inv_m_l * (lane_above(F) - F);
    }
    /// Tangent of the f_c adjoint, one substage of each chain: the position part ld_p - ld_{p+1} of
    /// ld_t = K lF_t + M d_t, lF_t = inv_m (kv_{p-1} - kv_p); kv = the velocity part of the substage's
    /// incoming adjoint (0 off the masses).
    __device__ static __forceinline__ double fc_adj_tan_pair(const Lane &L, const char *buf, int koff, int dt_off, double kv,
                                                             double *xs) {
        const double lF = inv_m * (L.nfirst_l * lane_below(kv) - kv);
        const char *kb = buf + koff;
        const double K0 = lds_at(kb, L.e8[0]), K1 = lds_at(kb, L.e8[1]), K2 = lds_at(kb, L.e8[2]);
        const double M0 = lds_at(kb, 2048 + L.e8[0]), M1 = lds_at(kb, 2048 + L.e8[1]), M2 = lds_at(kb, 2048 + L.e8[2]);
        const double D0 = lds_at(buf, dt_off), D1 = lds_at(buf, dt_off + 8), D2 = lds_at(buf, dt_off + 16);
        const Comps c = comps(L, lF, xs);
        double ld = K0 * c.c0;
        ld        = __builtin_fma(K1, c.c1, ld);
        ld        = __builtin_fma(K2, c.c2, ld);
        ld        = __builtin_fma(M0, D0, ld);
        ld        = __builtin_fma(M1, D1, ld);
        ld        = __builtin_fma(M2, D2, ld);
        // end ? ld : ld - n, without the select: one rounding either way; the lane after the end point holds a
        // finite value, so 0 * n is 0
        return __builtin_fma(-L.nend_l, lane_above(ld), ld);
    }
    /// Tangent of one RK4 step; the stage's K rows are in LDS at `buf`.
    __device__ static __forceinline__ State rk4_tan(const Lane &L, const char *buf, const State &s_in, double uc_t,
                                                    double4 *st_stage, double *xs) {
        char *st = reinterpret_cast<char *>(st_stage);
        const double x = s_in.x, v = L.end ? uc_t : s_in.v;
        // k_j = {kx_j, a_j}: kx_1 = v, kx_{j+1} = v + c_j a_j; a_j = a(x + c_{j-1} kx_{j-1})
        const double x2  = __builtin_fma(hTs, v, x);
        const double af  = fc_tan_pair(L, buf, L.k_off, L.st_off, L.hi ? x2 : x, st, xs);             // a1 | a2
        const double kxf = __builtin_fma(hTs, af, v);                                                 // kx2 | kx3
        const double xs2 = __builtin_fma(L.hi ? Ts : hTs, kxf, x);                                    // x3 | x4
        const double as  = fc_tan_pair(L, buf, L.k_off + 1024, L.st_off + 512, xs2, st, xs);          // a3 | a4
        double a1, a2, a3, a4;
        halves(af, a1, a2);
        halves(as, a3, a4);
        const double kx2 = __builtin_fma(hTs, a1, v), kx3 = __builtin_fma(hTs, a2, v);
        const double kx4 = __builtin_fma(Ts, a3, v);
        double tx = __builtin_fma(2.0, kx2, v), tv = __builtin_fma(2.0, a2, a1);
        tx        = __builtin_fma(2.0, kx3, tx);
        tv        = __builtin_fma(2.0, a3, tv);
        tx        = tx + kx4;
        tv        = tv + a4;
        return {__builtin_fma(c6, tx, x), __builtin_fma(c6, tv, v)};
    }
    /// Tangent of the RK4 adjoint (second-order adjoint): each half meets the d_t it stored in the forward sweep.
    /// ubar_t is valid on the end point's lanes.
    __device__ static __forceinline__ void rk4_adj_tan(const Lane &L, const char *buf, const State &mu_t, State &sbar_t,
                                                       double &ubar_t, double *xs) {
        // incoming substage adjoints kb_j = {kbx_j, kbv_j}; the adjoint of f_c returns {G_j, m_j}:
        // G_j from kbv_j (springs), m_j = kbx_j on the masses (0 elsewhere: keeps kbv at 0 off the masses)
        const double sgx = c6 * mu_t.x, sgv = c6 * mu_t.v;
        const double sgx2 = 2.0 * sgx, sgv2 = 2.0 * sgv;
        // m = mass ? x : 0 as one multiplication (the x of these sweeps are finite; a zero's sign may differ)
        const double m4   = L.mass_l * sgx;
        const double kbv3 = __builtin_fma(Ts, m4, sgv2);
        const double Gf   = fc_adj_tan_pair(L, buf, L.k_off + 1024, L.dt_off, L.hi ? sgv : kbv3, xs); // G3 | G4
        const double kbx  = __builtin_fma(L.hi ? Ts : hTs, Gf, sgx2);                                 // kbx2 | kbx3
        const double mm   = L.mass_l * kbx;
        const double kvs  = __builtin_fma(hTs, mm, L.hi ? sgv2 : sgv);                                // kbv1 | kbv2
        const double Gs   = fc_adj_tan_pair(L, buf, L.k_off, L.dt_off - 512, kvs, xs);                // G1 | G2
        double G3, G4, G1, G2;
        halves(Gf, G3, G4);
        halves(Gs, G1, G2);
        const double kbx3 = __builtin_fma(Ts, G4, sgx2), kbx2 = __builtin_fma(hTs, G3, sgx2);
        const double kbx1 = __builtin_fma(hTs, G2, sgx);
        const double m3 = L.mass_l * kbx3, m2 = L.mass_l * kbx2, m1 = L.mass_l * kbx1;
        sbar_t = {(((mu_t.x + G4) + G3) + G2) + G1, (((mu_t.v + m4) + m3) + m2) + m1};
        ubar_t = ((sgx + kbx3) + kbx2) + kbx1;
    }

    /// Adjoint of one RK4 step.  mu = adjoint of x_next; sbar = adjoint of s; ubar (valid on the
    /// end point's lanes) = adjoint of u through the dynamics.  With tape_stage != nullptr the
    /// K and M rows of the four substages are recorded (each half its two).
    __device__ static __forceinline__ void rk4_adj(const Lane &L, const State &s_in, double uc, const State &mu,
                                                   State &sbar, double &ubar, double4 *tape_stage, double *xs) {
        // the springs of the four substages: lower half 1 and 3, upper half 2 and 4 (rk4's evaluation order)
        Springs spf, sps;
        {
            const State s   = with_input(L, s_in, uc);
            const double x2 = s.x + 0.5 * (Ts * s.v);
            const double af = accel(L, L.hi ? x2 : s.x, spf, xs);
            const double kn = s.v + 0.5 * (Ts * af);
            const double t  = Ts * kn;
            springs(L, s.x + (L.hi ? t : 0.5 * t), sps, xs);
        }
        // incoming substage adjoints kb_j = {kbx_j, kbv_j}: kb_4 = kb_1 = c6 mu, kb_3 = kb_2 = 2 c6 mu, then
        // kb_3 += Ts a_4, kb_2 += Ts/2 a_3, kb_1 += Ts/2 a_2 with a_j = {G_j, m_j} the adjoint of f_c at substage j
        // (G_j from kbv_j through the springs, m_j = kbx_j on the masses): G_2 needs only G_4, G_1 only G_3.
        // Upper half: substages 4 then 2 (its springs sps, spf); lower half: 3 then 1.
        AdjAux ax;
        const double sgx = c6 * mu.x, sgv = c6 * mu.v;
        const double sgx2 = 2.0 * sgx, sgv2 = 2.0 * sgv;
        const double m4   = L.mass ? sgx : 0.0;
        const double kbv3 = sgv2 + Ts * m4;
        const double Gf   = fc_adj(L, sps, L.hi ? sgv : kbv3, ax, xs);            // G3 | G4
        if (tape_stage)
            tape_store(L, tape_stage, 2 + L.h, sps, ax);
        const double tf  = Ts * Gf;
        const double kbx = sgx2 + (L.hi ? tf : 0.5 * tf);                          // kbx2 | kbx3
        const double mm  = L.mass ? kbx : 0.0;
        const double kvs = (L.hi ? sgv2 : sgv) + 0.5 * (Ts * mm);                  // kbv1 | kbv2
        const double Gs  = fc_adj(L, spf, kvs, ax, xs);                            // G1 | G2
        if (tape_stage)
            tape_store(L, tape_stage, L.h, spf, ax);
        double G3, G4, G1, G2;
        halves(Gf, G3, G4);
        halves(Gs, G1, G2);
        const double kbx3 = sgx2 + Ts * G4, kbx2 = sgx2 + 0.5 * (Ts * G3), kbx1 = sgx + 0.5 * (Ts * G2);
        const double m3 = L.mass ? kbx3 : 0.0, m2 = L.mass ? kbx2 : 0.0, m1 = L.mass ? kbx1 : 0.0;
        sbar = {(((mu.x + G4) + G3) + G2) + G1, (((mu.v + m4) + m3) + m2) + m1};
        ubar = ((sgx + kbx3) + kbx2) + kbx1;
    }

    // ---- cost (hanging_chain.cpp:127-131) ---------------------------------------------------------
    /// Per-component running sums: the end point's lanes add alpha (x_end - ref)^2 then gamma u^2
    /// to acc.x, the masses' lanes add beta v^2 to acc.v.
    __device__ static __forceinline__ void cost_accumulate(const Lane &L, const State &s, double uc, bool has_u,
                                                           State &acc) {
        double t  = s.x - (L.c == 0 ? 1.0 : 0.0); // reference position of the end point: (1, 0, 0)
        double a1 = alpha * (t * t);
        double b1 = beta * (s.v * s.v);
        acc.x     = L.end ? acc.x + a1 : acc.x;
        acc.v     = L.mass ? acc.v + b1 : acc.v;
        if (has_u) {
            double a2 = gamma * (uc * uc);
            acc.x     = L.end ? acc.x + a2 : acc.x;
        }
    }
    /// Total cost: per point ((((x0 + x1) + x2) + v0) + v1) + v2 on the lane of its component 0 in the lower half
    /// (lanes 0..9), zero elsewhere, then the 64-lane butterfly: its two cross-row levels add zeros, its four in-row
    /// levels pair p with p^8, p^4, p^2, p^1 - the additions, in the order, of the oracle's butterfly with point p
    /// at lane 4p.
    __device__ static __forceinline__ double cost_total(const Lane &L, const State &acc, double *xs) {
        const Comps x = comps(L, acc.x, xs), v = comps(L, acc.v, xs);
        double p = ((((x.c0 + x.c1) + x.c2) + v.c0) + v.c1) + v.c2;
        return uniform(wave_sum(L.lead ? p : 0.0));
    }
    /// lam + d(l or l_N)/dx.
    __device__ static __forceinline__ void cost_grad_x_add(const Lane &L, const State &s, State &lam) {
        double ge = c2a * (s.x - (L.c == 0 ? 1.0 : 0.0));
        double gv = c2b * s.v;
        lam.x     = L.end ? lam.x + ge : lam.x;
        lam.v     = L.mass ? lam.v + gv : lam.v;
    }
    /// Tangent of cost_grad_x_add; xt = the lane's xt_store()d component of s_t.
    __device__ static __forceinline__ void cost_grad_x_add_tan(const Lane &L, double xt, State &lam_t) {
        lam_t.x = L.end ? __builtin_fma(c2a, xt, lam_t.x) : lam_t.x;
        lam_t.v = L.mass ? __builtin_fma(c2b, xt, lam_t.v) : lam_t.v;
    }
};

} // namespace pantr
