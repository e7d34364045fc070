// THE ROW LAYOUT OF ROUNDS 2-4, kept for the A/B microbenchmarks (hv_layout_ab.hip, hv_pair_ab.hip) only: the product
// library's model is csrc/chain_model.hip.h, where both substage chains of a stage run side by side on the halves of the
// wavefront (round 5).  Not part of the library, not in its source hash.
//
// Hanging-chain stage model for one 64-wide wavefront (gfx950), lane = (chain point, component).
//
// What the reference computes (paths relative to /root/reference):
//   constants ............ problems/problems/hanging_chain.cpp:28-42
//   f_c (springs, gravity) problems/problems/hanging_chain.cpp:85-100
//   RK4 ................... problems/problems/formulation.hpp:21-28
//   l, l_N ................ problems/problems/hanging_chain.cpp:127-132
//   input box ............. problems/problems/hanging_chain.cpp:107-110
//   initial state ......... problems/problems/hanging_chain.cpp:121-124
// The reference turns these into straight-line scalar C through CasADi.  Here lane 16c+p
// (c = 0..2: one 16-lane row per component, row 3 idle; p = 0..9, lanes 10..15 of a row idle) owns
// component c of chain point p: its position and, for the nine free masses p < 9, its velocity.
// A lone wavefront pays for every instruction it issues (~5 cycles independent, 8-12 in a
// dependent fp64 chain, 60 for an LDS crossbar round trip), so the layout
// minimises instructions:
//   * one instruction handles a component of all ten points,
//   * the neighbouring point is the neighbouring lane: one DPP row shift (two v_mov_b32_dpp per
//     fp64 value; the anchor before point 0 is the zero bound_ctrl supplies),
//   * the three components of a point sit at one lane position of rows 0..2: a 3-term sum is the
//     cross-row sum (v_permlane16_swap, v_permlane32_swap), all three components reach every row
//     with three swaps,
//   * position and velocity of a point share a lane (no shuffle between them).
// (Round 1 kept the components in a DPP quad, lane 4p+c, and paid eight moves per neighbour
// exchange; a substage has two neighbour exchanges and one component gather.)
// Second-order information (Hessian-vector products): the gradient sweep at the current point
// records, per RK4 substage and spring, the 3x3 matrices K = dF/dd and M = d(K lF)/dd; a product
// is then two linear sweeps of a few FMAs per substage over that tape.
#pragma once
#include "../../pantr-cdc2023-experiments_amd/csrc/wave_math.hip.h"

namespace pantr {

struct ChainRowModel {
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
    /// First slot of spring p's six entries in a row: 6p, with one slot left free before spring 8.  A
    /// 16-lane row of the wavefront reads entry e of its ten springs in one LDS access over 32 banks of
    /// 4 bytes: at a stride of 48 bytes springs 8 and 9 would fall on the banks of springs 0 and 1 (a
    /// two-way conflict in every read of K and M); shifted by 8 bytes they fall on free ones.
    __host__ __device__ static constexpr int spring_base(int p) { return 6 * p + (p >= 8 ? 1 : 0); }
    /// Per-lane role.
    struct Lane {
        int l, p, c, q; ///< lane, point, component, compact slot 3p+c
        int pc;         ///< point clamped to 0..9 (idle lanes read valid LDS)
        int e[DIM];     ///< upper-triangle indices of row c of a symmetric 3x3
        bool active, mass, end, first, lead, grav;
        double inv_m_l; ///< 1/m on the masses' lanes, 0 elsewhere (keeps idle lanes' derivatives at 0)
        double mass_l;  ///< 1 on the masses' lanes, 0 elsewhere: `mass ? x : 0` as one multiplication in the tangent sweeps
        double nend_l;  ///< 0 on the end point's lanes, 1 elsewhere: `end ? a : a - b` as fma(-nend_l, b, a)
        __device__ explicit Lane(int lane) {
            l      = lane;
            p      = lane & 15;
            c      = lane >> 4;
            active = p < NP && c < DIM;
            mass   = active && p < NB;
            end    = active && p == NB;
            first  = p == 0;
            lead   = active && c == 0;
            grav   = mass && c == DIM - 1;
            inv_m_l = mass ? inv_m : 0.0;
            mass_l  = mass ? 1.0 : 0.0;
            nend_l  = end ? 0.0 : 1.0;
            q      = active ? DIM * p + c : 31;
            pc     = p < NP ? p : NP - 1;
            int cc = c < DIM ? c : 0;
            for (int k = 0; k < DIM; ++k)
                e[k] = sym(cc, k);
        }
    };

    /// Component of the state (or adjoint, or time derivative) of one chain point.
    struct State {
        double x, v;
    };

    // ---- cross-lane ----------------------------------------------------------------------------
    /// The three components of this lane's point, on every row: v_permlane16_swap pairs rows (0,1)
    /// and (2,3), v_permlane32_swap pairs (0,2) and (1,3)  (scripts/microbench/dpp_row_rotate_test.hip).
    struct Comps {
        double c0, c1, c2;
    };
    __device__ static __forceinline__ Comps comps(double v) {
        double even, odd, c0, c2, c1, c3;
        row_partner<false>(v, even, odd); // rows 0,1: (row 0, row 1); rows 2,3: (row 2, row 3)
        row_partner<true>(even, c0, c2);  // row 0's and row 2's value, on all rows
        row_partner<true>(odd, c1, c3);   // row 1's (and the idle row 3's)
        return {c0, c1, c2};
    }
    /// The same through LDS: one 8-byte write to the point's compact slot, three reads - 3 instructions
    /// against 12 (copies + swaps), a round trip ~25 cycles longer.  The sweeps issue an instruction
    /// every 6-7 cycles and their dependent chains are shorter than that adds up to, so instructions are
    /// what counts: one product 58.8 -> 52.6 us at N = 60 (round 3).  xs: 32 doubles of this wavefront's
    /// dynamic LDS.  LDS executes a wavefront's accesses in order: the fence is for the compiler only.
    __device__ static __forceinline__ Comps comps_lds(const Lane &L, double v, double *xs) {
        xs[L.q] = v;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double *s = xs + DIM * L.pc;
        return {s[0], s[1], s[2]};
    }
    /// Two gathers side by side, one each way: A through the permlane swaps (12 instructions, ~45 cycles),
    /// B through the exchange row (3 instructions, ~120 cycles behind the tape reads queued before it).
    /// The paired sweeps are bound by their dependent chains, with issue slots to spare: the swaps of
    /// one chain fill the LDS wait of the other (both through LDS: 51.6 us per product at N = 60, both
    /// through swaps: 50.4, one each: 48.4).
    __device__ static __forceinline__ void comps2(const Lane &L, double vA, double vB, double *xs, Comps &cA, Comps &cB) {
        cB = comps_lds(L, vB, xs);
        cA = comps(vA);
    }
    /// (v0 + v1) + v2 over the three components of this lane's point: the cross-row sum
    /// (row 0 + row 1) + (row 2 + row 3) with the idle row 3 holding 0.  Lanes of row 3 contribute
    /// whatever they hold; they hold exactly 0 by construction: load_x0 / traj_load give them a zero
    /// state, and every quantity summed here is a product with a factor derived from that state by
    /// differences and products (d = x - prev(x) = 0, so d d = 0 and d lF = 0).  A mask here would cost
    /// two instructions on the dependent chain of every spring evaluation; instead
    /// tests/test_gpu_parity.py::test_idle_row_of_the_lane_layout_stays_zero reads the stored
    /// trajectories back and pins the invariant.
    __device__ static __forceinline__ double quad_sum(double v) {
        double a, b;
        row_partner<false>(v, a, b);
        v = a + b;
        row_partner<true>(v, a, b);
        return a + b;
    }
    /// Same component of the previous / next point: one DPP row shift (0 before point 0; the value
    /// after point 9 is never used).
    __device__ static __forceinline__ double prev_point(const Lane &, double v) { return row_from_prev(v); }
    __device__ static __forceinline__ double next_point(const Lane &, double v) { return row_from_next(v); }

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
        if (L.active)
            x[L.q] = s.x;
        if (L.mass)
            x[VOFF + L.q] = s.v;
    }
    /// Trajectory rows in the HBM workspace: {x, v} per lane, 1 KiB per stage.
    static constexpr int TRAJ_DOUBLES = 2 * WAVE;
    __device__ static __forceinline__ void traj_store(const Lane &L, double *row, const State &s) {
        reinterpret_cast<double2 *>(row)[L.l] = make_double2(s.x, s.v);
    }
    __device__ static __forceinline__ State traj_load(const Lane &L, const double *row) {
        double2 a = reinterpret_cast<const double2 *>(row)[L.l];
        return {a.x, a.y};
    }
    /// Tangent-state row of the Hv sweeps: one double per compact slot, the component the cost
    /// Hessian multiplies (x of the end point, v of the masses).
    __device__ static __forceinline__ void xt_store(const Lane &L, double *row, const State &s_t) {
        row[L.q] = L.end ? s_t.x : s_t.v; // idle lanes: slot 31
    }
    __device__ static __forceinline__ double xt_load(const Lane &L, const double *row) { return row[L.q]; }
    /// This lane's component of u_k from a decision vector in LDS.
    __device__ static __forceinline__ double load_input(const Lane &L, const double *u_k) {
        return u_k[L.c < DIM ? L.c : DIM - 1];
    }
    /// The end point's lanes own the adjoint of the stage input.
    __device__ static __forceinline__ void store_input_grad(const Lane &L, double *g_k, double g) {
        if (L.end)
            g_k[L.c] = g;
    }

    // ---- springs (primal) -------------------------------------------------------------------------
    struct Springs {
        double d, n2, r, c;
    };
    __device__ static __forceinline__ void springs(const Lane &L, const State &s, Springs &sp) {
        sp.d      = s.x - prev_point(L, s.x); // point 0 is anchored at the origin: x - 0
        double sq = sp.d * sp.d;
        sp.n2     = quad_sum(sq);
        sp.r      = sqrt_inrange(sp.n2); // = sqrt(n2), L / r: wave_math.hip.h
        sp.c      = Dspring * (1.0 - div_inrange(Lspring, sp.r));
    }
    /// The end point has no velocity state: its lanes carry the stage input there instead, so the
    /// time derivative of every position is just "the v slot" (with_input() once per stage; the
    /// slot's own derivative is 0 off the masses, so the input stays put through the Runge-Kutta
    /// combinations).
    __device__ static __forceinline__ State with_input(const Lane &L, const State &s, double uc) {
        return {s.x, L.end ? uc : s.v};
    }
    /// ds = f_c(s, u); s comes from with_input().
    __device__ static __forceinline__ State fc(const Lane &L, const State &s, Springs &sp) {
        springs(L, s, sp);
        double F  = sp.d * sp.c;
        double Fn = next_point(L, F); // force of the spring towards point p+1
        double a  = inv_m * (Fn - F);
        a         = L.grav ? a - g_grav : a;
        return {s.v, L.mass ? a : 0.0}; // a select, not inv_m_l: lanes beyond the chain hold 0 * inf
    }
    struct AdjAux {
        double lF, dt, w;
    };
    /// Adjoint of f_c: kbar = adjoint of ds, returns the adjoint of s; the adjoint of u is kbar.x
    /// on the end point's lanes.
    __device__ static __forceinline__ State fc_adj(const Lane &L, const Springs &sp, const State &kbar, AdjAux &ax) {
        double A   = prev_point(L, kbar.v);   // adjoint of a_{p-1} (0 for point 0)
        double B   = kbar.v;                  // adjoint of a_p (stays 0 off the masses: see the return)
        ax.lF      = inv_m * (A - B);
        double pr  = sp.d * ax.lF;
        ax.dt      = quad_sum(pr);
        double n2r = sp.n2 * sp.r;
        ax.w       = div_inrange(DL, n2r);
        double t   = ax.dt * ax.w;
        double ld  = (sp.c * ax.lF) + (t * sp.d);
        double ldn = next_point(L, ld);
        return {L.end ? ld : ld - ldn, L.mass ? kbar.x : 0.0};
    }

    /// One RK4 step of the plant (formulation.hpp:21-28).
    __device__ static __forceinline__ State rk4(const Lane &L, const State &s_in, double uc) {
        Springs sp;
        State s  = with_input(L, s_in, uc);
        State k1 = fc(L, s, sp);
        State k2 = fc(L, {s.x + 0.5 * (Ts * k1.x), s.v + 0.5 * (Ts * k1.v)}, sp);
        State k3 = fc(L, {s.x + 0.5 * (Ts * k2.x), s.v + 0.5 * (Ts * k2.v)}, sp);
        State k4 = fc(L, {s.x + Ts * k3.x, s.v + Ts * k3.v}, sp);
        return {s.x + c6 * (((k1.x + 2.0 * k2.x) + 2.0 * k3.x) + k4.x),
                s.v + c6 * (((k1.v + 2.0 * k2.v) + 2.0 * k3.v) + k4.v)};
    }

    // ---- the tape ------------------------------------------------------------------------------
    // Per substage j and spring p: the upper triangles of K = c I + w d d^T and of
    // M = w [ lF d^T + d lF^T + dt (I - (3/n2) d d^T) ] (both symmetric), six doubles each in the
    // order (0,0) (0,1) (0,2) (1,1) (1,2) (2,2): spring p's block at doubles [spring_base(p), +6) of a
    // 512-byte row.  Stage layout in HBM (4 KiB): [K rows of j = 0..3] (2 KiB: all the forward sweep
    // needs) then [M rows of j = 0..3].  Lane (p,c) reads row c of the full matrices from LDS.
    // The forward sweep of each product writes its tangent spring vectors d_t to a second tape
    // ("st": [j][32 slots] doubles, 1 KiB per stage) that its adjoint sweep streams back, and one
    // 256-byte row per stage with the tangent-state components the cost Hessian needs.
    static constexpr int TAPE_STAGE_BYTES  = 4096;
    static constexpr int ST_STAGE_BYTES    = 1024;
    static constexpr int LDS_STAGE_BYTES   = TAPE_STAGE_BYTES + ST_STAGE_BYTES;
    static constexpr int TAPE_D4_PER_STAGE = TAPE_STAGE_BYTES / 32;
    static constexpr int ST_D4_PER_STAGE   = ST_STAGE_BYTES / 32;
    static constexpr int XT_DOUBLES        = 32; ///< tangent-state row per stage (compact slots)

    __device__ static __forceinline__ void tape_store(const Lane &L, double4 *stage, int j, const Springs &sp,
                                                      const AdjAux &ax, double *xs) {
        const Comps dc = comps_lds(L, sp.d, xs), lc = comps_lds(L, ax.lF, xs);
        const double dq[DIM] = {dc.c0, dc.c1, dc.c2}, lq[DIM] = {lc.c0, lc.c1, lc.c2};
        const double e  = div_inrange(3.0, sp.n2);
        const double wd = ax.w * sp.d;
        const double g  = e * sp.d;
        // Every lane computes its three entries of the row and stores them - the lanes that own no entry
        // of the upper triangle (c > c2, idle lanes) into the row's unread last slot: stores under an exec
        // mask would put each entry's arithmetic into a basic block of its own (twelve per stage), which
        // the scheduler cannot overlap with the adjoint recursion.
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
    struct TapeK {
        double K[DIM];
    };
    struct TapeKM {
        double K[DIM], M[DIM], D[DIM]; ///< D = d_t of this lane's point (from the tangent tape)
    };
    __device__ static __forceinline__ TapeK tape_fwd(const Lane &L, const char *buf, int j) {
        const double *K = reinterpret_cast<const double *>(buf + j * 512) + spring_base(L.pc);
        return {{K[L.e[0]], K[L.e[1]], K[L.e[2]]}};
    }
    __device__ static __forceinline__ TapeKM tape_rev(const Lane &L, const char *buf, int j) {
        const double *K  = reinterpret_cast<const double *>(buf + j * 512) + spring_base(L.pc);
        const double *M  = reinterpret_cast<const double *>(buf + 2048 + j * 512) + spring_base(L.pc);
        const double *st = reinterpret_cast<const double *>(buf + TAPE_STAGE_BYTES + j * 256) + DIM * L.pc;
        return {{K[L.e[0]], K[L.e[1]], K[L.e[2]]}, {M[L.e[0]], M[L.e[1]], M[L.e[2]]}, {st[0], st[1], st[2]}};
    }

    // ---- tangent sweeps over the tape ---------------------------------------------------------------
    // Within a stage the four substage forces form two independent chains of two: the position a
    // substage evaluates f_c at depends on the force of the substage two before it (x_2 = x + h/2 v
    // needs none, x_3 needs a_1, x_4 needs a_2), and the same holds for the adjoint substages.  The
    // sweeps are written pair by pair - both neighbour differences, both gathers (comps2), both
    // products - so that one chain's LDS round trip and FMA latencies are filled by the other's
    // instructions.  Every value is computed by the expression of the substage-by-substage
    // form.
    /// Tangent of f_c at two positions: a = inv_m (F_{p+1} - F_p), F = K d_t.  Writes both d_t to the
    /// tangent tape (substages jA, jB).
    __device__ static __forceinline__ void fc_tan2(const Lane &L, const char *buf, int jA, int jB, double xA,
                                                   double xB, double4 *st_stage, double *xs, double &aA, double &aB) {
        double dA = xA - prev_point(L, xA);
        double dB = xB - prev_point(L, xB);
        reinterpret_cast<double *>(reinterpret_cast<char *>(st_stage) + jA * 256)[L.q] = dA; // idle lanes: slot 31
        reinterpret_cast<double *>(reinterpret_cast<char *>(st_stage) + jB * 256)[L.q] = dB;
        const TapeK gA = tape_fwd(L, buf, jA), gB = tape_fwd(L, buf, jB);
        Comps cA, cB;
        comps2(L, dA, dB, xs, cA, cB);
        double FA = gA.K[0] * cA.c0;
        double FB = gB.K[0] * cB.c0;
        FA        = __builtin_fma(gA.K[1], cA.c1, FA);
        FB        = __builtin_fma(gB.K[1], cB.c1, FB);
        FA        = __builtin_fma(gA.K[2], cA.c2, FA);
        FB        = __builtin_fma(gB.K[2], cB.c2, FB);
        aA        = L.inv_m_l * (next_point(L, FA) - FA);
        aB        = L.inv_m_l * (next_point(L, FB) - FB);
    }
    /// Tangent of the f_c adjoint for two substages: the position part ld_p - ld_{p+1} of
    /// ld_t = K lF_t + M d_t, lF_t = inv_m (kv_{p-1} - kv_p); kv = the velocity part of the substage's
    /// incoming adjoint (0 off the masses).
    __device__ static __forceinline__ void fc_adj_tan2(const Lane &L, const char *buf, int jA, int jB, double kvA,
                                                       double kvB, double *xs, double &oA, double &oB) {
        double lFA = inv_m * (prev_point(L, kvA) - kvA);
        double lFB = inv_m * (prev_point(L, kvB) - kvB);
        const TapeKM gA = tape_rev(L, buf, jA), gB = tape_rev(L, buf, jB);
        Comps cA, cB;
        comps2(L, lFA, lFB, xs, cA, cB);
        double ldA = gA.K[0] * cA.c0;
        double ldB = gB.K[0] * cB.c0;
        ldA        = __builtin_fma(gA.K[1], cA.c1, ldA);
        ldB        = __builtin_fma(gB.K[1], cB.c1, ldB);
        ldA        = __builtin_fma(gA.K[2], cA.c2, ldA);
        ldB        = __builtin_fma(gB.K[2], cB.c2, ldB);
        ldA        = __builtin_fma(gA.M[0], gA.D[0], ldA);
        ldB        = __builtin_fma(gB.M[0], gB.D[0], ldB);
        ldA        = __builtin_fma(gA.M[1], gA.D[1], ldA);
        ldB        = __builtin_fma(gB.M[1], gB.D[1], ldB);
        ldA        = __builtin_fma(gA.M[2], gA.D[2], ldA);
        ldB        = __builtin_fma(gB.M[2], gB.D[2], ldB);
        double nA = next_point(L, ldA), nB = next_point(L, ldB);
        // end ? ld : ld - n, without the select: one rounding either way; the lane after the end point holds a
        // finite value (clamped tape reads), so 0 * n is 0
        oA = __builtin_fma(-L.nend_l, nA, ldA);
        oB = __builtin_fma(-L.nend_l, nB, ldB);
    }
    /// Tangent of one RK4 step; the stage's K rows are in LDS at `buf`.
    __device__ static __forceinline__ State rk4_tan(const Lane &L, const char *buf, const State &s_in, double uc_t,
                                                    double4 *st_stage, double *xs) {
        const State s_t = with_input(L, s_in, uc_t);
        // k_j = {kx_j, a_j}: kx_1 = v, kx_{j+1} = v + c_j a_j; a_j = a(x + c_{j-1} kx_{j-1})
        double a1, a2, a3, a4;
        const double x2 = __builtin_fma(hTs, s_t.v, s_t.x);
        fc_tan2(L, buf, 0, 1, s_t.x, x2, st_stage, xs, a1, a2);
        const double kx2 = __builtin_fma(hTs, a1, s_t.v), kx3 = __builtin_fma(hTs, a2, s_t.v);
        const double x3 = __builtin_fma(hTs, kx2, s_t.x), x4 = __builtin_fma(Ts, kx3, s_t.x);
        fc_tan2(L, buf, 2, 3, x3, x4, st_stage, xs, a3, a4);
        const double kx4 = __builtin_fma(Ts, a3, s_t.v);
        double tx = __builtin_fma(2.0, kx2, s_t.v), tv = __builtin_fma(2.0, a2, a1);
        tx        = __builtin_fma(2.0, kx3, tx);
        tv        = __builtin_fma(2.0, a3, tv);
        tx        = tx + kx4;
        tv        = tv + a4;
        return {__builtin_fma(c6, tx, s_t.x), __builtin_fma(c6, tv, s_t.v)};
    }
    /// Tangent of the RK4 adjoint (second-order adjoint).  ubar_t is valid on the end point's lanes.
    __device__ static __forceinline__ void rk4_adj_tan(const Lane &L, const char *buf, const State &mu_t,
                                                       State &sbar_t, double &ubar_t, double *xs) {
        // incoming substage adjoints kb_j = {kbx_j, kbv_j}; the adjoint of f_c returns {G_j, m_j}:
        // G_j from kbv_j (springs), m_j = kbx_j on the masses (0 elsewhere: keeps kbv at 0 off the masses)
        const double sgx = c6 * mu_t.x, sgv = c6 * mu_t.v;
        const double sgx2 = 2.0 * sgx, sgv2 = 2.0 * sgv;
        // m = mass ? x : 0 as one multiplication (the x of these sweeps are finite; a zero's sign may differ): 66.3 against
        // 67.3 ms per pass of the bench batch with selects
        const double m4   = L.mass_l * sgx;
        const double kbv3 = __builtin_fma(Ts, m4, sgv2);
        double G4, G3, G2, G1;
        fc_adj_tan2(L, buf, 3, 2, sgv, kbv3, xs, G4, G3);
        const double kbx3 = __builtin_fma(Ts, G4, sgx2);
        const double m3   = L.mass_l * kbx3;
        const double kbv2 = __builtin_fma(hTs, m3, sgv2);
        const double kbx2 = __builtin_fma(hTs, G3, sgx2);
        const double m2   = L.mass_l * kbx2;
        const double kbv1 = __builtin_fma(hTs, m2, sgv);
        fc_adj_tan2(L, buf, 1, 0, kbv2, kbv1, xs, G2, G1);
        const double kbx1 = __builtin_fma(hTs, G2, sgx);
        const double m1   = L.mass_l * kbx1;
        sbar_t = {(((mu_t.x + G4) + G3) + G2) + G1, (((mu_t.v + m4) + m3) + m2) + m1};
        ubar_t = ((sgx + kbx3) + kbx2) + kbx1;
    }

    // ---- the tangent sweeps with both substage chains SIDE BY SIDE (round 5) ---------------------------------
    // A lone wavefront is bound by the instructions it issues (5-6 cycles each), and the
    // instructions above use 30 of their 64 lanes.  The two independent substage chains of a stage - forward
    // (1, 3) and (2, 4), adjoint (3, 1) and (4, 2) - are the same program on different data, so here the lower
    // half of the wavefront evaluates one chain and the upper half the other WITH THE SAME INSTRUCTIONS:
    //     lane = 32 chain + 10 component + point            (lanes 30, 31 of a half idle)
    // Within a half the ten points of a component are contiguous, so the neighbouring point is the neighbouring
    // lane of the wavefront (DPP wave_shr:1 / wave_shl:1); what lies beyond a component's ends - the anchor
    // before point 0, the lane after the end point - is masked by a lane constant inside an fma, which leaves
    // the arithmetic as it is (fma(-1, b, a) is a - b in one rounding; fma(-0, b, a) is a for finite b).  The
    // components of a point are 10 lanes apart: gathered through an LDS row (one write, three reads; a lone
    // chain has nothing to fill a gather's latency with, and ds_bpermute is no faster:
    // scripts/microbench/hv_pair_ab.hip).  Both halves carry the full tangent state; once per stage they
    // exchange their two accelerations (adjoint: their two G) with v_permlane32_swap and form the Runge-Kutta
    // combination by the expressions of rk4_tan / rk4_adj_tan above.  Same operations on the same operands:
    // same bits (the microbenchmark compares the two forms; the parity tests run this one).  Per stage
    // 528 + 750 cycles against 711 + 949 for the interleaved form on a lone wavefront; two WAVEFRONTS with one
    // chain each lose (730 + 1033: an exchange through LDS and s_barrier costs 200 cycles a stage).
    struct PairLane {
        int l, h, c, p, q; ///< lane, chain (half), component, point, compact slot 3p+c (idle lanes: 31)
        bool active, mass, end, hi;
        double inv_m_l, mass_l, nend_l, nfirst_l; ///< nfirst_l: 0 on point 0 (and idle lanes), 1 elsewhere
        int k_off;  ///< byte offset in a landing buffer of this lane's spring block in the K row of the chain's FIRST forward substage
        int e8[DIM]; ///< byte offsets of the three entries of row c inside a spring block
        int st_off; ///< byte offset in a tangent-tape stage of this lane's d_t slot, first forward substage (the second: + 512)
        int dt_off; ///< byte offset in a landing buffer of this point's d_t triple, first ADJOINT substage (the second: - 512)
        int xs_w, xs_r; ///< doubles: own slot / first component of the point in the exchange row of this half
        __device__ explicit PairLane(int lane) {
            l  = lane;
            h  = lane >> 5;
            hi = h != 0;
            const int r = lane & 31;
            active = r < DIM * NP;
            c      = active ? r / NP : 0;
            p      = active ? r % NP : NB;
            mass   = active && p < NB;
            end    = active && p == NB;
            inv_m_l  = mass ? inv_m : 0.0;
            mass_l   = mass ? 1.0 : 0.0;
            nend_l   = end ? 0.0 : 1.0;
            nfirst_l = (active && p > 0) ? 1.0 : 0.0;
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
    static constexpr int PAIR_XS_DOUBLES = 64; ///< exchange rows of the pair gather: 32 doubles per half
    /// st stores of a forward stage: one per substage PAIR (each stores both chains' d_t)
    static constexpr int PAIR_ST_STORES_PER_STAGE = 2;
    __device__ static __forceinline__ double lane_above(double v) { return dpp_move<0x130>(v); } // wave_shl:1
    __device__ static __forceinline__ double lds_at(const char *base, int off) {
        return *reinterpret_cast<const double *>(base + off);
    }
    __device__ static __forceinline__ Comps comps_pair(const PairLane &L, double v, double *xs) {
        xs[L.xs_w] = v;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double *s = xs + L.xs_r;
        return {s[0], s[1], s[2]};
    }
    __device__ static __forceinline__ double load_input(const PairLane &L, const double *u_k) { return u_k[L.c]; }
    __device__ static __forceinline__ void store_input_grad(const PairLane &L, double *g_k, double g) {
        if (L.end && !L.hi)
            g_k[L.c] = g;
    }
    /// (both halves store the same value to the same slot)
    __device__ static __forceinline__ void xt_store(const PairLane &L, double *row, const State &s_t) {
        row[L.q] = L.end ? s_t.x : s_t.v;
    }
    __device__ static __forceinline__ double xt_load(const PairLane &L, const double *row) { return row[L.q]; }
    __device__ static __forceinline__ double xt_lds(const PairLane &L, const char *buf) {
        return reinterpret_cast<const double *>(buf + LDS_STAGE_BYTES)[L.q];
    }
    __device__ static __forceinline__ void cost_grad_x_add_tan(const PairLane &L, double xt, State &lam_t) {
        lam_t.x = L.end ? __builtin_fma(c2a, xt, lam_t.x) : lam_t.x;
        lam_t.v = L.mass ? __builtin_fma(c2b, xt, lam_t.v) : lam_t.v;
    }
    /// fc_tan2's expressions, one substage of each chain: K at buf + koff (+ e8[.]), d_t to st_stage + st_off.
    __device__ static __forceinline__ double fc_tan_pair(const PairLane &L, const char *buf, int koff, int st_off, double x,
                                                         char *st_stage, double *xs) {
        const double d = __builtin_fma(-L.nfirst_l, lane_below(x), x); // x - prev (point 0: the anchor at the origin)
        *reinterpret_cast<double *>(st_stage + st_off) = d;            // idle lanes: slot 31
        const char *kb = buf + koff;
        const double K0 = lds_at(kb, L.e8[0]), K1 = lds_at(kb, L.e8[1]), K2 = lds_at(kb, L.e8[2]);
        const Comps c = comps_pair(L, d, xs);
        double F = K0 * c.c0;
        F        = __builtin_fma(K1, c.c1, F);
        F        = __builtin_fma(K2, c.c2, F);
        return L.inv_m_l * (lane_above(F) - F);
    }
    /// fc_adj_tan2's expressions, one substage of each chain.
    __device__ static __forceinline__ double fc_adj_tan_pair(const PairLane &L, const char *buf, int koff, int dt_off, double kv,
                                                             double *xs) {
        const double lF = inv_m * (L.nfirst_l * lane_below(kv) - kv);
        const char *kb = buf + koff;
        const double K0 = lds_at(kb, L.e8[0]), K1 = lds_at(kb, L.e8[1]), K2 = lds_at(kb, L.e8[2]);
        const double M0 = lds_at(kb, 2048 + L.e8[0]), M1 = lds_at(kb, 2048 + L.e8[1]), M2 = lds_at(kb, 2048 + L.e8[2]);
        const double D0 = lds_at(buf, dt_off), D1 = lds_at(buf, dt_off + 8), D2 = lds_at(buf, dt_off + 16);
        const Comps c = comps_pair(L, lF, xs);
        double ld = K0 * c.c0;
        ld        = __builtin_fma(K1, c.c1, ld);
        ld        = __builtin_fma(K2, c.c2, ld);
        ld        = __builtin_fma(M0, D0, ld);
        ld        = __builtin_fma(M1, D1, ld);
        ld        = __builtin_fma(M2, D2, ld);
        return __builtin_fma(-L.nend_l, lane_above(ld), ld);
    }
    /// rk4_tan with the chains side by side: the lower half evaluates substages (1, 3), the upper (2, 4).
    __device__ static __forceinline__ State rk4_tan(const PairLane &L, const char *buf, const State &s_in, double uc_t,
                                                    double4 *st_stage, double *xs) {
        char *st = reinterpret_cast<char *>(st_stage);
        const double x = s_in.x, v = L.end ? uc_t : s_in.v;
        const double x2  = __builtin_fma(hTs, v, x);
        const double af  = fc_tan_pair(L, buf, L.k_off, L.st_off, L.hi ? x2 : x, st, xs);             // a1 | a2
        const double kxf = __builtin_fma(hTs, af, v);                                                 // kx2 | kx3
        const double xs2 = __builtin_fma(L.hi ? Ts : hTs, kxf, x);                                    // x3 | x4
        const double as  = fc_tan_pair(L, buf, L.k_off + 1024, L.st_off + 512, xs2, st, xs);          // a3 | a4
        double a1, a2, a3, a4;
        row_partner<true>(af, a1, a2); // the lower half's and the upper half's value, on every lane
        row_partner<true>(as, a3, a4);
        const double kx2 = __builtin_fma(hTs, a1, v), kx3 = __builtin_fma(hTs, a2, v);
        const double kx4 = __builtin_fma(Ts, a3, v);
        double tx = __builtin_fma(2.0, kx2, v), tv = __builtin_fma(2.0, a2, a1);
        tx        = __builtin_fma(2.0, kx3, tx);
        tv        = __builtin_fma(2.0, a3, tv);
        tx        = tx + kx4;
        tv        = tv + a4;
        return {__builtin_fma(c6, tx, x), __builtin_fma(c6, tv, v)};
    }
    /// rk4_adj_tan with the chains side by side: the lower half evaluates substages (3, 1), the upper (4, 2) - each
    /// half meets the d_t it stored in the forward sweep.
    __device__ static __forceinline__ void rk4_adj_tan(const PairLane &L, const char *buf, const State &mu_t, State &sbar_t,
                                                       double &ubar_t, double *xs) {
        const double sgx = c6 * mu_t.x, sgv = c6 * mu_t.v;
        const double sgx2 = 2.0 * sgx, sgv2 = 2.0 * sgv;
        const double m4   = L.mass_l * sgx;
        const double kbv3 = __builtin_fma(Ts, m4, sgv2);
        const double Gf   = fc_adj_tan_pair(L, buf, L.k_off + 1024, L.dt_off, L.hi ? sgv : kbv3, xs); // G3 | G4
        const double kbx  = __builtin_fma(L.hi ? Ts : hTs, Gf, sgx2);                                 // kbx2 | kbx3
        const double mm   = L.mass_l * kbx;
        const double kvs  = __builtin_fma(hTs, mm, L.hi ? sgv2 : sgv);                                // kbv1 | kbv2
        const double Gs   = fc_adj_tan_pair(L, buf, L.k_off, L.dt_off - 512, kvs, xs);                // G1 | G2
        double G3, G4, G1, G2;
        row_partner<true>(Gf, G3, G4);
        row_partner<true>(Gs, G1, G2);
        const double kbx3 = __builtin_fma(Ts, G4, sgx2), kbx2 = __builtin_fma(hTs, G3, sgx2);
        const double kbx1 = __builtin_fma(hTs, G2, sgx);
        const double m3 = L.mass_l * kbx3, m2 = L.mass_l * kbx2, m1 = L.mass_l * kbx1;
        sbar_t = {(((mu_t.x + G4) + G3) + G2) + G1, (((mu_t.v + m4) + m3) + m2) + m1};
        ubar_t = ((sgx + kbx3) + kbx2) + kbx1;
    }

    /// Adjoint of one RK4 step.  mu = adjoint of x_next; sbar = adjoint of s; ubar (valid on the
    /// end point's lanes) = adjoint of u through the dynamics.  With tape_stage != nullptr the
    /// K and M rows of the four substages are recorded.
    __device__ static __forceinline__ void rk4_adj(const Lane &L, const State &s_in, double uc, const State &mu,
                                                   State &sbar, double &ubar, double4 *tape_stage = nullptr,
                                                   double *xs = nullptr) {
        Springs sp1, sp2, sp3, sp4;
        AdjAux ax;
        State s = with_input(L, s_in, uc);
        State k = fc(L, s, sp1);
        k       = fc(L, {s.x + 0.5 * (Ts * k.x), s.v + 0.5 * (Ts * k.v)}, sp2);
        k       = fc(L, {s.x + 0.5 * (Ts * k.x), s.v + 0.5 * (Ts * k.v)}, sp3);
        springs(L, {s.x + Ts * k.x, s.v + Ts * k.v}, sp4);

        double sgx = c6 * mu.x, sgv = c6 * mu.v;
        State kb1{sgx, sgv}, kb2{2.0 * sgx, 2.0 * sgv}, kb3{2.0 * sgx, 2.0 * sgv}, kb4{sgx, sgv};
        sbar    = mu;
        State a = fc_adj(L, sp4, kb4, ax);
        if (tape_stage)
            tape_store(L, tape_stage, 3, sp4, ax, xs);
        sbar = {sbar.x + a.x, sbar.v + a.v};
        kb3  = {kb3.x + Ts * a.x, kb3.v + Ts * a.v};
        a    = fc_adj(L, sp3, kb3, ax);
        if (tape_stage)
            tape_store(L, tape_stage, 2, sp3, ax, xs);
        sbar = {sbar.x + a.x, sbar.v + a.v};
        kb2  = {kb2.x + 0.5 * (Ts * a.x), kb2.v + 0.5 * (Ts * a.v)};
        a    = fc_adj(L, sp2, kb2, ax);
        if (tape_stage)
            tape_store(L, tape_stage, 1, sp2, ax, xs);
        sbar = {sbar.x + a.x, sbar.v + a.v};
        kb1  = {kb1.x + 0.5 * (Ts * a.x), kb1.v + 0.5 * (Ts * a.v)};
        a    = fc_adj(L, sp1, kb1, ax);
        if (tape_stage)
            tape_store(L, tape_stage, 0, sp1, ax, xs);
        sbar = {sbar.x + a.x, sbar.v + a.v};
        ubar = ((kb4.x + kb3.x) + kb2.x) + kb1.x;
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
    /// Total cost: per point ((((x0 + x1) + x2) + v0) + v1) + v2 on the lane of its component 0
    /// (row 0, lane p), then the 64-lane butterfly: its two cross-row levels add the zeros of rows
    /// 1..3, its four in-row levels pair p with p^8, p^4, p^2, p^1 - the additions, in the order, of
    /// the oracle's butterfly with point p at lane 4p.
    __device__ static __forceinline__ double cost_total(const Lane &L, const State &acc) {
        const Comps x = comps(acc.x), v = comps(acc.v);
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
