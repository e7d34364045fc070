// A/B: the hanging chain's Hessian-vector product sweeps on ONE wavefront (V0: csrc/chain_model.hip.h, the two
// independent substage chains of a stage interleaved on one instruction stream) against TWO wavefronts of one
// workgroup, one chain each (V2: round-4 verdict, item 2).
//   forward:  a_3 needs only a_1, a_4 only a_2   -> wavefront 0 evaluates substages (1, 3), wavefront 1 (2, 4)
//   adjoint:  G_2 needs only G_4, G_1 only G_3   -> wavefront 0 evaluates (3, 1), wavefront 1 (4, 2): each wavefront
//             meets again the tangent spring vectors d_t it wrote itself in the forward sweep
// Both wavefronts carry the full tangent state / adjoint; once per stage they exchange their two accelerations
// (adjoint: their two G) through LDS - one 16-byte write, `s_waitcnt lgkmcnt(0); s_barrier`, one 16-byte read, the
// exchange rows double-buffered by stage parity so that one barrier per stage is enough - and each forms the Runge-
// Kutta combination by the expressions of rk4_tan / rk4_adj_tan: same operations on the same operands, same bits.
// Same harness as hv_layout_ab.hip: eight stages of synthetic K / M / d_t rows in LDS used round-robin, no LDS-DMA.
// GATHER: 0 = component gather through the permlane swaps, 1 = through the LDS exchange row (a lone chain has no
// second chain to fill a gather's latency with, so the shorter one should win).
// Prints shader cycles per stage and checks bit equality of the final tangent state, every stored d_t, Hv and the
// final adjoint.   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -o hv_pair_ab hv_pair_ab.hip
#include "chain_row_layout.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace pantr;
using M = ChainRowModel;

constexpr int NST   = 60;
constexpr int RING  = 8;
constexpr int STAGE = M::LDS_STAGE_BYTES;
constexpr int NV    = 192;

struct Out {
    double x[32], v[32], lx[32], lv[32], hv[NV];
    long long cyc_fwd, cyc_adj;
};

__device__ __forceinline__ void fill_lds(char *ring, double *vin, const double *g_tape, const double *g_v, int nthreads) {
    for (int i = threadIdx.x; i < RING * STAGE / 8; i += nthreads)
        reinterpret_cast<double *>(ring)[i] = g_tape[i];
    for (int i = threadIdx.x; i < NV; i += nthreads)
        vin[i] = g_v[i];
    __syncthreads();
}

// ---- V0: the product's sweeps on one wavefront (as hv_layout_ab.hip) -------------------------------------------
__global__ __launch_bounds__(64) void v0_kernel(const double *g_tape, const double *g_v, double *g_st, Out *out, int reps) {
    __shared__ __attribute__((aligned(16))) char ring[RING * STAGE];
    __shared__ double vin[NV], hv[NV], xs[32];
    const M::Lane L(threadIdx.x);
    fill_lds(ring, vin, g_tape, g_v, 64);
    M::State s_t{0, 0}, lam_t{0, 0};
    long long cf = 0, ca = 0;
    for (int r = 0; r < reps; ++r) {
        s_t = {0.0, 0.0};
        long long t0 = clock64();
        for (int k = 0; k < NST; ++k) {
            const double uc_t = M::load_input(L, vin + M::nu * k);
            s_t = M::rk4_tan(L, ring + (k % RING) * STAGE, s_t, uc_t,
                             reinterpret_cast<double4 *>(g_st) + (size_t)k * M::ST_D4_PER_STAGE, xs);
        }
        wave_sync();
        long long t1 = clock64();
        lam_t = {0.0, 0.0};
        M::cost_grad_x_add_tan(L, L.end ? s_t.x : s_t.v, lam_t);
        for (int k = NST - 1; k >= 0; --k) {
            const double uc_t = M::load_input(L, vin + M::nu * k);
            M::State sbar_t;
            double ubar_t;
            M::rk4_adj_tan(L, ring + (k % RING) * STAGE, lam_t, sbar_t, ubar_t, xs);
            M::store_input_grad(L, hv + M::nu * k, __builtin_fma(M::c2g, uc_t, ubar_t));
            lam_t = sbar_t;
            M::cost_grad_x_add_tan(L, uc_t, lam_t);
        }
        wave_sync();
        long long t2 = clock64();
        cf += t1 - t0;
        ca += t2 - t1;
    }
    if (L.active) {
        out->x[L.q] = s_t.x; out->v[L.q] = s_t.v;
        out->lx[L.q] = lam_t.x; out->lv[L.q] = lam_t.v;
    }
    for (int i = threadIdx.x; i < M::nu * NST; i += WAVE)
        out->hv[i] = hv[i];
    if (threadIdx.x == 0) {
        out->cyc_fwd = cf;
        out->cyc_adj = ca;
    }
}

// ---- V2: one chain per wavefront ---------------------------------------------------------------------------------
template <int GATHER>
__device__ __forceinline__ M::Comps gather(const M::Lane &L, double v, double *xs) {
    if constexpr (GATHER == 1)
        return M::comps_lds(L, v, xs);
    else
        return M::comps(v);
}
/// fc_tan2's expressions for one substage.
template <int GATHER>
__device__ __forceinline__ double fc_tan1(const M::Lane &L, const char *buf, int j, double x, double4 *st_stage, double *xs) {
    const double d = x - M::prev_point(L, x);
    reinterpret_cast<double *>(reinterpret_cast<char *>(st_stage) + j * 256)[L.q] = d;
    const M::TapeK g = M::tape_fwd(L, buf, j);
    const M::Comps c = gather<GATHER>(L, d, xs);
    double F = g.K[0] * c.c0;
    F        = __builtin_fma(g.K[1], c.c1, F);
    F        = __builtin_fma(g.K[2], c.c2, F);
    return L.inv_m_l * (M::next_point(L, F) - F);
}
/// fc_adj_tan2's expressions for one substage.
template <int GATHER>
__device__ __forceinline__ double fc_adj_tan1(const M::Lane &L, const char *buf, int j, double kv, double *xs) {
    const double lF = M::inv_m * (M::prev_point(L, kv) - kv);
    const M::TapeKM g = M::tape_rev(L, buf, j);
    const M::Comps c = gather<GATHER>(L, lF, xs);
    double ld = g.K[0] * c.c0;
    ld        = __builtin_fma(g.K[1], c.c1, ld);
    ld        = __builtin_fma(g.K[2], c.c2, ld);
    ld        = __builtin_fma(g.M[0], g.D[0], ld);
    ld        = __builtin_fma(g.M[1], g.D[1], ld);
    ld        = __builtin_fma(g.M[2], g.D[2], ld);
    const double n = M::next_point(L, ld);
    return __builtin_fma(-L.nend_l, n, ld);
}
/// Publish (a, b) to the partner wavefront and take its pair: one barrier; `row`: this stage parity's two rows.
__device__ __forceinline__ void swap_with_partner(double2 *row, int w, int lane, double a, double b, double &oa, double &ob) {
    row[w * WAVE + lane] = make_double2(a, b);
    // the write has executed in LDS when lgkmcnt reaches 0; no vmcnt wait here (the d_t stores stay in flight)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const double2 o = row[(1 - w) * WAVE + lane];
    oa = o.x;
    ob = o.y;
}

template <int GATHER>
__global__ __launch_bounds__(128) void v2_kernel(const double *g_tape, const double *g_v, double *g_st, Out *out, int reps) {
    __shared__ __attribute__((aligned(16))) char ring[RING * STAGE];
    __shared__ double vin[NV], hv[NV], xs2[2][32];
    __shared__ __attribute__((aligned(16))) double2 ex[2][2 * WAVE];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const M::Lane L(lane);
    double *xs = xs2[w];
    fill_lds(ring, vin, g_tape, g_v, 128);
    // which substages this wavefront evaluates, and the coefficients that go with them (scalars)
    const int jf_fwd = w, js_fwd = 2 + w;         // forward: (1, 3) | (2, 4)  [0-based j]
    const int jf_adj = 2 + w, js_adj = w;         // adjoint: (3, 1) | (4, 2)
    const double cx_first = w ? M::hTs : 0.0, cx_second = w ? M::Ts : M::hTs;
    M::State s_t{0, 0}, lam_t{0, 0};
    long long cf = 0, ca = 0;
    for (int r = 0; r < reps; ++r) {
        s_t = {0.0, 0.0};
        __syncthreads();
        long long t0 = clock64();
        for (int k = 0; k < NST; ++k) {
            const char *buf = ring + (k % RING) * STAGE;
            double4 *st_k   = reinterpret_cast<double4 *>(g_st) + (size_t)k * M::ST_D4_PER_STAGE;
            const double uc_t = M::load_input(L, vin + M::nu * k);
            const M::State s = M::with_input(L, s_t, uc_t);
            const double xf  = __builtin_fma(cx_first, s.v, s.x); // w = 0: x (0 * v + x is x: finite sweeps); w = 1: x2
            const double af  = fc_tan1<GATHER>(L, buf, jf_fwd, w ? xf : s.x, st_k, xs);
            const double kxf = __builtin_fma(M::hTs, af, s.v);
            const double xsd = __builtin_fma(cx_second, kxf, s.x);
            const double as  = fc_tan1<GATHER>(L, buf, js_fwd, xsd, st_k, xs);
            double of, os;
            swap_with_partner(ex[k & 1], w, lane, af, as, of, os);
            const double a1 = w ? of : af, a2 = w ? af : of, a3 = w ? os : as, a4 = w ? as : os;
            const double kx2 = __builtin_fma(M::hTs, a1, s.v), kx3 = __builtin_fma(M::hTs, a2, s.v);
            const double kx4 = __builtin_fma(M::Ts, a3, s.v);
            double tx = __builtin_fma(2.0, kx2, s.v), tv = __builtin_fma(2.0, a2, a1);
            tx        = __builtin_fma(2.0, kx3, tx);
            tv        = __builtin_fma(2.0, a3, tv);
            tx        = tx + kx4;
            tv        = tv + a4;
            s_t       = {__builtin_fma(M::c6, tx, s.x), __builtin_fma(M::c6, tv, s.v)};
        }
        __syncthreads();
        long long t1 = clock64();
        lam_t = {0.0, 0.0};
        M::cost_grad_x_add_tan(L, L.end ? s_t.x : s_t.v, lam_t);
        for (int k = NST - 1; k >= 0; --k) {
            const char *buf   = ring + (k % RING) * STAGE;
            const double uc_t = M::load_input(L, vin + M::nu * k);
            const double sgx = M::c6 * lam_t.x, sgv = M::c6 * lam_t.v;
            const double sgx2 = 2.0 * sgx, sgv2 = 2.0 * sgv;
            const double m4   = L.mass_l * sgx;
            const double kbv3 = __builtin_fma(M::Ts, m4, sgv2);
            const double Gf   = fc_adj_tan1<GATHER>(L, buf, jf_adj, w ? sgv : kbv3, xs);       // w = 0: G3, w = 1: G4
            const double kbx  = __builtin_fma(w ? M::Ts : M::hTs, Gf, sgx2);                     // w = 0: kbx2, w = 1: kbx3
            const double mm   = L.mass_l * kbx;
            const double kvs  = __builtin_fma(M::hTs, mm, w ? sgv2 : sgv);                       // w = 0: kbv1, w = 1: kbv2
            const double Gs   = fc_adj_tan1<GATHER>(L, buf, js_adj, kvs, xs);                    // w = 0: G1, w = 1: G2
            double of, os;
            swap_with_partner(ex[k & 1], w, lane, Gf, Gs, of, os);
            const double G4 = w ? Gf : of, G3 = w ? of : Gf, G2 = w ? Gs : os, G1 = w ? os : Gs;
            const double kbx3 = __builtin_fma(M::Ts, G4, sgx2), kbx2 = __builtin_fma(M::hTs, G3, sgx2);
            const double kbx1 = __builtin_fma(M::hTs, G2, sgx);
            const double m3 = L.mass_l * kbx3, m2 = L.mass_l * kbx2, m1 = L.mass_l * kbx1;
            const M::State sbar_t{(((lam_t.x + G4) + G3) + G2) + G1, (((lam_t.v + m4) + m3) + m2) + m1};
            const double ubar_t = ((sgx + kbx3) + kbx2) + kbx1;
            if (w == 0)
                M::store_input_grad(L, hv + M::nu * k, __builtin_fma(M::c2g, uc_t, ubar_t));
            lam_t = sbar_t;
            M::cost_grad_x_add_tan(L, uc_t, lam_t);
        }
        __syncthreads();
        long long t2 = clock64();
        cf += t1 - t0;
        ca += t2 - t1;
    }
    if (w == 0) {
        if (L.active) {
            out->x[L.q] = s_t.x; out->v[L.q] = s_t.v;
            out->lx[L.q] = lam_t.x; out->lv[L.q] = lam_t.v;
        }
        for (int i = lane; i < M::nu * NST; i += WAVE)
            out->hv[i] = hv[i];
        if (lane == 0) {
            out->cyc_fwd = cf;
            out->cyc_adj = ca;
        }
    }
}


// ---- V3: V2 with the stage's LDS reads (tape rows, stage input) issued a stage ahead -----------------------------
// The barrier of the exchange is a scheduling fence for the compiler: in V2 every stage starts by waiting for its
// own reads of the stage input and of the K (adjoint: K, M, d_t) entries.  Here they are read before the exchange
// of the stage before, into registers.
struct KRow { double K[3]; };
struct KMRow { double K[3], M[3], D[3]; };
__device__ __forceinline__ double fc_tan1_pre(const M::Lane &L, const M::TapeK &g, int j, double x, double4 *st_stage) {
    const double d = x - M::prev_point(L, x);
    reinterpret_cast<double *>(reinterpret_cast<char *>(st_stage) + j * 256)[L.q] = d;
    const M::Comps c = M::comps(d);
    double F = g.K[0] * c.c0;
    F        = __builtin_fma(g.K[1], c.c1, F);
    F        = __builtin_fma(g.K[2], c.c2, F);
    return L.inv_m_l * (M::next_point(L, F) - F);
}
__device__ __forceinline__ double fc_adj_tan1_pre(const M::Lane &L, const M::TapeKM &g, double kv) {
    const double lF = M::inv_m * (M::prev_point(L, kv) - kv);
    const M::Comps c = M::comps(lF);
    double ld = g.K[0] * c.c0;
    ld        = __builtin_fma(g.K[1], c.c1, ld);
    ld        = __builtin_fma(g.K[2], c.c2, ld);
    ld        = __builtin_fma(g.M[0], g.D[0], ld);
    ld        = __builtin_fma(g.M[1], g.D[1], ld);
    ld        = __builtin_fma(g.M[2], g.D[2], ld);
    const double n = M::next_point(L, ld);
    return __builtin_fma(-L.nend_l, n, ld);
}
__global__ __launch_bounds__(128) void v3_kernel(const double *g_tape, const double *g_v, double *g_st, Out *out, int reps) {
    __shared__ __attribute__((aligned(16))) char ring[RING * STAGE];
    __shared__ double vin[NV + 8], hv[NV];
    __shared__ __attribute__((aligned(16))) double2 ex[2][2 * WAVE];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const M::Lane L(lane);
    fill_lds(ring, vin, g_tape, g_v, 128);
    const int jf_fwd = w, js_fwd = 2 + w, jf_adj = 2 + w, js_adj = w;
    const double cx_first = w ? M::hTs : 0.0, cx_second = w ? M::Ts : M::hTs;
    M::State s_t{0, 0}, lam_t{0, 0};
    long long cf = 0, ca = 0;
    for (int r = 0; r < reps; ++r) {
        s_t = {0.0, 0.0};
        __syncthreads();
        long long t0 = clock64();
        M::TapeK gf = M::tape_fwd(L, ring, jf_fwd), gs = M::tape_fwd(L, ring, js_fwd);
        double uc_t = M::load_input(L, vin);
        for (int k = 0; k < NST; ++k) {
            double4 *st_k   = reinterpret_cast<double4 *>(g_st) + (size_t)k * M::ST_D4_PER_STAGE;
            const M::State s = M::with_input(L, s_t, uc_t);
            const double xf  = __builtin_fma(cx_first, s.v, s.x);
            const double af  = fc_tan1_pre(L, gf, jf_fwd, w ? xf : s.x, st_k);
            const double kxf = __builtin_fma(M::hTs, af, s.v);
            const double xsd = __builtin_fma(cx_second, kxf, s.x);
            const double as  = fc_tan1_pre(L, gs, js_fwd, xsd, st_k);
            // the next stage's reads, ahead of the exchange (stage NST reads a valid ring slot and vin's padding)
            const char *nbuf = ring + ((k + 1) % RING) * STAGE;
            gf   = M::tape_fwd(L, nbuf, jf_fwd);
            gs   = M::tape_fwd(L, nbuf, js_fwd);
            uc_t = M::load_input(L, vin + M::nu * (k + 1 < NST ? k + 1 : k));
            double of, os;
            swap_with_partner(ex[k & 1], w, lane, af, as, of, os);
            const double a1 = w ? of : af, a2 = w ? af : of, a3 = w ? os : as, a4 = w ? as : os;
            const double kx2 = __builtin_fma(M::hTs, a1, s.v), kx3 = __builtin_fma(M::hTs, a2, s.v);
            const double kx4 = __builtin_fma(M::Ts, a3, s.v);
            double tx = __builtin_fma(2.0, kx2, s.v), tv = __builtin_fma(2.0, a2, a1);
            tx        = __builtin_fma(2.0, kx3, tx);
            tv        = __builtin_fma(2.0, a3, tv);
            tx        = tx + kx4;
            tv        = tv + a4;
            s_t       = {__builtin_fma(M::c6, tx, s.x), __builtin_fma(M::c6, tv, s.v)};
        }
        __syncthreads();
        long long t1 = clock64();
        lam_t = {0.0, 0.0};
        M::cost_grad_x_add_tan(L, L.end ? s_t.x : s_t.v, lam_t);
        M::TapeKM hf = M::tape_rev(L, ring + ((NST - 1) % RING) * STAGE, jf_adj),
                  hs = M::tape_rev(L, ring + ((NST - 1) % RING) * STAGE, js_adj);
        uc_t = M::load_input(L, vin + M::nu * (NST - 1));
        for (int k = NST - 1; k >= 0; --k) {
            const double uc_k = uc_t;
            const double sgx = M::c6 * lam_t.x, sgv = M::c6 * lam_t.v;
            const double sgx2 = 2.0 * sgx, sgv2 = 2.0 * sgv;
            const double m4   = L.mass_l * sgx;
            const double kbv3 = __builtin_fma(M::Ts, m4, sgv2);
            const double Gf   = fc_adj_tan1_pre(L, hf, w ? sgv : kbv3);
            const double kbx  = __builtin_fma(w ? M::Ts : M::hTs, Gf, sgx2);
            const double mm   = L.mass_l * kbx;
            const double kvs  = __builtin_fma(M::hTs, mm, w ? sgv2 : sgv);
            const double Gs   = fc_adj_tan1_pre(L, hs, kvs);
            const char *nbuf = ring + ((k > 0 ? k - 1 : 0) % RING) * STAGE;
            hf   = M::tape_rev(L, nbuf, jf_adj);
            hs   = M::tape_rev(L, nbuf, js_adj);
            uc_t = M::load_input(L, vin + M::nu * (k > 0 ? k - 1 : 0));
            double of, os;
            swap_with_partner(ex[k & 1], w, lane, Gf, Gs, of, os);
            const double G4 = w ? Gf : of, G3 = w ? of : Gf, G2 = w ? Gs : os, G1 = w ? os : Gs;
            const double kbx3 = __builtin_fma(M::Ts, G4, sgx2), kbx2 = __builtin_fma(M::hTs, G3, sgx2);
            const double kbx1 = __builtin_fma(M::hTs, G2, sgx);
            const double m3 = L.mass_l * kbx3, m2 = L.mass_l * kbx2, m1 = L.mass_l * kbx1;
            const M::State sbar_t{(((lam_t.x + G4) + G3) + G2) + G1, (((lam_t.v + m4) + m3) + m2) + m1};
            const double ubar_t = ((sgx + kbx3) + kbx2) + kbx1;
            if (w == 0)
                M::store_input_grad(L, hv + M::nu * k, __builtin_fma(M::c2g, uc_k, ubar_t));
            lam_t = sbar_t;
            M::cost_grad_x_add_tan(L, uc_k, lam_t);
        }
        __syncthreads();
        long long t2 = clock64();
        cf += t1 - t0;
        ca += t2 - t1;
    }
    if (w == 0) {
        if (L.active) {
            out->x[L.q] = s_t.x; out->v[L.q] = s_t.v;
            out->lx[L.q] = lam_t.x; out->lv[L.q] = lam_t.v;
        }
        for (int i = lane; i < M::nu * NST; i += WAVE)
            out->hv[i] = hv[i];
        if (lane == 0) {
            out->cyc_fwd = cf;
            out->cyc_adj = ca;
        }
    }
}

// ---- V4: both chains in ONE wavefront, side by side: lane = 32 chain + 10 component + point ------------------------
// A lone wavefront is bound by the instructions it issues (5-6 cycles each), and V0's instructions use 30 of their
// 64 lanes.  Here the lower half of the wavefront evaluates substages (1, 3) [adjoint: (3, 1)] and the upper half
// (2, 4) [(4, 2)] with the SAME instructions: one f_c tangent serves both chains.  Within a half the three
// components of the ten points are contiguous (lanes 10 c + p; lanes 30, 31 idle), so the neighbouring point is
// the neighbouring lane of the whole wavefront (DPP wave_shr:1 / wave_shl:1; the anchor before point 0 and the
// lane after the end point are masked by lane constants inside an fma, as chain_model.hip.h does for the end
// point), the components of a point are 10 lanes apart (gathered through an LDS row: one write, three reads), and the
// halves exchange their two accelerations once per stage with v_permlane32_swap.  Same expressions, same bits.
struct PairLane {
    int l, h, c, p, q;
    bool active, mass, end;
    double inv_m_l, mass_l, nend_l, nfirst_l;
    int kf[3], ks[3];   ///< byte offsets in a landing buffer of this lane's three K entries, first / second substage (forward)
    int af[3], as[3];   ///< the same for the adjoint sweep's first / second substage (K; M is 2048 bytes further)
    int st_f, st_s;     ///< byte offsets in a tangent-tape stage of this lane's d_t slot, forward first / second substage
    int dt_f, dt_s;     ///< byte offsets in a landing buffer of this lane's point's d_t (three doubles), adjoint first / second
    int xs_w, xs_r;     ///< double offsets in the exchange rows: own slot, first component of the point
    __device__ explicit PairLane(int lane) {
        l = lane; h = lane >> 5;
        const int r = lane & 31;
        active = r < 30;
        c = active ? r / 10 : 0;
        p = active ? r % 10 : 9;
        mass = active && p < M::NB;
        end  = active && p == M::NB;
        inv_m_l = mass ? M::inv_m : 0.0;
        mass_l  = mass ? 1.0 : 0.0;
        nend_l  = end ? 0.0 : 1.0;
        nfirst_l = (active && p > 0) ? 1.0 : 0.0;
        q = active ? 3 * p + c : 31;
        const int jff = h, jfs = 2 + h, jaf = 2 + h, jas = h;
        for (int k = 0; k < 3; ++k) {
            const int ent = (M::spring_base(p) + M::sym(c, k)) * 8;
            kf[k] = jff * 512 + ent; ks[k] = jfs * 512 + ent;
            af[k] = jaf * 512 + ent; as[k] = jas * 512 + ent;
        }
        st_f = jff * 256 + q * 8; st_s = jfs * 256 + q * 8;
        dt_f = M::TAPE_STAGE_BYTES + jaf * 256 + 3 * p * 8; dt_s = M::TAPE_STAGE_BYTES + jas * 256 + 3 * p * 8;
        xs_w = 32 * h + q; xs_r = 32 * h + 3 * p;
    }
};
__device__ __forceinline__ double lane_above(double v) { return dpp_move<0x130>(v); } // wave_shl:1
__device__ __forceinline__ double ld_lds(const char *base, int off) { return *reinterpret_cast<const double *>(base + off); }
__device__ __forceinline__ M::Comps gather_pair(const PairLane &L, double v, double *xs) {
    xs[L.xs_w] = v;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double *s = xs + L.xs_r;
    return {s[0], s[1], s[2]};
}
__device__ __forceinline__ double fc_tan_pair(const PairLane &L, const char *buf, const int (&ko)[3], int st_off, double x,
                                              char *st_stage, double *xs) {
    const double d = __builtin_fma(-L.nfirst_l, lane_below(x), x); // x - prev (point 0: x - 0)
    *reinterpret_cast<double *>(st_stage + st_off) = d;
    const double K0 = ld_lds(buf, ko[0]), K1 = ld_lds(buf, ko[1]), K2 = ld_lds(buf, ko[2]);
    const M::Comps c = gather_pair(L, d, xs);
    double F = K0 * c.c0;
    F        = __builtin_fma(K1, c.c1, F);
    F        = __builtin_fma(K2, c.c2, F);
    return L.inv_m_l * (lane_above(F) - F);
}
__device__ __forceinline__ double fc_adj_tan_pair(const PairLane &L, const char *buf, const int (&ko)[3], int dt_off, double kv,
                                                  double *xs) {
    const double lF = M::inv_m * (L.nfirst_l * lane_below(kv) - kv);
    const double K0 = ld_lds(buf, ko[0]), K1 = ld_lds(buf, ko[1]), K2 = ld_lds(buf, ko[2]);
    const double M0 = ld_lds(buf, 2048 + ko[0]), M1 = ld_lds(buf, 2048 + ko[1]), M2 = ld_lds(buf, 2048 + ko[2]);
    const double D0 = ld_lds(buf, dt_off), D1 = ld_lds(buf, dt_off + 8), D2 = ld_lds(buf, dt_off + 16);
    const M::Comps c = gather_pair(L, lF, xs);
    double ld = K0 * c.c0;
    ld        = __builtin_fma(K1, c.c1, ld);
    ld        = __builtin_fma(K2, c.c2, ld);
    ld        = __builtin_fma(M0, D0, ld);
    ld        = __builtin_fma(M1, D1, ld);
    ld        = __builtin_fma(M2, D2, ld);
    return __builtin_fma(-L.nend_l, lane_above(ld), ld);
}
__global__ __launch_bounds__(64) void v4_kernel(const double *g_tape, const double *g_v, double *g_st, Out *out, int reps) {
    __shared__ __attribute__((aligned(16))) char ring[RING * STAGE];
    __shared__ double vin[NV], hv[NV], xs[64];
    const PairLane L(threadIdx.x);
    fill_lds(ring, vin, g_tape, g_v, 64);
    const bool hi = L.h != 0;
    double sx = 0, sv = 0, lx = 0, lv = 0;
    long long cf = 0, ca = 0;
    for (int r = 0; r < reps; ++r) {
        sx = sv = 0.0;
        long long t0 = clock64();
        for (int k = 0; k < NST; ++k) {
            const char *buf = ring + (k % RING) * STAGE;
            char *st_k      = reinterpret_cast<char *>(g_st) + (size_t)k * M::ST_STAGE_BYTES;
            const double uc_t = vin[M::nu * k + L.c];
            const double x = sx, v = L.end ? uc_t : sv;
            const double x2 = __builtin_fma(M::hTs, v, x);
            const double af = fc_tan_pair(L, buf, L.kf, L.st_f, hi ? x2 : x, st_k, xs);     // lower half: a1, upper: a2
            const double kxf = __builtin_fma(M::hTs, af, v);                               // kx2 | kx3
            const double xsd = __builtin_fma(hi ? M::Ts : M::hTs, kxf, x);                 // x3 | x4
            const double as = fc_tan_pair(L, buf, L.ks, L.st_s, xsd, st_k, xs);            // a3 | a4
            double a1, a2, a3, a4;
            row_partner<true>(af, a1, a2);
            row_partner<true>(as, a3, a4);
            const double kx2 = __builtin_fma(M::hTs, a1, v), kx3 = __builtin_fma(M::hTs, a2, v);
            const double kx4 = __builtin_fma(M::Ts, a3, v);
            double tx = __builtin_fma(2.0, kx2, v), tv = __builtin_fma(2.0, a2, a1);
            tx        = __builtin_fma(2.0, kx3, tx);
            tv        = __builtin_fma(2.0, a3, tv);
            tx        = tx + kx4;
            tv        = tv + a4;
            sx        = __builtin_fma(M::c6, tx, x);
            sv        = __builtin_fma(M::c6, tv, v);
        }
        wave_sync();
        long long t1 = clock64();
        lx = lv = 0.0;
        {
            const double xt = L.end ? sx : sv;
            lx = L.end ? __builtin_fma(M::c2a, xt, lx) : lx;
            lv = L.mass ? __builtin_fma(M::c2b, xt, lv) : lv;
        }
        for (int k = NST - 1; k >= 0; --k) {
            const char *buf   = ring + (k % RING) * STAGE;
            const double uc_t = vin[M::nu * k + L.c];
            const double sgx = M::c6 * lx, sgv = M::c6 * lv;
            const double sgx2 = 2.0 * sgx, sgv2 = 2.0 * sgv;
            const double m4   = L.mass_l * sgx;
            const double kbv3 = __builtin_fma(M::Ts, m4, sgv2);
            const double Gf   = fc_adj_tan_pair(L, buf, L.af, L.dt_f, hi ? sgv : kbv3, xs);   // lower: G3, upper: G4
            const double kbx  = __builtin_fma(hi ? M::Ts : M::hTs, Gf, sgx2);                // kbx2 | kbx3
            const double mm   = L.mass_l * kbx;
            const double kvs  = __builtin_fma(M::hTs, mm, hi ? sgv2 : sgv);                  // kbv1 | kbv2
            const double Gs   = fc_adj_tan_pair(L, buf, L.as, L.dt_s, kvs, xs);              // lower: G1, upper: G2
            double G3, G4, G1, G2;
            row_partner<true>(Gf, G3, G4);
            row_partner<true>(Gs, G1, G2);
            const double kbx3 = __builtin_fma(M::Ts, G4, sgx2), kbx2 = __builtin_fma(M::hTs, G3, sgx2);
            const double kbx1 = __builtin_fma(M::hTs, G2, sgx);
            const double m3 = L.mass_l * kbx3, m2 = L.mass_l * kbx2, m1 = L.mass_l * kbx1;
            const double nx = (((lx + G4) + G3) + G2) + G1, nv = (((lv + m4) + m3) + m2) + m1;
            const double ubar_t = ((sgx + kbx3) + kbx2) + kbx1;
            if (L.end && !hi)
                hv[M::nu * k + L.c] = __builtin_fma(M::c2g, uc_t, ubar_t);
            lx = L.end ? __builtin_fma(M::c2a, uc_t, nx) : nx;
            lv = L.mass ? __builtin_fma(M::c2b, uc_t, nv) : nv;
        }
        wave_sync();
        long long t2 = clock64();
        cf += t1 - t0;
        ca += t2 - t1;
    }
    if (L.active && !hi) {
        out->x[L.q] = sx; out->v[L.q] = sv;
        out->lx[L.q] = lx; out->lv[L.q] = lv;
    }
    for (int i = threadIdx.x; i < M::nu * NST; i += WAVE)
        out->hv[i] = hv[i];
    if (threadIdx.x == 0) {
        out->cyc_fwd = cf;
        out->cyc_adj = ca;
    }
}


// ---- V6: V4 with M d_t summed before K lF_t in the adjoint (another rounding: NOT bit-identical - timing only) ----
__device__ __forceinline__ double fc_adj_tan_pair6(const PairLane &L, const char *buf, const int (&ko)[3], int dt_off, double kv,
                                                  double *xs) {
    const double lF = M::inv_m * __builtin_fma(L.nfirst_l, lane_below(kv), -kv);
    const double K0 = ld_lds(buf, ko[0]), K1 = ld_lds(buf, ko[1]), K2 = ld_lds(buf, ko[2]);
    const double M0 = ld_lds(buf, 2048 + ko[0]), M1 = ld_lds(buf, 2048 + ko[1]), M2 = ld_lds(buf, 2048 + ko[2]);
    const double D0 = ld_lds(buf, dt_off), D1 = ld_lds(buf, dt_off + 8), D2 = ld_lds(buf, dt_off + 16);
    const M::Comps c = gather_pair(L, lF, xs);
    double ld = M0 * D0;                       // (needs nothing of the gather: off the dependent chain)
    ld        = __builtin_fma(M1, D1, ld);
    ld        = __builtin_fma(M2, D2, ld);
    ld        = __builtin_fma(K0, c.c0, ld);
    ld        = __builtin_fma(K1, c.c1, ld);
    ld        = __builtin_fma(K2, c.c2, ld);
    return __builtin_fma(-L.nend_l, lane_above(ld), ld);
}
__global__ __launch_bounds__(64) void v6_kernel(const double *g_tape, const double *g_v, double *g_st, Out *out, int reps) {
    __shared__ __attribute__((aligned(16))) char ring[RING * STAGE];
    __shared__ double vin[NV], hv[NV], xs[64];
    const PairLane L(threadIdx.x);
    fill_lds(ring, vin, g_tape, g_v, 64);
    const bool hi = L.h != 0;
    double sx = 0, sv = 0, lx = 0, lv = 0;
    long long cf = 0, ca = 0;
    for (int r = 0; r < reps; ++r) {
        sx = sv = 0.0;
        long long t0 = clock64();
        for (int k = 0; k < NST; ++k) {
            const char *buf = ring + (k % RING) * STAGE;
            char *st_k      = reinterpret_cast<char *>(g_st) + (size_t)k * M::ST_STAGE_BYTES;
            const double uc_t = vin[M::nu * k + L.c];
            const double x = sx, v = L.end ? uc_t : sv;
            const double x2 = __builtin_fma(M::hTs, v, x);
            const double af = fc_tan_pair(L, buf, L.kf, L.st_f, hi ? x2 : x, st_k, xs);     // lower half: a1, upper: a2
            const double kxf = __builtin_fma(M::hTs, af, v);                               // kx2 | kx3
            const double xsd = __builtin_fma(hi ? M::Ts : M::hTs, kxf, x);                 // x3 | x4
            const double as = fc_tan_pair(L, buf, L.ks, L.st_s, xsd, st_k, xs);            // a3 | a4
            double a1, a2, a3, a4;
            row_partner<true>(af, a1, a2);
            row_partner<true>(as, a3, a4);
            const double kx2 = __builtin_fma(M::hTs, a1, v), kx3 = __builtin_fma(M::hTs, a2, v);
            const double kx4 = __builtin_fma(M::Ts, a3, v);
            double tx = __builtin_fma(2.0, kx2, v), tv = __builtin_fma(2.0, a2, a1);
            tx        = __builtin_fma(2.0, kx3, tx);
            tv        = __builtin_fma(2.0, a3, tv);
            tx        = tx + kx4;
            tv        = tv + a4;
            sx        = __builtin_fma(M::c6, tx, x);
            sv        = __builtin_fma(M::c6, tv, v);
        }
        wave_sync();
        long long t1 = clock64();
        lx = lv = 0.0;
        {
            const double xt = L.end ? sx : sv;
            lx = L.end ? __builtin_fma(M::c2a, xt, lx) : lx;
            lv = L.mass ? __builtin_fma(M::c2b, xt, lv) : lv;
        }
        for (int k = NST - 1; k >= 0; --k) {
            const char *buf   = ring + (k % RING) * STAGE;
            const double uc_t = vin[M::nu * k + L.c];
            const double sgx = M::c6 * lx, sgv = M::c6 * lv;
            const double sgx2 = 2.0 * sgx, sgv2 = 2.0 * sgv;
            const double m4   = L.mass_l * sgx;
            const double kbv3 = __builtin_fma(M::Ts, m4, sgv2);
            const double Gf   = fc_adj_tan_pair6(L, buf, L.af, L.dt_f, hi ? sgv : kbv3, xs);   // lower: G3, upper: G4
            const double kbx  = __builtin_fma(hi ? M::Ts : M::hTs, Gf, sgx2);                // kbx2 | kbx3
            const double mm   = L.mass_l * kbx;
            const double kvs  = __builtin_fma(M::hTs, mm, hi ? sgv2 : sgv);                  // kbv1 | kbv2
            const double Gs   = fc_adj_tan_pair6(L, buf, L.as, L.dt_s, kvs, xs);              // lower: G1, upper: G2
            double G3, G4, G1, G2;
            row_partner<true>(Gf, G3, G4);
            row_partner<true>(Gs, G1, G2);
            const double kbx3 = __builtin_fma(M::Ts, G4, sgx2), kbx2 = __builtin_fma(M::hTs, G3, sgx2);
            const double kbx1 = __builtin_fma(M::hTs, G2, sgx);
            const double m3 = L.mass_l * kbx3, m2 = L.mass_l * kbx2, m1 = L.mass_l * kbx1;
            const double nx = (((lx + G4) + G3) + G2) + G1, nv = (((lv + m4) + m3) + m2) + m1;
            const double ubar_t = ((sgx + kbx3) + kbx2) + kbx1;
            if (L.end && !hi)
                hv[M::nu * k + L.c] = __builtin_fma(M::c2g, uc_t, ubar_t);
            lx = L.end ? __builtin_fma(M::c2a, uc_t, nx) : nx;
            lv = L.mass ? __builtin_fma(M::c2b, uc_t, nv) : nv;
        }
        wave_sync();
        long long t2 = clock64();
        cf += t1 - t0;
        ca += t2 - t1;
    }
    if (L.active && !hi) {
        out->x[L.q] = sx; out->v[L.q] = sv;
        out->lx[L.q] = lx; out->lv[L.q] = lv;
    }
    for (int i = threadIdx.x; i < M::nu * NST; i += WAVE)
        out->hv[i] = hv[i];
    if (threadIdx.x == 0) {
        out->cyc_fwd = cf;
        out->cyc_adj = ca;
    }
}


// ---- V5: V4 with all of a stage's tape reads issued at the stage start (both substages), GATHER5: 0 LDS row, 1 ds_bpermute
template <int G>
__device__ __forceinline__ M::Comps gather5(const PairLane &L, double v, double *xs) {
    if constexpr (G == 0) {
        return gather_pair(L, v, xs);
    } else {
        const int base = (32 * L.h + L.p) * 4; // byte address of the lane holding component 0 of this point
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const int l0 = __builtin_amdgcn_ds_bpermute(base, lo), h0 = __builtin_amdgcn_ds_bpermute(base, hi);
        const int l1 = __builtin_amdgcn_ds_bpermute(base + 40, lo), h1 = __builtin_amdgcn_ds_bpermute(base + 40, hi);
        const int l2 = __builtin_amdgcn_ds_bpermute(base + 80, lo), h2 = __builtin_amdgcn_ds_bpermute(base + 80, hi);
        return {__hiloint2double(h0, l0), __hiloint2double(h1, l1), __hiloint2double(h2, l2)};
    }
}
struct K3 { double k[3]; };
struct KMD { double k[3], m[3], d[3]; };
__device__ __forceinline__ K3 ldK(const char *buf, const int (&ko)[3]) { return {{ld_lds(buf, ko[0]), ld_lds(buf, ko[1]), ld_lds(buf, ko[2])}}; }
__device__ __forceinline__ KMD ldKMD(const char *buf, const int (&ko)[3], int dt) {
    return {{ld_lds(buf, ko[0]), ld_lds(buf, ko[1]), ld_lds(buf, ko[2])},
            {ld_lds(buf, 2048 + ko[0]), ld_lds(buf, 2048 + ko[1]), ld_lds(buf, 2048 + ko[2])},
            {ld_lds(buf, dt), ld_lds(buf, dt + 8), ld_lds(buf, dt + 16)}};
}
template <int G>
__device__ __forceinline__ double fc_tan_pair5(const PairLane &L, const K3 &g, int st_off, double x, char *st_stage, double *xs) {
    const double d = __builtin_fma(-L.nfirst_l, lane_below(x), x);
    *reinterpret_cast<double *>(st_stage + st_off) = d;
    const M::Comps c = gather5<G>(L, d, xs);
    double F = g.k[0] * c.c0;
    F        = __builtin_fma(g.k[1], c.c1, F);
    F        = __builtin_fma(g.k[2], c.c2, F);
    return L.inv_m_l * (lane_above(F) - F);
}
template <int G>
__device__ __forceinline__ double fc_adj_tan_pair5(const PairLane &L, const KMD &g, double kv, double *xs) {
    const double lF = M::inv_m * (L.nfirst_l * lane_below(kv) - kv);
    const M::Comps c = gather5<G>(L, lF, xs);
    double ld = g.k[0] * c.c0;
    ld        = __builtin_fma(g.k[1], c.c1, ld);
    ld        = __builtin_fma(g.k[2], c.c2, ld);
    ld        = __builtin_fma(g.m[0], g.d[0], ld);
    ld        = __builtin_fma(g.m[1], g.d[1], ld);
    ld        = __builtin_fma(g.m[2], g.d[2], ld);
    return __builtin_fma(-L.nend_l, lane_above(ld), ld);
}
template <int G>
__global__ __launch_bounds__(64) void v5_kernel(const double *g_tape, const double *g_v, double *g_st, Out *out, int reps) {
    __shared__ __attribute__((aligned(16))) char ring[RING * STAGE];
    __shared__ double vin[NV], hv[NV], xs[64];
    const PairLane L(threadIdx.x);
    fill_lds(ring, vin, g_tape, g_v, 64);
    const bool hi = L.h != 0;
    double sx = 0, sv = 0, lx = 0, lv = 0;
    long long cf = 0, ca = 0;
    for (int r = 0; r < reps; ++r) {
        sx = sv = 0.0;
        long long t0 = clock64();
        for (int k = 0; k < NST; ++k) {
            const char *buf = ring + (k % RING) * STAGE;
            char *st_k      = reinterpret_cast<char *>(g_st) + (size_t)k * M::ST_STAGE_BYTES;
            const K3 gF = ldK(buf, L.kf), gS = ldK(buf, L.ks);
            const double uc_t = vin[M::nu * k + L.c];
            const double x = sx, v = L.end ? uc_t : sv;
            const double x2 = __builtin_fma(M::hTs, v, x);
            const double af = fc_tan_pair5<G>(L, gF, L.st_f, hi ? x2 : x, st_k, xs);     // lower half: a1, upper: a2
            const double kxf = __builtin_fma(M::hTs, af, v);                               // kx2 | kx3
            const double xsd = __builtin_fma(hi ? M::Ts : M::hTs, kxf, x);                 // x3 | x4
            const double as = fc_tan_pair5<G>(L, gS, L.st_s, xsd, st_k, xs);            // a3 | a4
            double a1, a2, a3, a4;
            row_partner<true>(af, a1, a2);
            row_partner<true>(as, a3, a4);
            const double kx2 = __builtin_fma(M::hTs, a1, v), kx3 = __builtin_fma(M::hTs, a2, v);
            const double kx4 = __builtin_fma(M::Ts, a3, v);
            double tx = __builtin_fma(2.0, kx2, v), tv = __builtin_fma(2.0, a2, a1);
            tx        = __builtin_fma(2.0, kx3, tx);
            tv        = __builtin_fma(2.0, a3, tv);
            tx        = tx + kx4;
            tv        = tv + a4;
            sx        = __builtin_fma(M::c6, tx, x);
            sv        = __builtin_fma(M::c6, tv, v);
        }
        wave_sync();
        long long t1 = clock64();
        lx = lv = 0.0;
        {
            const double xt = L.end ? sx : sv;
            lx = L.end ? __builtin_fma(M::c2a, xt, lx) : lx;
            lv = L.mass ? __builtin_fma(M::c2b, xt, lv) : lv;
        }
        for (int k = NST - 1; k >= 0; --k) {
            const char *buf   = ring + (k % RING) * STAGE;
            const KMD hF = ldKMD(buf, L.af, L.dt_f), hS = ldKMD(buf, L.as, L.dt_s);
            const double uc_t = vin[M::nu * k + L.c];
            const double sgx = M::c6 * lx, sgv = M::c6 * lv;
            const double sgx2 = 2.0 * sgx, sgv2 = 2.0 * sgv;
            const double m4   = L.mass_l * sgx;
            const double kbv3 = __builtin_fma(M::Ts, m4, sgv2);
            const double Gf   = fc_adj_tan_pair5<G>(L, hF, hi ? sgv : kbv3, xs);   // lower: G3, upper: G4
            const double kbx  = __builtin_fma(hi ? M::Ts : M::hTs, Gf, sgx2);                // kbx2 | kbx3
            const double mm   = L.mass_l * kbx;
            const double kvs  = __builtin_fma(M::hTs, mm, hi ? sgv2 : sgv);                  // kbv1 | kbv2
            const double Gs   = fc_adj_tan_pair5<G>(L, hS, kvs, xs);              // lower: G1, upper: G2
            double G3, G4, G1, G2;
            row_partner<true>(Gf, G3, G4);
            row_partner<true>(Gs, G1, G2);
            const double kbx3 = __builtin_fma(M::Ts, G4, sgx2), kbx2 = __builtin_fma(M::hTs, G3, sgx2);
            const double kbx1 = __builtin_fma(M::hTs, G2, sgx);
            const double m3 = L.mass_l * kbx3, m2 = L.mass_l * kbx2, m1 = L.mass_l * kbx1;
            const double nx = (((lx + G4) + G3) + G2) + G1, nv = (((lv + m4) + m3) + m2) + m1;
            const double ubar_t = ((sgx + kbx3) + kbx2) + kbx1;
            if (L.end && !hi)
                hv[M::nu * k + L.c] = __builtin_fma(M::c2g, uc_t, ubar_t);
            lx = L.end ? __builtin_fma(M::c2a, uc_t, nx) : nx;
            lv = L.mass ? __builtin_fma(M::c2b, uc_t, nv) : nv;
        }
        wave_sync();
        long long t2 = clock64();
        cf += t1 - t0;
        ca += t2 - t1;
    }
    if (L.active && !hi) {
        out->x[L.q] = sx; out->v[L.q] = sv;
        out->lx[L.q] = lx; out->lv[L.q] = lv;
    }
    for (int i = threadIdx.x; i < M::nu * NST; i += WAVE)
        out->hv[i] = hv[i];
    if (threadIdx.x == 0) {
        out->cyc_fwd = cf;
        out->cyc_adj = ca;
    }
}


/// One lone chain without any exchange (wrong numbers, right instruction stream): what a wavefront of V3 spends
/// outside the exchange.
__global__ __launch_bounds__(64) void chain_only_kernel(const double *g_tape, const double *g_v, double *g_st, Out *out, int reps) {
    __shared__ __attribute__((aligned(16))) char ring[RING * STAGE];
    __shared__ double vin[NV + 8];
    const int w = 0, lane = threadIdx.x;
    const M::Lane L(lane);
    fill_lds(ring, vin, g_tape, g_v, 64);
    M::State s_t{0, 0};
    long long cf = 0;
    for (int r = 0; r < reps; ++r) {
        s_t = {0.0, 0.0};
        long long t0 = clock64();
        M::TapeK gf = M::tape_fwd(L, ring, 0), gs = M::tape_fwd(L, ring, 2);
        double uc_t = M::load_input(L, vin);
        for (int k = 0; k < NST; ++k) {
            double4 *st_k   = reinterpret_cast<double4 *>(g_st) + (size_t)k * M::ST_D4_PER_STAGE;
            const M::State s = M::with_input(L, s_t, uc_t);
            const double af  = fc_tan1_pre(L, gf, 0, s.x, st_k);
            const double kxf = __builtin_fma(M::hTs, af, s.v);
            const double xsd = __builtin_fma(M::hTs, kxf, s.x);
            const double as  = fc_tan1_pre(L, gs, 2, xsd, st_k);
            const char *nbuf = ring + ((k + 1) % RING) * STAGE;
            gf   = M::tape_fwd(L, nbuf, 0);
            gs   = M::tape_fwd(L, nbuf, 2);
            uc_t = M::load_input(L, vin + M::nu * (k + 1 < NST ? k + 1 : k));
            asm volatile("" ::: "memory");
            const double a1 = af, a2 = af, a3 = as, a4 = as;
            const double kx2 = __builtin_fma(M::hTs, a1, s.v), kx3 = __builtin_fma(M::hTs, a2, s.v);
            const double kx4 = __builtin_fma(M::Ts, a3, s.v);
            double tx = __builtin_fma(2.0, kx2, s.v), tv = __builtin_fma(2.0, a2, a1);
            tx        = __builtin_fma(2.0, kx3, tx);
            tv        = __builtin_fma(2.0, a3, tv);
            tx        = tx + kx4;
            tv        = tv + a4;
            s_t       = {__builtin_fma(M::c6, tx, s.x), __builtin_fma(M::c6, tv, s.v)};
        }
        wave_sync();
        cf += clock64() - t0;
    }
    if (L.active)
        out->x[L.q] = s_t.x;
    if (lane == 0) {
        out->cyc_fwd = cf;
        out->cyc_adj = w;
    }
}

/// The cost of the exchange alone: the swap_with_partner of a stage loop with nothing around it.
__global__ __launch_bounds__(128) void exchange_kernel(long long *cycles, int n) {
    __shared__ __attribute__((aligned(16))) double2 ex[2][2 * WAVE];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double a = lane, b = w;
    __syncthreads();
    long long t0 = clock64();
    for (int k = 0; k < n; ++k) {
        double oa, ob;
        swap_with_partner(ex[k & 1], w, lane, a, b, oa, ob);
        a = oa + 1.0; // (dependent: the next exchange cannot start before this one's read returned)
        b = ob;
    }
    long long t1 = clock64();
    if (threadIdx.x == 0)
        cycles[0] = t1 - t0;
    if (a == -1.0)
        cycles[1] = (long long)b;
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 50;
    srand(1);
    auto rnd = []() { return 2.0 * rand() / RAND_MAX - 1.0; };
    std::vector<double> tape(RING * STAGE / 8), v(NV, 0.0);
    for (int s = 0; s < RING; ++s)
        for (int i = 0; i < STAGE / 8; ++i)
            tape[s * (STAGE / 8) + i] = (i < 512 ? 0.05 : 0.02) * rnd();
    for (int i = 0; i < M::nu * NST; ++i)
        v[i] = rnd();
    const size_t stn = (size_t)NST * 128;
    double *d_tape, *d_v, *d_st[9];
    Out *d_o[9];
    long long *d_c;
    (void)hipMalloc(&d_tape, tape.size() * 8); (void)hipMalloc(&d_v, NV * 8); (void)hipMalloc(&d_c, 16);
    (void)hipMemcpy(d_tape, tape.data(), tape.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_v, v.data(), NV * 8, hipMemcpyHostToDevice);
    for (int i = 0; i < 9; ++i) {
        (void)hipMalloc(&d_st[i], stn * 8); (void)hipMalloc(&d_o[i], sizeof(Out));
        (void)hipMemset(d_st[i], 0, stn * 8); (void)hipMemset(d_o[i], 0, sizeof(Out));
    }
    for (int warm = 0; warm < 2; ++warm) {
        hipLaunchKernelGGL(v0_kernel, 1, 64, 0, 0, d_tape, d_v, d_st[0], d_o[0], reps);
        hipLaunchKernelGGL(v2_kernel<0>, 1, 128, 0, 0, d_tape, d_v, d_st[1], d_o[1], reps);
        hipLaunchKernelGGL(v2_kernel<1>, 1, 128, 0, 0, d_tape, d_v, d_st[2], d_o[2], reps);
        hipLaunchKernelGGL(v3_kernel, 1, 128, 0, 0, d_tape, d_v, d_st[3], d_o[3], reps);
        hipLaunchKernelGGL(chain_only_kernel, 1, 64, 0, 0, d_tape, d_v, d_st[4], d_o[4], reps);
        hipLaunchKernelGGL(v4_kernel, 1, 64, 0, 0, d_tape, d_v, d_st[5], d_o[5], reps);
        hipLaunchKernelGGL(v5_kernel<0>, 1, 64, 0, 0, d_tape, d_v, d_st[6], d_o[6], reps);
        hipLaunchKernelGGL(v5_kernel<1>, 1, 64, 0, 0, d_tape, d_v, d_st[7], d_o[7], reps);
        hipLaunchKernelGGL(v6_kernel, 1, 64, 0, 0, d_tape, d_v, d_st[8], d_o[8], reps);
        hipLaunchKernelGGL(exchange_kernel, 1, 128, 0, 0, d_c, 1000);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    }
    Out o[9];
    std::vector<double> st[9];
    for (int i = 0; i < 9; ++i) {
        (void)hipMemcpy(&o[i], d_o[i], sizeof(Out), hipMemcpyDeviceToHost);
        st[i].resize(stn);
        (void)hipMemcpy(st[i].data(), d_st[i], stn * 8, hipMemcpyDeviceToHost);
    }
    long long exc[2];
    (void)hipMemcpy(exc, d_c, 16, hipMemcpyDeviceToHost);
    const char *names[9] = {"V0 one wavefront, chains interleaved      ", "V2 two wavefronts, gather through swaps   ",
                            "V2 two wavefronts, gather through LDS row ", "V3 = V2 (swaps) + LDS reads a stage ahead ", "",
                            "V4 ONE wavefront, chains side by side     ",
                            "V5 = V4 + tape reads at the stage start   ", "V5 with the gather through ds_bpermute    ",
                            "V6 = V4, M d_t summed first (other bits!)  "};
    int total_bad = 0;
    for (int i = 0; i < 9; ++i) {
        if (i == 4)
            continue;
        int bad = 0, bad_st = 0;
        for (int q = 0; q < 30; ++q) {
            bad += o[0].x[q] != o[i].x[q]; bad += o[0].v[q] != o[i].v[q];
            bad += o[0].lx[q] != o[i].lx[q]; bad += o[0].lv[q] != o[i].lv[q];
        }
        for (int k = 0; k < M::nu * NST; ++k)
            bad += o[0].hv[k] != o[i].hv[k];
        for (int k = 0; k < NST; ++k)
            for (int j = 0; j < 4; ++j)
                for (int q = 0; q < 30; ++q)
                    bad_st += st[0][(size_t)k * 128 + j * 32 + q] != st[i][(size_t)k * 128 + j * 32 + q];
        printf("%s: forward %6.0f cycles/stage, adjoint %6.0f cycles/stage; mismatches vs V0: %d (+ %d in d_t)\n", names[i],
               (double)o[i].cyc_fwd / reps / NST, (double)o[i].cyc_adj / reps / NST, bad, bad_st);
        if (i != 8)
            total_bad += bad + bad_st;
    }
    printf("one chain of V3's forward sweep without the exchange: %.0f cycles/stage\n", (double)o[4].cyc_fwd / reps / NST);
    printf("exchange alone (16-byte write, lgkmcnt(0), s_barrier, 16-byte read, dependent): %.0f cycles\n", (double)exc[0] / 1000);
    printf("Hv[0] = %.17g, |x_t| <= %.3g\n", o[0].hv[0], fabs(o[0].x[27]));
    return total_bad == 0 ? 0 : 2;
}
