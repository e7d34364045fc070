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
#include "../../pantr-cdc2023-experiments_amd/csrc/chain_model.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace pantr;
using M = ChainModel;

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
    double *d_tape, *d_v, *d_st[3];
    Out *d_o[3];
    long long *d_c;
    (void)hipMalloc(&d_tape, tape.size() * 8); (void)hipMalloc(&d_v, NV * 8); (void)hipMalloc(&d_c, 16);
    (void)hipMemcpy(d_tape, tape.data(), tape.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_v, v.data(), NV * 8, hipMemcpyHostToDevice);
    for (int i = 0; i < 3; ++i) {
        (void)hipMalloc(&d_st[i], stn * 8); (void)hipMalloc(&d_o[i], sizeof(Out));
        (void)hipMemset(d_st[i], 0, stn * 8); (void)hipMemset(d_o[i], 0, sizeof(Out));
    }
    for (int warm = 0; warm < 2; ++warm) {
        hipLaunchKernelGGL(v0_kernel, 1, 64, 0, 0, d_tape, d_v, d_st[0], d_o[0], reps);
        hipLaunchKernelGGL(v2_kernel<0>, 1, 128, 0, 0, d_tape, d_v, d_st[1], d_o[1], reps);
        hipLaunchKernelGGL(v2_kernel<1>, 1, 128, 0, 0, d_tape, d_v, d_st[2], d_o[2], reps);
        hipLaunchKernelGGL(exchange_kernel, 1, 128, 0, 0, d_c, 1000);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    }
    Out o[3];
    std::vector<double> st[3];
    for (int i = 0; i < 3; ++i) {
        (void)hipMemcpy(&o[i], d_o[i], sizeof(Out), hipMemcpyDeviceToHost);
        st[i].resize(stn);
        (void)hipMemcpy(st[i].data(), d_st[i], stn * 8, hipMemcpyDeviceToHost);
    }
    long long exc[2];
    (void)hipMemcpy(exc, d_c, 16, hipMemcpyDeviceToHost);
    const char *names[3] = {"V0 one wavefront, chains interleaved      ", "V2 two wavefronts, gather through swaps   ",
                            "V2 two wavefronts, gather through LDS row "};
    int total_bad = 0;
    for (int i = 0; i < 3; ++i) {
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
        total_bad += bad + bad_st;
    }
    printf("exchange alone (16-byte write, lgkmcnt(0), s_barrier, 16-byte read, dependent): %.0f cycles\n", (double)exc[0] / 1000);
    printf("Hv[0] = %.17g, |x_t| <= %.3g\n", o[0].hv[0], fabs(o[0].x[27]));
    return total_bad == 0 ? 0 : 2;
}
