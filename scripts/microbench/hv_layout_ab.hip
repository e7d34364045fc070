// A/B of two lane layouts for the hanging chain's Hessian-vector product sweeps on a lone wavefront
// (development aid).
//   V0: the product's layout (csrc/chain_model.hip.h): lane = (component, point), 30 lanes busy, one
//       instruction per component-wise operation, a cross-row gather of d_t / lF_t per substage.
//   V1: "components in registers": lane = (substage chain, point) - row 0 of the wavefront evaluates the
//       substages (1, 3) of a stage, row 1 the substages (2, 4); the three components of a point are
//       three registers of its lane: K d_t is three independent FMA chains without any gather, the
//       rows exchange their accelerations (adjoint: their G's) once per substage pair through
//       v_permlane16_swap.  Points 0..9 sit on lanes 6..15 of a row so that the end point's
//       "next point" is the zero a DPP row shift supplies.
// Both variants read the same K / M / d_t rows from LDS (eight stages' worth, used round-robin, so that
// nothing can be hoisted out of the stage loop; no HBM traffic in the timed loops besides the d_t stores)
// and must produce the same bits: the program compares the final tangent state, every stored d_t, the
// product's output vector and the final adjoint, then prints shader cycles per stage for each sweep.
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -o hv_layout_ab hv_layout_ab.hip && ./hv_layout_ab
#include "chain_row_layout.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace pantr;
using M = ChainRowModel;

constexpr int NST   = 60;                 // stages per sweep
constexpr int RING  = 8;                  // stages of tape kept in LDS
constexpr int STAGE = M::LDS_STAGE_BYTES; // 4 KiB tape + 1 KiB d_t rows
constexpr int NV    = 192;                // decision vector (3 x 60, padded)

struct Out {
    double x[32], v[32];   // final tangent state, compact slot 3p+c
    double lx[32], lv[32]; // final second-order adjoint
    double hv[NV];
    long long cyc_fwd, cyc_adj;
};

__device__ __forceinline__ void fill_lds(char *ring, double *vin, const double *g_tape, const double *g_v) {
    for (int i = threadIdx.x; i < RING * STAGE / 8; i += WAVE)
        reinterpret_cast<double *>(ring)[i] = g_tape[i];
    for (int i = threadIdx.x; i < NV; i += WAVE)
        vin[i] = g_v[i];
    wave_sync();
}

// ---------------------------------------------------------------------------------------------------
// V0: the product's sweeps, as WaveProblem::hess_prod_tape drives them (without the LDS-DMA)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void v0_kernel(const double *g_tape, const double *g_v, double *g_st, Out *out, int reps) {
    __shared__ __attribute__((aligned(16))) char ring[RING * STAGE];
    __shared__ double vin[NV], hv[NV], xs[32];
    const M::Lane L(threadIdx.x);
    fill_lds(ring, vin, g_tape, g_v);
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
            M::cost_grad_x_add_tan(L, uc_t, lam_t); // (any per-lane value does for the cost term here)
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

// ---------------------------------------------------------------------------------------------------
// V1: lane = (substage chain, point), components in registers
// ---------------------------------------------------------------------------------------------------
struct LaneR {
    int l, i, p, pc, chain;
    bool active, mass, end;
    double inv_m_l, mass_l;
    double cx1, cx2; ///< forward: x-evaluation coefficients of the chain's first / second substage
    double cA, sA, cB, sB; ///< adjoint: kv of the first substage = fma(cA, m4, sA sgv); kbx = fma(cB, G, sgx2); kv' = fma(hTs, m, sB sgv)
    __device__ explicit LaneR(int lane) {
        l = lane; i = lane & 15; chain = (lane >> 4) & 1; // rows 2, 3 mirror rows 0, 1
        p = i - 6; active = p >= 0; pc = active ? p : 0;
        mass = active && p < M::NB; end = p == M::NB;
        inv_m_l = mass ? M::inv_m : 0.0; mass_l = mass ? 1.0 : 0.0;
        cx1 = chain ? M::hTs : 0.0; cx2 = chain ? M::Ts : M::hTs;
        cA = chain ? M::Ts : 0.0; sA = chain ? 2.0 : 1.0;
        cB = chain ? M::hTs : M::Ts; sB = chain ? 1.0 : 2.0;
    }
};
struct V3 {
    double c[3];
};
__device__ __forceinline__ V3 prev3(const V3 &a) { return {{row_from_prev(a.c[0]), row_from_prev(a.c[1]), row_from_prev(a.c[2])}}; }
__device__ __forceinline__ V3 next3(const V3 &a) { return {{row_from_next(a.c[0]), row_from_next(a.c[1]), row_from_next(a.c[2])}}; }
/// Row 0's and row 1's value of `a`, on every row.
__device__ __forceinline__ void rows2(const V3 &a, V3 &even, V3 &odd) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
        row_partner<false>(a.c[c], even.c[c], odd.c[c]);
}
struct Sym6 {
    double e[6]; // (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
};
__device__ __forceinline__ Sym6 load6(const char *row, const LaneR &L) {
    const double *s = reinterpret_cast<const double *>(row) + M::spring_base(L.pc);
    return {{s[0], s[1], s[2], s[3], s[4], s[5]}};
}
/// out_c = sum_c2 S[c][c2] v_c2, continued from `acc` when CONT:  c0 v0, then fma(c1, v1, .), fma(c2, v2, .)
template <bool CONT>
__device__ __forceinline__ V3 symv(const Sym6 &S, const V3 &v, const V3 &acc) {
    // rows of the symmetric matrix: (e0 e1 e2), (e1 e3 e4), (e2 e4 e5)
    const double r[3][3] = {{S.e[0], S.e[1], S.e[2]}, {S.e[1], S.e[3], S.e[4]}, {S.e[2], S.e[4], S.e[5]}};
    V3 o;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double t = CONT ? __builtin_fma(r[c][0], v.c[0], acc.c[c]) : r[c][0] * v.c[0];
        t        = __builtin_fma(r[c][1], v.c[1], t);
        o.c[c]   = __builtin_fma(r[c][2], v.c[2], t);
    }
    return o;
}

struct State3 {
    V3 x, v;
};

/// f_c tangent at xe for this lane's substage j: a = inv_m (F_{p+1} - F_p), F = K d_t; stores d_t.
__device__ __forceinline__ V3 fc_tan_r(const LaneR &L, const char *buf, int j, const V3 &xe, double *st_stage) {
    const V3 pv = prev3(xe);
    V3 d;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        d.c[c] = xe.c[c] - pv.c[c];
    double *strow = st_stage + j * 32;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        strow[L.active ? 3 * L.p + c : 30 + (c & 1)] = d.c[c];
    const Sym6 K = load6(buf + j * 512, L);
    const V3 F = symv<false>(K, d, d), Fn = next3(F);
    V3 a;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        a.c[c] = L.inv_m_l * (Fn.c[c] - F.c[c]);
    return a;
}
__device__ __forceinline__ State3 rk4_tan_r(const LaneR &L, const char *buf, const State3 &s_in, const double *u_t, double *st_stage) {
    State3 s = s_in;
    const double u0 = u_t[0], u1 = u_t[1], u2 = u_t[2]; // (read by every lane: a select, not a branch)
    s.v.c[0] = L.end ? u0 : s.v.c[0];
    s.v.c[1] = L.end ? u1 : s.v.c[1];
    s.v.c[2] = L.end ? u2 : s.v.c[2];
    V3 xe, kx;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        xe.c[c] = __builtin_fma(L.cx1, s.v.c[c], s.x.c[c]);
    const V3 af = fc_tan_r(L, buf, L.chain, xe, st_stage); // row 0: a1, row 1: a2
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        kx.c[c] = __builtin_fma(M::hTs, af.c[c], s.v.c[c]);
        xe.c[c] = __builtin_fma(L.cx2, kx.c[c], s.x.c[c]);
    }
    const V3 as = fc_tan_r(L, buf, 2 + L.chain, xe, st_stage); // row 0: a3, row 1: a4
    V3 a1, a2, a3, a4;
    rows2(af, a1, a2);
    rows2(as, a3, a4);
    State3 o;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double v = s.v.c[c];
        const double kx2 = __builtin_fma(M::hTs, a1.c[c], v), kx3 = __builtin_fma(M::hTs, a2.c[c], v);
        const double kx4 = __builtin_fma(M::Ts, a3.c[c], v);
        double tx = __builtin_fma(2.0, kx2, v), tv = __builtin_fma(2.0, a2.c[c], a1.c[c]);
        tx        = __builtin_fma(2.0, kx3, tx);
        tv        = __builtin_fma(2.0, a3.c[c], tv);
        tx        = tx + kx4;
        tv        = tv + a4.c[c];
        o.x.c[c]  = __builtin_fma(M::c6, tx, s.x.c[c]);
        o.v.c[c]  = __builtin_fma(M::c6, tv, v);
    }
    return o;
}
/// position part of the f_c adjoint tangent for this lane's substage j
__device__ __forceinline__ V3 fc_adj_tan_r(const LaneR &L, const char *buf, int j, const V3 &kv) {
    const V3 pk = prev3(kv);
    V3 lF;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        lF.c[c] = M::inv_m * (pk.c[c] - kv.c[c]);
    const Sym6 K = load6(buf + j * 512, L), Mm = load6(buf + 2048 + j * 512, L);
    const double *st = reinterpret_cast<const double *>(buf + M::TAPE_STAGE_BYTES + j * 256) + 3 * L.pc;
    const V3 D{{st[0], st[1], st[2]}};
    V3 ld       = symv<false>(K, lF, lF);
    ld          = symv<true>(Mm, D, ld);
    const V3 ln = next3(ld); // the end point sits on lane 15 of its row: 0
    V3 G;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        G.c[c] = ld.c[c] - ln.c[c];
    return G;
}
__device__ __forceinline__ void rk4_adj_tan_r(const LaneR &L, const char *buf, const State3 &mu, State3 &sbar, V3 &ubar) {
    V3 sgx, sgv, sgx2, m4, kv, Gf, Gs;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        sgx.c[c]  = M::c6 * mu.x.c[c];
        sgv.c[c]  = M::c6 * mu.v.c[c];
        sgx2.c[c] = 2.0 * sgx.c[c];
        m4.c[c]   = L.mass_l * sgx.c[c];
        kv.c[c]   = __builtin_fma(L.cA, m4.c[c], L.sA * sgv.c[c]); // row 0: sgv (substage 4), row 1: kbv3
    }
    Gf = fc_adj_tan_r(L, buf, 3 - L.chain, kv); // row 0: G4, row 1: G3
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double kbx = __builtin_fma(L.cB, Gf.c[c], sgx2.c[c]); // row 0: kbx3, row 1: kbx2
        const double m   = L.mass_l * kbx;
        kv.c[c]          = __builtin_fma(M::hTs, m, L.sB * sgv.c[c]); // row 0: kbv2, row 1: kbv1
    }
    Gs = fc_adj_tan_r(L, buf, 1 - L.chain, kv); // row 0: G2, row 1: G1
    V3 G4, G3, G2, G1;
    rows2(Gf, G4, G3);
    rows2(Gs, G2, G1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double kbx3 = __builtin_fma(M::Ts, G4.c[c], sgx2.c[c]), kbx2 = __builtin_fma(M::hTs, G3.c[c], sgx2.c[c]);
        const double kbx1 = __builtin_fma(M::hTs, G2.c[c], sgx.c[c]);
        const double m3 = L.mass_l * kbx3, m2 = L.mass_l * kbx2, m1 = L.mass_l * kbx1;
        sbar.x.c[c] = (((mu.x.c[c] + G4.c[c]) + G3.c[c]) + G2.c[c]) + G1.c[c];
        sbar.v.c[c] = (((mu.v.c[c] + m4.c[c]) + m3) + m2) + m1;
        ubar.c[c]   = ((sgx.c[c] + kbx3) + kbx2) + kbx1;
    }
}

__global__ __launch_bounds__(64) void v1_kernel(const double *g_tape, const double *g_v, double *g_st, Out *out, int reps) {
    __shared__ __attribute__((aligned(16))) char ring[RING * STAGE];
    __shared__ double vin[NV], hv[NV];
    const LaneR L(threadIdx.x);
    fill_lds(ring, vin, g_tape, g_v);
    const double c2a_l = L.end ? M::c2a : 0.0, c2b_l = L.mass ? M::c2b : 0.0;
    State3 s_t{}, lam_t{};
    long long cf = 0, ca = 0;
    for (int r = 0; r < reps; ++r) {
        s_t = State3{};
        long long t0 = clock64();
        for (int k = 0; k < NST; ++k)
            s_t = rk4_tan_r(L, ring + (k % RING) * STAGE, s_t, vin + M::nu * k, g_st + (size_t)k * 128);
        wave_sync();
        long long t1 = clock64();
        lam_t = State3{};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double xt = L.end ? s_t.x.c[c] : s_t.v.c[c];
            lam_t.x.c[c] = __builtin_fma(c2a_l, xt, lam_t.x.c[c]);
            lam_t.v.c[c] = __builtin_fma(c2b_l, xt, lam_t.v.c[c]);
        }
        for (int k = NST - 1; k >= 0; --k) {
            const double *u_t = vin + M::nu * k;
            State3 sbar;
            V3 ubar;
            rk4_adj_tan_r(L, ring + (k % RING) * STAGE, lam_t, sbar, ubar);
            const double u0 = u_t[0], u1 = u_t[1], u2 = u_t[2];
            const double h0 = __builtin_fma(M::c2g, u0, ubar.c[0]), h1 = __builtin_fma(M::c2g, u1, ubar.c[1]),
                         h2 = __builtin_fma(M::c2g, u2, ubar.c[2]);
            if (L.end) { // (rows 1..3 store the same values to the same slots)
                hv[M::nu * k + 0] = h0; hv[M::nu * k + 1] = h1; hv[M::nu * k + 2] = h2;
            }
            lam_t = sbar;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                lam_t.x.c[c] = __builtin_fma(c2a_l, u_t[c], lam_t.x.c[c]);
                lam_t.v.c[c] = __builtin_fma(c2b_l, u_t[c], lam_t.v.c[c]);
            }
        }
        wave_sync();
        long long t2 = clock64();
        cf += t1 - t0;
        ca += t2 - t1;
    }
    if (L.active && L.l < 16) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            out->x[3 * L.p + c] = s_t.x.c[c]; out->v[3 * L.p + c] = s_t.v.c[c];
            out->lx[3 * L.p + c] = lam_t.x.c[c]; out->lv[3 * L.p + c] = lam_t.v.c[c];
        }
    }
    for (int i = threadIdx.x; i < M::nu * NST; i += WAVE)
        out->hv[i] = hv[i];
    if (threadIdx.x == 0) {
        out->cyc_fwd = cf;
        out->cyc_adj = ca;
    }
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 50;
    srand(1);
    auto rnd = []() { return 2.0 * rand() / RAND_MAX - 1.0; };
    std::vector<double> tape(RING * STAGE / 8), v(NV, 0.0);
    // K ~ c I + w d d^T with c ~ 0.1, entries small enough that 60 stages stay well inside the range
    for (int s = 0; s < RING; ++s)
        for (int i = 0; i < STAGE / 8; ++i)
            tape[s * (STAGE / 8) + i] = (i < 512 ? 0.05 : 0.02) * rnd();
    for (int i = 0; i < M::nu * NST; ++i)
        v[i] = rnd();
    double *d_tape, *d_v, *d_st0, *d_st1;
    Out *d_o0, *d_o1;
    const size_t stn = (size_t)NST * 128;
    (void)hipMalloc(&d_tape, tape.size() * 8); (void)hipMalloc(&d_v, NV * 8);
    (void)hipMalloc(&d_st0, stn * 8); (void)hipMalloc(&d_st1, stn * 8);
    (void)hipMalloc(&d_o0, sizeof(Out)); (void)hipMalloc(&d_o1, sizeof(Out));
    (void)hipMemcpy(d_tape, tape.data(), tape.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_v, v.data(), NV * 8, hipMemcpyHostToDevice);
    (void)hipMemset(d_st0, 0, stn * 8); (void)hipMemset(d_st1, 0, stn * 8);
    (void)hipMemset(d_o0, 0, sizeof(Out)); (void)hipMemset(d_o1, 0, sizeof(Out));
    Out o0, o1;
    for (int warm = 0; warm < 2; ++warm) {
        hipLaunchKernelGGL(v0_kernel, 1, 64, 0, 0, d_tape, d_v, d_st0, d_o0, reps);
        hipLaunchKernelGGL(v1_kernel, 1, 64, 0, 0, d_tape, d_v, d_st1, d_o1, reps);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    }
    (void)hipMemcpy(&o0, d_o0, sizeof(Out), hipMemcpyDeviceToHost);
    (void)hipMemcpy(&o1, d_o1, sizeof(Out), hipMemcpyDeviceToHost);
    std::vector<double> st0(stn), st1(stn);
    (void)hipMemcpy(st0.data(), d_st0, stn * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(st1.data(), d_st1, stn * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    double big = 0;
    for (int q = 0; q < 30; ++q) {
        bad += o0.x[q] != o1.x[q]; bad += o0.v[q] != o1.v[q];
        bad += o0.lx[q] != o1.lx[q]; bad += o0.lv[q] != o1.lv[q];
        big = fmax(big, fabs(o0.x[q]));
    }
    for (int i = 0; i < M::nu * NST; ++i)
        bad += o0.hv[i] != o1.hv[i];
    int bad_st = 0;
    for (int k = 0; k < NST; ++k)
        for (int j = 0; j < 4; ++j)
            for (int q = 0; q < 30; ++q)
                bad_st += st0[(size_t)k * 128 + j * 32 + q] != st1[(size_t)k * 128 + j * 32 + q];
    printf("V0 (component, point) lanes : forward %6.0f cycles/stage, adjoint %6.0f cycles/stage\n",
           (double)o0.cyc_fwd / reps / NST, (double)o0.cyc_adj / reps / NST);
    printf("V1 (chain, point) lanes, components in registers: forward %6.0f, adjoint %6.0f cycles/stage\n",
           (double)o1.cyc_fwd / reps / NST, (double)o1.cyc_adj / reps / NST);
    printf("bit-equal results: %s (state/adjoint/Hv mismatches %d, stored d_t mismatches %d; |x_t| up to %.3g, Hv[0] = %.17g)\n",
           bad + bad_st == 0 ? "yes" : "NO", bad, bad_st, big, o0.hv[0]);
    return bad + bad_st == 0 ? 0 : 2;
}
