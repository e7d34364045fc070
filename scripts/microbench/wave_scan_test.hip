// The wavefront scans of the quadcopter-family products and adjoint sweep (gfx950): wave_running_sum,
// lane_below (wave_shr:1), wave_affine_scan of wave_math.hip.h against a host restatement of the grouping
// the oracle specifies (oracle/quad_models.hpp: prefix_scan, affine_scan) - bit equality, and cycles.
// Pins the DPP forms they rely on: row_shr:1/2/4/8 with bound_ctrl (a lane without a source reads 0) or
// without it (the lane keeps `old`), row_bcast:15 with row_mask 0xA, row_bcast:31 with row_mask 0xC.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -I ../../pantr-cdc2023-experiments_amd/csrc -o wave_scan_test wave_scan_test.hip && ./wave_scan_test
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include "wave_math.hip.h"
using namespace pantr;

__global__ void k(const double *in, double *out, long long *cyc) {
    const int l = threadIdx.x;
    const double x = in[l];
    // the timed value feeds the next clock read through a dependency the scheduler cannot move
    long long t0 = wall_clock64();
    double s = x + (double)(t0 & 0);
    for (int r = 0; r < 64; ++r)
        s = wave_running_sum(s) * 0x1p-6;
    long long t1 = wall_clock64() + (long long)(s != s);
    s = wave_running_sum(x);
    out[l]      = s;
    out[64 + l] = lane_below(s);
    Affine2 m{in[64 + l], in[128 + l], in[192 + l], in[256 + l], in[320 + l], in[384 + l]};
    long long t2 = wall_clock64();
    Affine2 P = m;
    P.b0 += (double)(t2 & 0);
    for (int r = 0; r < 64; ++r) {
        P = wave_affine_scan(P);
        P.a00 *= 0x1p-20; P.a01 *= 0x1p-20; P.a10 *= 0x1p-20; P.a11 *= 0x1p-20;
    }
    long long t3 = wall_clock64() + (long long)(P.b0 != P.b0);
    P = wave_affine_scan(m);
    out[128 + l] = P.a00; out[192 + l] = P.a01; out[256 + l] = P.a10; out[320 + l] = P.a11;
    out[384 + l] = P.b0;  out[448 + l] = P.b1;
    if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t3 - t2; }
}

struct A2 { double a00, a01, a10, a11, b0, b1; };
static A2 compose(const A2 &l, const A2 &e) {
    return {std::fma(l.a01, e.a10, l.a00 * e.a00), std::fma(l.a01, e.a11, l.a00 * e.a01),
            std::fma(l.a11, e.a10, l.a10 * e.a00), std::fma(l.a11, e.a11, l.a10 * e.a01),
            std::fma(l.a01, e.b1, std::fma(l.a00, e.b0, l.b0)), std::fma(l.a11, e.b1, std::fma(l.a10, e.b0, l.b1))};
}
int main() {
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(-1, 1);
    double in[448], out[512], ref[64];
    for (double &v : in) v = U(rng);
    double *din, *dout; long long *dc, cyc[2];
    hipMalloc(&din, sizeof in); hipMalloc(&dout, sizeof out); hipMalloc(&dc, sizeof cyc);
    hipMemcpy(din, in, sizeof in, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, dc);
    hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
    hipMemcpy(cyc, dc, sizeof cyc, hipMemcpyDeviceToHost);
    // host: the grouping of prefix_scan / affine_scan
    double x[64], t[64];
    memcpy(x, in, sizeof x);
    for (int d = 1; d <= 8; d *= 2) { for (int i = 0; i < 64; ++i) t[i] = x[i] + ((i & 15) >= d ? x[i - d] : 0.0); memcpy(x, t, sizeof x); }
    for (int i = 0; i < 64; ++i) t[i] = x[i] + ((i & 16) ? x[(i & ~15) - 1] : 0.0); memcpy(x, t, sizeof x);
    for (int i = 0; i < 64; ++i) t[i] = x[i] + ((i & 32) ? x[31] : 0.0); memcpy(ref, t, sizeof ref);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        bad += memcmp(&ref[i], &out[i], 8) != 0;
        const double below = i ? ref[i - 1] : 0.0;
        bad += memcmp(&below, &out[64 + i], 8) != 0;
    }
    A2 m[64], u[64]; const A2 id{1, 0, 0, 1, 0, 0};
    for (int i = 0; i < 64; ++i) m[i] = {in[64 + i], in[128 + i], in[192 + i], in[256 + i], in[320 + i], in[384 + i]};
    for (int d = 1; d <= 8; d *= 2) { for (int i = 0; i < 64; ++i) u[i] = compose(m[i], (i & 15) >= d ? m[i - d] : id); memcpy(m, u, sizeof m); }
    for (int i = 0; i < 64; ++i) u[i] = compose(m[i], (i & 16) ? m[(i & ~15) - 1] : id); memcpy(m, u, sizeof m);
    for (int i = 0; i < 64; ++i) u[i] = compose(m[i], (i & 32) ? m[31] : id); memcpy(m, u, sizeof m);
    for (int i = 0; i < 64; ++i) {
        const double g[6] = {out[128 + i], out[192 + i], out[256 + i], out[320 + i], out[384 + i], out[448 + i]};
        bad += memcmp(g, &m[i], sizeof g) != 0;
    }
    printf("running sum + lane_below + affine scan vs the oracle's grouping: %s (%d mismatches)\n", bad ? "DIFFERENT" : "bit-identical", bad);
    printf("one wavefront, 64 dependent repetitions, 100 MHz clock: wave_running_sum %.0f ns each, wave_affine_scan %.0f ns each\n",
           cyc[0] * 10.0 / 64, cyc[1] * 10.0 / 64);
    return bad != 0;
}
