// Does gfx950 take v_fmac_f64 with a row_newbcast DPP source?  (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *o) {
    double x = 1.0 + threadIdx.x, a = 10.0, acc = 0.5;
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(a));
    double m;
    m = 0; // (v_mul_f64 has no DPP form on gfx950; v_fmac_f64 does)
    // fused?  (1+e)^2 - 1 = 2e + e^2 keeps the e^2 term only with a single rounding
    double e1 = 1.0 + 9.313225746154785e-10, fz = -1.0, fr = -1.0;
    asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %1, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(fz) : "v"(e1));
    fr = __builtin_fma(e1, e1, fr);
    if (threadIdx.x == 0) { o[193] = fz; o[194] = fr; }
    long long t0 = __builtin_readcyclecounter();
    double c0 = acc, c1 = acc, c2 = acc, c3 = acc;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        asm volatile("v_fmac_f64_dpp %0, %4, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f64_dpp %1, %4, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f64_dpp %2, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f64_dpp %3, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x), "v"(a));
    long long t1 = __builtin_readcyclecounter();
    o[threadIdx.x] = acc; o[64 + threadIdx.x] = m; o[128 + threadIdx.x] = c0 + c1 + c2 + c3;
    if (threadIdx.x == 0) o[192] = (double)(t1 - t0) / 64;
}
int main() {
    double *d, h[195];
    (void)hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("fmac (expect 40.5 on lanes 0-15, 200.5 on 16-31 ...): %g %g %g %g\n", h[0], h[7], h[16], h[40]);
    printf("fused check: dpp %.17g builtin_fma %.17g %s\n", h[193], h[194], h[193] == h[194] ? "same" : "DIFFERENT");
    printf("4-way independent v_fmac_f64_dpp: %.1f cycles each\n", h[192]);
    return 0;
}
