// Does a VALU fp64 instruction get cheaper when only some lanes of the wavefront are active?
// (development aid: the scalar quadcopter sweeps need 13 lanes at most)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
__global__ void k(long long *out, double *sink, int active) {
    double a = sink[threadIdx.x], b = sink[64 + threadIdx.x], c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3;
    long long t0 = 0, t1 = 0, t2 = 0;
    if ((int)threadIdx.x < active) {
        t0 = __builtin_readcyclecounter();
        REP64(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(c0) : "v"(b));)
        t1 = __builtin_readcyclecounter();
        REP16(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4"
                           : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(b));)
        t2 = __builtin_readcyclecounter();
    }
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; }
    sink[threadIdx.x] = c0 + c1 + c2 + c3;
}
int main() {
    long long *d_out, h[2];
    double *d_s, hs[128];
    for (int i = 0; i < 128; ++i) hs[i] = 1.0 + 1e-9 * i;
    (void)hipMalloc(&d_out, sizeof h); (void)hipMalloc(&d_s, sizeof hs);
    (void)hipMemcpy(d_s, hs, sizeof hs, hipMemcpyHostToDevice);
    for (int active : {64, 32, 16, 8, 1}) {
        for (int r = 0; r < 2; ++r) {
            hipLaunchKernelGGL(k, 1, 64, 0, 0, d_out, d_s, active);
            (void)hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
        }
        printf("%2d active lanes: dependent v_fma_f64 %.1f cycles, 4 independent chains %.1f cycles each\n", active,
               h[0] / 64.0, h[1] / 64.0);
    }
    return 0;
}
