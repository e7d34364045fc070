// Cost of the cross-row primitives of the row layouts on gfx950 (lone wavefront, shader cycles):
// v_permlane16_swap / v_permlane32_swap (dependent and independent), the row shift pair of a 64-bit
// value, and the three-swap component gather with its register copies as the compiler emits it.
//   hipcc -O3 --offload-arch=gfx950 -o permlane_swap_latency permlane_swap_latency.hip && ./permlane_swap_latency
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

__global__ void bench(long long *out, int *sink) {
    int a = sink[threadIdx.x], b = sink[64 + threadIdx.x], c = a + 1, d = b + 1;
    long long t[8];
    int n = 0;
    auto cyc = [&]() { return (long long)__builtin_readcyclecounter(); };
    t[n++] = cyc();
    REP64(asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));)                     // 0: dependent
    t[n++] = cyc();
    REP64(asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));)                     // 1: dependent
    t[n++] = cyc();
    REP16(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3"
                       : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)                             // 2: two interleaved chains
    t[n++] = cyc();
    REP64(asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                       "v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a), "+v"(b));) // 3: row shift pair
    t[n++] = cyc();
    REP16(asm volatile("v_mov_b32 %2, %0\n v_mov_b32 %3, %1\n v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n"
                       "v_mov_b32 %2, %0\n v_mov_b32 %3, %1\n v_permlane32_swap_b32 %0, %2\n v_permlane32_swap_b32 %1, %3"
                       : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)                             // 4: copy + swap groups (64-bit value, 2 levels)
    t[n++] = cyc();
    if (threadIdx.x == 0)
        for (int i = 0; i + 1 < n; ++i)
            out[i] = t[i + 1] - t[i];
    sink[threadIdx.x] = a + b + c + d;
}

int main() {
    long long *d_out, h[8] = {0};
    int *d_s, hs[128];
    for (int i = 0; i < 128; ++i) hs[i] = i;
    (void)hipMalloc(&d_out, sizeof h);
    (void)hipMalloc(&d_s, sizeof hs);
    (void)hipMemcpy(d_s, hs, sizeof hs, hipMemcpyHostToDevice);
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(bench, 1, 64, 0, 0, d_out, d_s);
        (void)hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    }
    const char *name[] = {"dependent v_permlane32_swap", "dependent v_permlane16_swap", "two interleaved swap chains (per instruction)",
                          "row_shr:1 pair (per instruction)", "copy x2 + swap x2, two levels (per group of 8)"};
    const int cnt[] = {64, 64, 64, 128, 16};
    for (int i = 0; i < 5; ++i)
        printf("%-52s %7.1f cycles each\n", name[i], (double)h[i] / cnt[i]);
    return 0;
}
