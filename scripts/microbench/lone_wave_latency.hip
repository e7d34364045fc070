// Lone-wavefront instruction cost on gfx950 (development aid): shader cycles per instruction for the
// patterns the solver kernels are made of.  One wavefront, s_memtime around unrolled asm blocks.
//   hipcc -O3 --offload-arch=gfx950 -o lone_wave_latency lone_wave_latency.hip && ./lone_wave_latency
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

__global__ void bench(long long *out, double *sink) {
    double a = sink[threadIdx.x], b = sink[64 + threadIdx.x], c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3;
    int i0 = threadIdx.x, i1 = i0 + 1;
    long long t[16];
    int n = 0;
    auto tick = [&]() {
        long long v;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");
        return v;
    };
    auto cyc = [&]() { return (long long)__builtin_readcyclecounter(); };
    // 0: dependent fma chain
    t[n++] = cyc();
    REP64(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(c0) : "v"(b));)
    t[n++] = cyc();
    // 1: four independent chains, interleaved
    REP16(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4"
                       : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(b));)
    t[n++] = cyc();
    // 2: dependent fma chain with an SGPR-pair operand
    double su = __builtin_bit_cast(double, (long long)__builtin_amdgcn_readfirstlane((int)a) | (1023ll << 52));
    REP64(asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(c0) : "s"(su));)
    t[n++] = cyc();
    // 3: v_readlane pairs (independent of each other)
    int s0, s1;
    REP64(asm volatile("v_readlane_b32 %0, %2, 3\n v_readlane_b32 %1, %3, 3" : "=s"(s0), "=s"(s1) : "v"(i0), "v"(i1));)
    t[n++] = cyc();
    // 4: fma -> readlane x2 -> fma (the hand-off pattern)
    REP16(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_fma_f64 %2, s[20:21], %2, %2\n"
                       "v_mov_b32 %0, %3\n" : "+v"(i0), "+v"(i1), "+v"(c0) : "v"(i1) : "s20", "s21");)
    t[n++] = cyc();
    // 5: v_mov_b32 dpp quad_perm dependent chain
    REP64(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(i0));)
    t[n++] = cyc();
    // 6: ds_bpermute dependent chain
    REP64(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(i0) : "v"(i1));)
    t[n++] = cyc();
    // 7: v_add_f64 dependent
    REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(c0) : "v"(b));)
    t[n++] = cyc();
    // 8: v_cndmask dependent (32-bit VALU)
    REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(i0) : "v"(i1));)
    t[n++] = cyc();
    // 9: independent 32-bit VALU
    REP64(asm volatile("v_add_u32 %0, %1, %1" : "=v"(i0) : "v"(i1));)
    t[n++] = cyc();
    // 10: LDS write -> read round trip
    __shared__ double l[64];
    REP16(asm volatile("ds_write_b64 %1, %0\n ds_read_b64 %0, %2\n s_waitcnt lgkmcnt(0)" : "+v"(c0) : "v"(threadIdx.x * 8), "v"((threadIdx.x ^ 1) * 8) : "memory");)
    t[n++] = cyc();
    if (threadIdx.x == 0)
        for (int i = 0; i + 1 < n; ++i)
            out[i] = t[i + 1] - t[i];
    sink[threadIdx.x] = c0 + c1 + c2 + c3 + i0 + s0 + s1 + l[0];
    (void)tick;
}

int main() {
    long long *d_out, h[16] = {0};
    double *d_s, hs[128];
    for (int i = 0; i < 128; ++i) hs[i] = 1.0 + 1e-9 * i;
    (void)hipMalloc(&d_out, sizeof h);
    (void)hipMalloc(&d_s, sizeof hs);
    (void)hipMemcpy(d_s, hs, sizeof hs, hipMemcpyHostToDevice);
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(bench, 1, 64, 0, 0, d_out, d_s);
        (void)hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    }
    const char *name[] = {"dependent v_fma_f64", "4 independent v_fma_f64 chains", "dependent v_fma_f64, SGPR operand",
                          "v_readlane_b32 (pairs)", "readlane x2 + fma + mov hand-off (per group of 4)", "dependent v_mov_b32 dpp",
                          "dependent ds_bpermute + wait", "dependent v_add_f64", "dependent v_add_u32", "independent v_add_u32",
                          "LDS write+read+wait round trip"};
    const int cnt[] = {64, 64, 64, 128, 16, 64, 64, 64, 64, 64, 16};
    for (int i = 0; i < 11; ++i)
        printf("%-52s %7.1f cycles each\n", name[i], (double)h[i] / cnt[i]);
    return 0;
}
