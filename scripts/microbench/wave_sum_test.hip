// 64-lane all-reduce sum without the LDS crossbar: v_permlane32_swap / v_permlane16_swap across
// rows, DPP moves inside a row.  Checks bit-equality with the xor butterfly (masks 32, 16, 8, 4, 2,
// 1: the reduction order of the arithmetic specification) done with ds_bpermute, and times both.
// (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ double lane_read(double v, int src) {
    int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
    int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sum_bpermute(double v) {
    for (int m = 32; m >= 1; m >>= 1)
        v = v + lane_read(v, (threadIdx.x & 63) ^ m);
    return v;
}
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sum_dpp(double v) {
    {   // halves: own + partner = p + q on every row
        int lo = __double2loint(v), hi = __double2hiint(v);
        auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    }
    {   // rows 0<->1, 2<->3
        int lo = __double2loint(v), hi = __double2hiint(v);
        auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    }
    v = v + dpp<0x128>(v);                                        // row_ror:8 = lane ^ 8
    {   // lane ^ 4: both moves with all lanes active, then the select (a DPP move inside a
        // divergent branch would read inactive source lanes)
        double up = dpp<0x104>(v), down = dpp<0x114>(v); // row_shl:4, row_shr:4
        v = v + ((threadIdx.x & 4) ? down : up);
    }
    v = v + dpp<0x4E>(v);                                         // quad_perm [2,3,0,1]
    v = v + dpp<0xB1>(v);                                         // quad_perm [1,0,3,2]
    return v;
}
__global__ void k(const double *in, double *o, long long *t) {
    double v = in[blockIdx.x * 64 + threadIdx.x];
    double a = sum_bpermute(v), b = sum_dpp(v);
    o[blockIdx.x * 128 + threadIdx.x] = a;
    o[blockIdx.x * 128 + 64 + threadIdx.x] = b;
    if (blockIdx.x == 0) {
        long long t0 = __builtin_readcyclecounter();
        double x = v;
        for (int i = 0; i < 32; ++i) x = sum_bpermute(x) * 0.015625;
        long long t1 = __builtin_readcyclecounter();
        double y = v;
        for (int i = 0; i < 32; ++i) y = sum_dpp(y) * 0.015625;
        long long t2 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) { t[0] = (t1 - t0) / 32; t[1] = (t2 - t1) / 32; }
        o[threadIdx.x] += 0.0 * (x + y);
    }
}
int main() {
    const int B = 256;
    double *h = (double *)malloc(B * 64 * 8), *ho = (double *)malloc(B * 128 * 8), *d, *dout;
    long long *dt, ht[2];
    srand(1);
    for (int i = 0; i < B * 64; ++i) h[i] = (rand() / (double)RAND_MAX - 0.5) * (1 + (i % 7 == 0) * 1e6);
    (void)hipMalloc(&d, B * 64 * 8); (void)hipMalloc(&dout, B * 128 * 8); (void)hipMalloc(&dt, 16);
    (void)hipMemcpy(d, h, B * 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, B, 64, 0, 0, d, dout, dt);
    (void)hipMemcpy(ho, dout, B * 128 * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(ht, dt, 16, hipMemcpyDeviceToHost);
    int bad = 0, nonuniform = 0;
    for (int b = 0; b < B; ++b)
        for (int l = 0; l < 64; ++l) {
            bad += ho[b * 128 + l] != ho[b * 128 + 64 + l];
            nonuniform += ho[b * 128 + 64 + l] != ho[b * 128 + 64];
        }
    printf("mismatches vs the bpermute butterfly: %d, lanes disagreeing: %d\n", bad, nonuniform);
    printf("cycles per all-reduce: bpermute %lld, dpp+permlane %lld\n", ht[0], ht[1]);
    return bad || nonuniform;
}
