// Semantics of the DPP forms the row-split tape product relies on (gfx950):
//   v_mov_b32_dpp row_ror:n with a row_mask (only the selected 16-lane row is written, the others keep `old`),
//   direction of row_ror, v_permlane16_swap / v_permlane32_swap of a register with itself.
//   hipcc -O3 --offload-arch=gfx950 -o dpp_row_rotate_test dpp_row_rotate_test.hip && ./dpp_row_rotate_test
#include <hip/hip_runtime.h>
#include <cstdio>

template <int N, int ROWMASK>
__device__ int ror(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, 0x120 + N, ROWMASK, 0xF, false);
}
__global__ void k(int *out) {
    const int l = threadIdx.x;
    int v = l;                    // value = lane id
    int r = v;
    r = ror<1, 0x1>(r, v);        // row 0 only
    r = ror<15, 0x4>(r, v);       // row 2 only
    r = ror<14, 0x8>(r, v);       // row 3 only
    out[l] = r;
    auto s16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    out[64 + l]  = s16[0];
    out[128 + l] = s16[1];
    auto s32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    out[192 + l] = s32[0];
    out[256 + l] = s32[1];
}
int main() {
    int *d, h[320];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[5] = {"row_ror 1|-|15|14 by row", "swap16 [0]", "swap16 [1]", "swap32 [0]", "swap32 [1]"};
    for (int b = 0; b < 5; ++b) {
        printf("%s:\n", names[b]);
        for (int r = 0; r < 4; ++r) {
            printf("  row %d:", r);
            for (int i = 0; i < 16; ++i)
                printf(" %2d", h[b * 64 + r * 16 + i]);
            printf("\n");
        }
    }
    return 0;
}
