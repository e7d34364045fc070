// Semantics of the immediate offset of global_load_lds_dwordx4 on gfx950 (development aid): is it added to the
// global address, to the LDS address, or to both?  (WaveProblem's prefetches use it to request the consecutive
// 1 KiB rows of a stage with one address register pair and one M0 value.)
//   hipcc -O3 --offload-arch=gfx950 -o lds_dma_offset_test lds_dma_offset_test.hip && ./lds_dma_offset_test
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const int *src, int *out) {
    __shared__ __attribute__((aligned(16))) int lds[2048]; // 8 KiB
    for (int i = threadIdx.x; i < 2048; i += 64)
        lds[i] = -1;
    __syncthreads();
    const char *g = reinterpret_cast<const char *>(src) + threadIdx.x * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds, 16, 1024, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64)
        out[i] = lds[i];
}
int main() {
    int *s, *o, hs[4096], ho[2048];
    for (int i = 0; i < 4096; ++i) hs[i] = i; // word i of the source holds i
    (void)hipMalloc(&s, sizeof hs); (void)hipMalloc(&o, sizeof ho);
    (void)hipMemcpy(s, hs, sizeof hs, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, s, o);
    (void)hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    int first = -1, last = -1;
    for (int i = 0; i < 2048; ++i) if (ho[i] != -1) { if (first < 0) first = i; last = i; }
    printf("offset:1024 -> LDS words [%d, %d] written (byte offset %d), first word holds source word %d (source byte offset %d)\n",
           first, last, first * 4, ho[first], ho[first] * 4);
    printf("%s\n", first * 4 == 1024 && ho[first] * 4 == 1024 ? "the immediate offset is added to BOTH addresses"
           : first == 0 && ho[first] * 4 == 1024 ? "the immediate offset is added to the GLOBAL address only"
           : first * 4 == 1024 && ho[first] == 0 ? "the immediate offset is added to the LDS address only" : "unexpected");
    return 0;
}
