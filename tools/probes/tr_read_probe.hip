// Probe: what does ds_read_b64_tr_b16 (gfx950) return?  LDS holds u16 value = element index; lane l passes byte address 8 l (its "own" elements 4 l .. 4 l + 3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)((__attribute__((address_space(3))) char*)lds + 8 * threadIdx.x));
    *reinterpret_cast<h4*>(out + threadIdx.x * 4) = v;
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) { printf("lane %2d:", l); for (int j = 0; j < 4; j++) printf(" %4d (lane %2d elem %d)", h[l * 4 + j], h[l * 4 + j] / 4, h[l * 4 + j] % 4); printf("\n"); }
    return 0;
}
