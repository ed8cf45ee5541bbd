// Probe: shader clock under matrix load.  s_memtime (clock64) against s_memrealtime (wall_clock64, 100 MHz) around a loop of dependent / independent MFMAs on every CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) k(unsigned long long* out, int iters, float* sink) {
    v16f a0, a1, a2, a3; v8h x, y;
    for (int i = 0; i < 16; i++) { a0[i] = 0; a1[i] = 1; a2[i] = 2; a3[i] = 3; }
    for (int i = 0; i < 8; i++) { x[i] = (_Float16)(threadIdx.x * 0.001f); y[i] = (_Float16)0.5f; }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a3, 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
    if (a0[0] + a1[1] + a2[2] + a3[3] == 12345.0f) *sink = 1.0f;
}
int main() {
    unsigned long long* d; float* s; hipMalloc(&d, 16); hipMalloc(&s, 4);
    for (int blocks : {1, 256, 1024}) {
        const int iters = 20000;
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, s);
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        const double us = h[1] * 0.01, mfma_per_wave = iters * 4.0;
        printf("blocks %4d (4 waves each): %.0f us, s_memtime ticks %.0f (%.1f MHz if it were the shader clock), %.1f cycles per MFMA per wave at 2.4 GHz nominal, %.2f us-ns per MFMA: %.1f ns\n",
               blocks, us, (double)h[0], h[0] / us, us * 2400.0 / mfma_per_wave, 0.0, us * 1000.0 / mfma_per_wave);
    }
    return 0;
}
