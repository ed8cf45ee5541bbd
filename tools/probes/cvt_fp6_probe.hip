// Probe: v_cvt_scalef32_pk32_f16_fp6 on gfx950 -- 32 FP6 (E2M3) codes in 6 VGPRs -> 32 f16 in 16 VGPRs, times an f32 scale.
// Questions: (1) which bits of the source hold element i, and which output half-word receives it; (2) the value of every code; (3) is the whole f32 scale applied
// (mantissa too) and is the product rounded once to f16 (RNE)?  (4) issue cost per wave-instruction against v_pk_fma_f16.
// build on the box: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/cvt_fp6_probe tools/probes/cvt_fp6_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef unsigned int v6u __attribute__((ext_vector_type(6)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void conv_kernel(const v6u* in, const float* sc, v32h* out) { out[threadIdx.x + blockIdx.x * blockDim.x] = __builtin_amdgcn_cvt_scalef32_pk32_f16_fp6(in[threadIdx.x + blockIdx.x * blockDim.x], sc[threadIdx.x + blockIdx.x * blockDim.x]); }

// issue cost: REP conversions per iteration on independent sources, results folded so that nothing is dead
template <int MODE> __global__ void __launch_bounds__(256) cost_kernel(const v6u* in, float scale, int iters, unsigned long long* cycles, float* sink) {
    v6u s0 = in[threadIdx.x], s1 = in[threadIdx.x + 256];
    v2h acc = {0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
            const v32h a = __builtin_amdgcn_cvt_scalef32_pk32_f16_fp6(s0, scale), b = __builtin_amdgcn_cvt_scalef32_pk32_f16_fp6(s1, scale);
#pragma unroll
            for (int i = 0; i < 32; i += 16) acc += v2h{a[i], b[i + 1]};
            s0[0] += 1; s1[3] += 1;
        } else {
            v2h x = __builtin_bit_cast(v2h, s0[0]), y = __builtin_bit_cast(v2h, s1[0]);
#pragma unroll
            for (int i = 0; i < 16; i++) { x = __builtin_elementwise_fma(x, v2h{(_Float16)1.0009765625f, (_Float16)1.0009765625f}, acc); y = __builtin_elementwise_fma(y, v2h{(_Float16)0.99951171875f, (_Float16)0.99951171875f}, acc); }
            acc += x + y;
            s0[0] += 1; s1[0] += 1;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    sink[threadIdx.x + blockIdx.x * 256] = (float)acc.x + (float)acc.y;
}

static float h2f(uint16_t h) { _Float16 x; memcpy(&x, &h, 2); return (float)x; }
int main() {
    const int n = 256;
    v6u* in; float* sc; v32h* out;
    CK(hipMalloc(&in, n * sizeof(v6u) * 2)); CK(hipMalloc(&sc, n * 4)); CK(hipMalloc(&out, n * sizeof(v32h)));
    std::vector<uint32_t> hin(n * 8, 0);      // sizeof(v6u) is 32: the 6-word vector is padded to 8
    std::vector<float> hs(n, 1.0f); std::vector<uint16_t> ho(n * 32);
    // (1) lanes 0..31: code 0x0C (+2.0 in E2M3 if bias 1: e = 2 -> 2^(2-1) * 1.0) at 6-bit position p = lane
    for (int p = 0; p < 32; p++) { const int bit = 6 * p; hin[p * 8 + bit / 32] |= 0x0Cu << (bit % 32); if (bit % 32 > 26) hin[p * 8 + bit / 32 + 1] |= 0x0Cu >> (32 - bit % 32); }
    // (2) lanes 64..127: code c = lane - 64 at position 0
    for (int c = 0; c < 64; c++) hin[(64 + c) * 8] = c;
    // (3) lanes 128..191: code 0x3F... use code 0x0D (2.25) and 0x1F (7.5) with awkward scales
    const float scales[8] = {1.0f, 3.0f, 0.75f, 1.17f, 0.0009765625f * 1.3f, 1.0f / 3.0f, 0.007843f, 1.99f};
    for (int i = 0; i < 64; i++) { hin[(128 + i) * 8] = (i & 1) ? 0x1F : 0x0D; hin[(128 + i) * 8] |= ((i >> 1) & 31) << 6; hs[128 + i] = scales[i & 7] * (1.0f + (i >> 3) * 0.0625f); }
    CK(hipMemcpy(in, hin.data(), n * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(sc, hs.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(conv_kernel, dim3(1), dim3(n), 0, 0, in, sc, out); CK(hipDeviceSynchronize());
    CK(hipMemcpy(ho.data(), out, n * 64, hipMemcpyDeviceToHost));
    printf("(1) code 0x0C at 6-bit position p -> non-zero output elements:\n");
    for (int p = 0; p < 32; p++) { printf("  p %2d:", p); for (int e = 0; e < 32; e++) if (ho[p * 32 + e]) printf(" out[%d] = %g", e, h2f(ho[p * 32 + e])); printf("\n"); }
    printf("(2) value of every code at position 0 (scale 1):\n  ");
    for (int c = 0; c < 64; c++) printf("%02x:%g ", c, h2f(ho[(64 + c) * 32]));
    printf("\n(3) scale handling: out vs f16(value * scale) rounded once from double\n");
    int bad = 0;
    for (int i = 0; i < 64; i++) {
        const double v = (i & 1) ? 7.5 : 2.25;
        const _Float16 want = (_Float16)(v * (double)hs[128 + i]);       // double product is exact (4 x 24 bits), one rounding to f16
        uint16_t wb; memcpy(&wb, &want, 2);
        if (wb != ho[(128 + i) * 32]) { bad++; if (bad < 10) printf("  scale %.9g value %g: got %04x (%g) want %04x (%g)\n", hs[128 + i], v, ho[(128 + i) * 32], h2f(ho[(128 + i) * 32]), wb, (float)want); }
    }
    printf("  %d of 64 differ from the once-rounded product\n", bad);
    // (4) cost
    unsigned long long* cyc; float* sink; CK(hipMalloc(&cyc, 1024 * 8)); CK(hipMalloc(&sink, 1024 * 256 * 4));
    for (int wg = 1; wg <= 2; wg++)
        for (int mode = 0; mode < 2; mode++) {
            const int iters = 20000;
            for (int rep = 0; rep < 2; rep++) {
                if (mode == 0) hipLaunchKernelGGL(cost_kernel<0>, dim3(256 * wg), dim3(256), 0, 0, in, 1.5f, iters, cyc, sink);
                else hipLaunchKernelGGL(cost_kernel<1>, dim3(256 * wg), dim3(256), 0, 0, in, 1.5f, iters, cyc, sink);
                CK(hipDeviceSynchronize());
            }
            unsigned long long c0; CK(hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost));
            printf("(4) %s, %d workgroup(s) of 4 waves per CU: %.1f shader-clock ticks per iteration (%s)\n", mode ? "32 v_pk_fma_f16" : "2 v_cvt_scalef32_pk32_f16_fp6", wg, (double)c0 / iters,
                   mode ? "32 instructions" : "2 instructions + 2 pk_add");
        }
    return 0;
}
