// Probe: v_mfma_scale_f32_32x32x64_f8f6f4 with FP6 (E2M3) operands on gfx950 -- operand layout (which lane / 6-bit field is A[i][k], B[k][j]), the E8M0 scale
// operands (per lane? which byte?), and the issue rate against v_mfma_f32_32x32x16_f16.  (The value of every E2M3 code is checked end to end by gemm_mx_probe.hip.)
// build on the box: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/mx_mfma_probe tools/probes/mx_mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define FMT 2      // cbsz / blgp code of FP6 E2M3 (0 fp8 e4m3, 1 bf8, 2 fp6 e2m3, 3 bf6 e3m2, 4 fp4)

__device__ __forceinline__ void put6(v8i& v, int p, uint32_t code) {      // 6-bit field p of the 192-bit little-endian operand
    const int bit = 6 * p, w = bit >> 5, s = bit & 31;
    uint32_t* u = reinterpret_cast<uint32_t*>(&v);
    u[w] |= code << s;
    if (s > 26) u[w + 1] |= code >> (32 - s);
}
// one test per workgroup (one wave): A one-hot or all-ones, B one-hot or all-ones; the whole C tile is written out
//   mode 0: A one-hot at (la, pa), B all ones      -> which row i does lane la own
//   mode 1: A all ones, B one-hot at (lb, pb)      -> which column j does lane lb own
//   mode 2: A one-hot (la, pa), B one-hot (lb, pb) -> do they meet (same k)?  C[i(la)][j(lb)] = 1 iff yes
__global__ void __launch_bounds__(64) layout_kernel(int mode, const int* spec, float* out) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const int la = spec[4 * t], pa = spec[4 * t + 1], lb = spec[4 * t + 2], pb = spec[4 * t + 3];
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
    const uint32_t ONE = 0x08;      // E2M3 1.0: sign 0, exponent 01 (bias 1), mantissa 000
    if (mode == 1) { for (int p = 0; p < 32; p++) put6(a, p, ONE); } else if (lane == la) put6(a, pa, ONE);
    if (mode == 0) { for (int p = 0; p < 32; p++) put6(b, p, ONE); } else if (lane == lb) put6(b, pb, ONE);
    v16f c;
    for (int i = 0; i < 16; i++) c[i] = 0.0f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, FMT, FMT, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int r = 0; r < 16; r++) out[(size_t)t * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = c[r];
}
// scales: A all ones, B all ones (C = 64 everywhere at scale 1); lane `ls` gets scale word `wa` for A (others 0x7F7F7F7F), opsel_a as given by OPA
template <int OPA, int OPB> __global__ void __launch_bounds__(64) scale_kernel(int ls, uint32_t wa, uint32_t wb, float* out) {
    const int lane = threadIdx.x;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
    for (int p = 0; p < 32; p++) { put6(a, p, 0x08); put6(b, p, 0x08); }
    v16f c;
    for (int i = 0; i < 16; i++) c[i] = 0.0f;
    const int sa = lane == ls ? (int)wa : 0x7F7F7F7F, sb = lane == ls ? (int)wb : 0x7F7F7F7F;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, FMT, FMT, OPA, sa, OPB, sb);
    for (int r = 0; r < 16; r++) out[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = c[r];
}
// issue rate: 8 independent accumulators per wave, 4 waves per workgroup, WG workgroups per CU
template <int KIND> __global__ void __launch_bounds__(256) rate_kernel(int iters, float* sink, unsigned long long* ticks) {
    v8i a, b;
    for (int i = 0; i < 8; i++) { a[i] = 0x08208208 + threadIdx.x; b[i] = 0x08208208; }
    v8h ah, bh;
    for (int i = 0; i < 8; i++) { ah[i] = (_Float16)(1.0f + threadIdx.x); bh[i] = (_Float16)0.5f; }
    v16f c[8];
    for (int j = 0; j < 8; j++) for (int i = 0; i < 16; i++) c[j][i] = 0.0f;
    const unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (KIND == 0) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c[j], 0, 0, 0);
            else if (KIND == 1) c[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[j], 2, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            else if (KIND == 2) c[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            else c[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[j], 0, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
    }
    const unsigned long long t1 = wall_clock64();
    float s = 0.0f;
    for (int j = 0; j < 8; j++) for (int i = 0; i < 16; i++) s += c[j][i];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
    // ---- layout
    std::vector<int> spec; int nt;
    float* out; CK(hipMalloc(&out, (size_t)8192 * 1024 * 4));
    int* dspec; CK(hipMalloc(&dspec, 8192 * 16));
    std::vector<float> h((size_t)8192 * 1024);
    auto run = [&](int mode) {
        nt = (int)spec.size() / 4;
        CK(hipMemcpy(dspec, spec.data(), spec.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(layout_kernel, dim3(nt), dim3(64), 0, 0, mode, dspec, out); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), out, (size_t)nt * 4096, hipMemcpyDeviceToHost));
    };
    // mode 0: every lane of A, position 0 and 31
    spec.clear(); for (int l = 0; l < 64; l++) for (int p : {0, 31}) { spec.push_back(l); spec.push_back(p); spec.push_back(0); spec.push_back(0); }
    run(0);
    printf("A: lane -> row(s) with a non-zero C (B all ones), positions 0 and 31\n");
    for (int t = 0; t < nt; t++) { printf("  lane %2d pos %2d: rows", spec[4 * t], spec[4 * t + 1]); for (int i = 0; i < 32; i++) if (h[(size_t)t * 1024 + i * 32] != 0.0f) printf(" %d(=%g)", i, h[(size_t)t * 1024 + i * 32]); printf("\n"); if (t == 5) { printf("  ...\n"); t = nt - 7; } }
    int okA = 1; for (int t = 0; t < nt; t++) { const int l = spec[4 * t]; for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) if ((h[(size_t)t * 1024 + i * 32 + j] != 0.0f) != (i == (l & 31))) okA = 0; }
    printf("  A lane l owns row l & 31 (every column set, nothing else): %s\n", okA ? "yes" : "NO");
    spec.clear(); for (int l = 0; l < 64; l++) for (int p : {0, 31}) { spec.push_back(0); spec.push_back(0); spec.push_back(l); spec.push_back(p); }
    run(1);
    int okB = 1; for (int t = 0; t < nt; t++) { const int l = spec[4 * t + 2]; for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) if ((h[(size_t)t * 1024 + i * 32 + j] != 0.0f) != (j == (l & 31))) okB = 0; }
    printf("  B lane l owns column l & 31: %s\n", okB ? "yes" : "NO");
    // mode 2: A (lane 0 / 32, every position) against B (lane 0 / 32, every position)
    for (int la : {0, 32}) {
        spec.clear(); for (int pa = 0; pa < 32; pa++) for (int lb : {0, 32}) for (int pb = 0; pb < 32; pb++) { spec.push_back(la); spec.push_back(pa); spec.push_back(lb); spec.push_back(pb); }
        run(2);
        printf("k pairing, A lane %d: A position -> B (lane, position) it meets:", la);
        int ident = 1;
        for (int pa = 0; pa < 32; pa++) {
            int found = 0;
            for (int q = 0; q < 64; q++) { const int t = pa * 64 + q; if (h[(size_t)t * 1024] != 0.0f) { if (!(spec[4 * t + 2] == la && spec[4 * t + 3] == pa)) { ident = 0; printf(" [%d -> (%d,%d)]", pa, spec[4 * t + 2], spec[4 * t + 3]); } found++; } }
            if (found != 1) { ident = 0; printf(" [%d: %d matches]", pa, found); }
        }
        printf(" %s\n", ident ? "identity: A (lane, p) meets B (same half, same p)" : "");
    }
    // ---- scales
    auto show = [&](const char* what) {
        CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), out, 4096, hipMemcpyDeviceToHost));
        printf("  %s:", what);
        int shown = 0;
        for (int i = 0; i < 32 && shown < 6; i++) for (int j = 0; j < 32 && shown < 6; j++) if (h[i * 32 + j] != 64.0f) { printf(" C[%d][%d]=%g", i, j, h[i * 32 + j]); shown++; }
        int cnt = 0; for (int i = 0; i < 1024; i++) if (h[i] != 64.0f) cnt++;
        printf("  (%d entries differ from 64)\n", cnt);
    };
    printf("scales (all operands 1.0: C = 64 at scale 1; one lane's scale word changed):\n");
    hipLaunchKernelGGL((scale_kernel<0, 0>), dim3(1), dim3(64), 0, 0, 5, 0x7F7F7F80u, 0x7F7F7F7Fu, out); show("A scale, lane 5, byte 0 = 128, opsel 0");
    hipLaunchKernelGGL((scale_kernel<0, 0>), dim3(1), dim3(64), 0, 0, 5, 0x7F7F807Fu, 0x7F7F7F7Fu, out); show("A scale, lane 5, byte 1 = 128, opsel 0");
    hipLaunchKernelGGL((scale_kernel<1, 0>), dim3(1), dim3(64), 0, 0, 5, 0x7F7F807Fu, 0x7F7F7F7Fu, out); show("A scale, lane 5, byte 1 = 128, opsel 1");
    hipLaunchKernelGGL((scale_kernel<2, 0>), dim3(1), dim3(64), 0, 0, 5, 0x7F807F7Fu, 0x7F7F7F7Fu, out); show("A scale, lane 5, byte 2 = 128, opsel 2");
    hipLaunchKernelGGL((scale_kernel<3, 0>), dim3(1), dim3(64), 0, 0, 5, 0x807F7F7Fu, 0x7F7F7F7Fu, out); show("A scale, lane 5, byte 3 = 128, opsel 3");
    hipLaunchKernelGGL((scale_kernel<0, 0>), dim3(1), dim3(64), 0, 0, 37, 0x7F7F7F80u, 0x7F7F7F7Fu, out); show("A scale, lane 37, byte 0 = 128, opsel 0");
    hipLaunchKernelGGL((scale_kernel<0, 0>), dim3(1), dim3(64), 0, 0, 5, 0x7F7F7F7Fu, 0x7F7F7F7Du, out); show("B scale, lane 5, byte 0 = 125, opsel 0");
    hipLaunchKernelGGL((scale_kernel<0, 0>), dim3(1), dim3(64), 0, 0, 37, 0x7F7F7F7Fu, 0x7F7F7F7Du, out); show("B scale, lane 37, byte 0 = 125, opsel 0");
    // ---- rate
    float* sink; unsigned long long* tk; CK(hipMalloc(&sink, 1024 * 256 * 4)); CK(hipMalloc(&tk, 1024 * 8));
    const char* names[4] = {"v_mfma_f32_32x32x16_f16 (16 k)", "f8f6f4 fp6 x fp6 (64 k)", "f8f6f4 fp8 x fp8 (64 k)", "f8f6f4 fp8 x fp6 (64 k)"};
    for (int wg = 1; wg <= 2; wg++)
        for (int kind = 0; kind < 4; kind++) {
            const int iters = 4000;
            for (int rep = 0; rep < 2; rep++) {
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventRecord(e0, 0));
                if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(256 * wg), dim3(256), 0, 0, iters, sink, tk);
                if (kind == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(256 * wg), dim3(256), 0, 0, iters, sink, tk);
                if (kind == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(256 * wg), dim3(256), 0, 0, iters, sink, tk);
                if (kind == 3) hipLaunchKernelGGL(rate_kernel<3>, dim3(256 * wg), dim3(256), 0, 0, iters, sink, tk);
                CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep == 1) {
                    const double k = kind == 0 ? 16 : 64, flop = 2.0 * 32 * 32 * k * 8.0 * iters * 4 * 256 * wg;
                    printf("rate %-32s %d wave(s) per SIMD: %.3f ms, %.0f TFLOP/s, %.1f ns per MFMA per SIMD\n", names[kind], wg, ms, flop / (ms * 1e-3) / 1e12, ms * 1e6 / (8.0 * iters * wg));
                }
            }
        }
    return 0;
}
