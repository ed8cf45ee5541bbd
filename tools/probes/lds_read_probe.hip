// Probe: cycles per wave-wide LDS fragment read (ds_read_b128 / ds_read_b64) for the address patterns of kr_prefill_ring.hip and some alternatives: is the
// source-side XOR swizzle conflict-free on gfx950?  One workgroup; 1 or 8 waves reading concurrently; 256 reads per wave, s_memtime around them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ int pattern_addr(int p, int lane) {
    const int n31 = lane & 31, kh = lane >> 5;
    switch (p) {
    case 0: return lane * 16;                                                        // linear: the ideal
    case 1: return n31 * 128 + (((2 * kh) ^ ((n31 >> 1) & 7)) * 16);                 // ring A: 128-byte rows, chunk ^ (row >> 1 & 7)
    case 2: return n31 * 128 + kh * 32;                                              // 128-byte rows, no swizzle
    case 3: return n31 * 128 + (((2 * kh) ^ (n31 & 7)) * 16);                        // chunk ^ (row & 7)
    case 4: return n31 * 144 + kh * 32;                                              // padded rows (+16 B)
    case 5: { const int q = n31 >> 3, cc = n31 & 7, sw = 2 * (cc >> 1) + (q & 1); return q * 1024 + cc * 128 + ((kh ^ sw) * 16); }   // ring B (b64 / b128)
    case 6: { const int q = n31 >> 3, cc = n31 & 7; return q * 1024 + cc * 128 + kh * 16; }                                          // B records, no swizzle
    case 7: return n31 * 528 + kh * 32;                                              // the register-staged kernel's A rows (528-byte pitch)
    case 8: return n31 * 136 + kh * 16;                                              // the register-staged kernel's B columns (136-byte pitch), b64
    }
    return 0;
}
template <int W>
__global__ void __launch_bounds__(512) probe(int p, unsigned long long* out, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 40960; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = i;
    __syncthreads();
    const char* base = smem + wave * 16384 + pattern_addr(p, lane);
    uint32_t acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 32; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (W == 16) { const u32x4 v = *reinterpret_cast<const u32x4*>(base + ((j & 1) * 64 + (j >> 1) * 4096)); acc += v.x ^ v.y ^ v.z ^ v.w; }
            else { const u32x2 v = *reinterpret_cast<const u32x2*>(base + ((j & 1) * 64 + (j >> 1) * 4096 + 8)); acc += v.x ^ v.y; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[wave] = t1 - t0;
    sink[threadIdx.x] = acc;
}
int main() {
    unsigned long long* out; uint32_t* sink; CK(hipMalloc(&out, 64)); CK(hipMalloc(&sink, 4096));
    CK(hipFuncSetAttribute((const void*)probe<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)probe<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const char* names[9] = {"linear lane * 16", "ring A: 128-B rows, chunk ^ (row >> 1 & 7)", "128-B rows, no swizzle", "128-B rows, chunk ^ (row & 7)", "144-B rows (padded)",
                            "ring B records, swizzled", "B records, no swizzle", "staged A rows (528-B pitch)", "staged B columns (136-B pitch)"};
    for (int w16 = 1; w16 >= 0; w16--) for (int nw = 1; nw <= 8; nw *= 8) for (int p = 0; p < 9; p++) {
        if (w16) hipLaunchKernelGGL(probe<16>, dim3(1), dim3(64 * nw), 160 * 1024, 0, p, out, sink); else hipLaunchKernelGGL(probe<8>, dim3(1), dim3(64 * nw), 160 * 1024, 0, p, out, sink);
        CK(hipDeviceSynchronize());
        unsigned long long h[8]; CK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
        double mx = 0; for (int i = 0; i < nw; i++) mx = h[i] > mx ? h[i] : mx;
        printf("%s, %d wave(s), %-45s: %.1f cycles per wave-read (%.0f B/clk per CU)\n", w16 ? "ds_read_b128" : "ds_read_b64 ", nw, names[p], mx / 256.0, nw * 64.0 * (w16 ? 16 : 8) / (mx / 256.0));
    }
    return 0;
}
