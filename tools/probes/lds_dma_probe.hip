// Probe: semantics of the LDS-DMA statements of kr_prefill_ring.hip (global_load_lds_dwordx4 / _dword with an SGPR base, a per-lane 32-bit offset and M0 as the
// LDS destination): where do lane i's bytes land, do LDS addresses above 64 KiB work, is the instruction offset field applied.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ void dma16(uint32_t lds_dst, uint32_t voff, const void* sbase) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}
__device__ __forceinline__ void dma4(uint32_t lds_dst, uint32_t voff, const void* sbase) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 2\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}
__device__ __forceinline__ void dma12(uint32_t lds_dst, uint32_t voff, const void* sbase) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 2\n\tglobal_load_lds_dwordx3 %1, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}
__global__ void __launch_bounds__(64) probe(const uint32_t* src, uint32_t* out, uint32_t lds_base, int mode) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x;
    uint32_t* w = reinterpret_cast<uint32_t*>(smem + lds_base);
    for (int i = lane; i < 512; i += 64) w[i] = 0xDEAD0000u + i;
    __syncthreads();
    const uint32_t l0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem + lds_base;
    if (mode == 0) dma16(l0, (uint32_t)((63 - lane) * 16), src);          // lane i fetches chunk 63 - i
    else if (mode == 1) dma4(l0, (uint32_t)((63 - lane) * 4), src);
    else if (mode == 2) dma12(l0, (uint32_t)((63 - lane) * 12), src);          // 12-byte pieces (kr_prefill_mx.hip): lane i fetches piece 63 - i
    else dma12(l0 + 8, (uint32_t)(lane * 12 + 4), src);                         // destination and source only 4-byte aligned
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = w[i];
}
int main() {
    std::vector<uint32_t> h(1024); for (int i = 0; i < 1024; i++) h[i] = i;
    uint32_t *src, *out; CK(hipMalloc(&src, 4096)); CK(hipMalloc(&out, 2048)); CK(hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const uint32_t bases[3] = {0, 60 * 1024, 150 * 1024};
    for (int mode = 0; mode < 4; mode++) for (int b = 0; b < 3; b++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 160 * 1024, 0, src, out, bases[b], mode); CK(hipDeviceSynchronize());
        std::vector<uint32_t> o(512); CK(hipMemcpy(o.data(), out, 2048, hipMemcpyDeviceToHost));
        const char* mn[4] = {"dwordx4", "dword  ", "dwordx3", "dwordx3 unaligned"};
        printf("mode %s lds base %6u: words 0..11: ", mn[mode], bases[b]); for (int i = 0; i < 12; i++) printf("%x ", o[i]);
        printf("| words 186..195: "); for (int i = 186; i < 196; i++) printf("%x ", o[i]); printf("| words 252..259: "); for (int i = 252; i < 260; i++) printf("%x ", o[i]); printf("\n");
    }
    return 0;
}
