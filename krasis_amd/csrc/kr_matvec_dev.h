// kr_matvec_dev.h -- device pieces of the streaming dequant-matvec shared by kr_moe_decode.hip (reference order, bit-exact) and
// kr_decode_fast.hip (tolerance mode): exact integer partial of one 128-wide quantization group against the INT16 activation image in LDS.
#pragma once
#include "kr_device.h"
#include "kr_kernels.h"

__device__ __forceinline__ void kr_dot_word_i4(uint32_t w, const u32x4 r, int& accH, uint32_t& accL) {
    const uint32_t lo = w & 0x0F0F0F0Fu;         // k = 0,2,4,6
    const uint32_t hi = (w >> 4) & 0x0F0F0F0Fu;  // k = 1,3,5,7
    accH = __builtin_amdgcn_sdot4((int)lo, (int)r.x, accH, false);
    accH = __builtin_amdgcn_sdot4((int)hi, (int)r.y, accH, false);
    accL = __builtin_amdgcn_udot4(lo, r.z, accL, false);
    accL = __builtin_amdgcn_udot4(hi, r.w, accL, false);
}

// one quantization group, INT4: lane's two packed words -> exact i32 partial of sum((q-8)*a)
__device__ __forceinline__ int kr_group_i4(uint32_t w0, uint32_t w1, int g, int l8, const KrActLds& L) {
    const int chunk = g * 16 + 2 * l8;
    const u32x4 r0 = L.planes[chunk], r1 = L.planes[chunk + 1];
    int accH = 0; uint32_t accL = 0;
    kr_dot_word_i4(w0, r0, accH, accL);
    kr_dot_word_i4(w1, r1, accH, accL);
    return (accH << 8) + (int)accL - 8 * L.asum16[g * 8 + l8];
}

// one quantization group, INT8: lane's 16 weights (natural k order)
__device__ __forceinline__ int kr_group_i8(const u32x4 w, int g, int l8, const KrActLds& L) {
    const u32x4 ah = L.planes8[(g * 8 + l8) * 2], al = L.planes8[(g * 8 + l8) * 2 + 1];
    int accH = 0, accL = 0, accW = 0;
    accH = __builtin_amdgcn_sdot4((int)w.x, (int)ah.x, accH, false);
    accH = __builtin_amdgcn_sdot4((int)w.y, (int)ah.y, accH, false);
    accH = __builtin_amdgcn_sdot4((int)w.z, (int)ah.z, accH, false);
    accH = __builtin_amdgcn_sdot4((int)w.w, (int)ah.w, accH, false);
    accL = __builtin_amdgcn_sdot4((int)w.x, (int)al.x, accL, false);
    accL = __builtin_amdgcn_sdot4((int)w.y, (int)al.y, accL, false);
    accL = __builtin_amdgcn_sdot4((int)w.z, (int)al.z, accL, false);
    accL = __builtin_amdgcn_sdot4((int)w.w, (int)al.w, accL, false);
    accW = __builtin_amdgcn_sdot4((int)w.x, 0x01010101, accW, false);
    accW = __builtin_amdgcn_sdot4((int)w.y, 0x01010101, accW, false);
    accW = __builtin_amdgcn_sdot4((int)w.z, 0x01010101, accW, false);
    accW = __builtin_amdgcn_sdot4((int)w.w, 0x01010101, accW, false);
    return (accH << 8) + accL + (accW << 7);  // AL = AL' + 128
}
