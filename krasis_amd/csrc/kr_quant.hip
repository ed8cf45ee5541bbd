// kr_quant.hip -- weight quantizers on the GPU, writing the resident lane-tiled layout directly.
//
// Replaces (bit-exactly) the reference's load-time quantizers for BF16 checkpoints:
//   quantize_int4 (src/weights/marlin.rs:145-207): per (row, 128-group) amax -> scale = bf16_rne(amax / 7) (1.0 if amax == 0),
//     q = clamp(round_half_away(v * (1 / bf16(scale))), -8, 7), stored as q + 8
//   quantize_int8 (src/weights/marlin.rs:65-114): scale = bf16_rne(amax / 127), q = clamp(round(v / ...), -128, 127)
// followed by the transposed [K/8, N] packing (src/weights/mod.rs:329-470) -- here the nibbles are written straight into the
// lane-tiled HBM layout of DESIGN.md §3.1, so a BF16 expert goes HtoD once and is never re-packed on the host.
//
// Source layout = the HF checkpoint's: W[n][k] bf16 row-major (n = output row, k = input), i.e. column n of the [K -> N] matrix.
#include "kr_device.h"
#include "kr_kernels.h"

__device__ __forceinline__ float kr_q_red8_max(float v) {
    v = fmaxf(v, __int_as_float(KR_DPP(__float_as_int(v), KR_DPP_XOR1)));
    v = fmaxf(v, __int_as_float(KR_DPP(__float_as_int(v), KR_DPP_XOR2)));
    v = fmaxf(v, __int_as_float(KR_DPP(__float_as_int(v), KR_DPP_HALF_MIRROR)));
    return v;
}

// grid (ceil(rows/8) tiles, ngp), 64 threads: lane = col*8 + l handles k = g*128 + 16l .. +16 of groups g0 = 2gp, g1 = 2gp+1 of row n = tile*8 + col
// dst tile index = tile0 + tile (w13: gate at tile0 = 0, up at tile0 = I/8)
template <int BITS>
__global__ void __launch_bounds__(64) kr_quant_bf16_kernel(const uint16_t* __restrict__ w, int rows, int K, void* __restrict__ qdst, uint32_t* __restrict__ sdst,
                                                          int tile0, int ng, int ngp) {
    const int tile = blockIdx.x, gp = blockIdx.y, lane = threadIdx.x, col = lane >> 3, l = lane & 7, n = tile * 8 + col;
    float v[2][16]; float amax[2] = {0.0f, 0.0f};
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int g = 2 * gp + h;
        const bool ok = n < rows && g < ng;
        u32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
        if (ok) { const u32x4* p = reinterpret_cast<const u32x4*>(w + (size_t)n * K + (size_t)g * 128 + 16 * l); a = p[0]; b = p[1]; }
        const uint32_t ww[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 8; i++) { v[h][2 * i] = __uint_as_float(ww[i] << 16); v[h][2 * i + 1] = __uint_as_float(ww[i] & 0xFFFF0000u); }
#pragma unroll
        for (int i = 0; i < 16; i++) amax[h] = fmaxf(amax[h], fabsf(v[h][i]));
        amax[h] = kr_q_red8_max(amax[h]);
    }
    const float qmax = BITS == 4 ? 7.0f : 127.0f;
    uint16_t sb[2]; float inv[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const float scale = amax[h] == 0.0f ? 1.0f : amax[h] / qmax;
        sb[h] = kr_f32_to_bf16(scale);
        const float sc = kr_bf16_to_f32(sb[h]);
        inv[h] = sc == 0.0f ? 0.0f : 1.0f / sc;
    }
    const size_t dtile = (size_t)(tile0 + tile);
    if (BITS == 4) {
        uint32_t words[4];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const bool pad = 2 * gp + h >= ng || n >= rows;      // padding = nibble 8 (weight 0), DESIGN.md §3.1
#pragma unroll
            for (int i = 0; i < 2; i++) {
                uint32_t word = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float q = roundf(v[h][8 * i + j] * inv[h]);
                    q = q < -8.0f ? -8.0f : (q > 7.0f ? 7.0f : q);
                    const uint32_t u4 = pad ? 8u : ((uint32_t)((int)q + 8) & 0xFu);
                    word |= u4 << (4 * j);
                }
                words[h * 2 + i] = word;
            }
        }
        u32x4 rec = {words[0], words[1], words[2], words[3]};
        reinterpret_cast<u32x4*>(qdst)[(dtile * ngp + gp) * 64 + lane] = rec;
    } else {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int g = 2 * gp + h;
            if (g >= ng) continue;
            uint32_t words[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t word = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float q = roundf(v[h][4 * i + j] * inv[h]);
                    q = q < -128.0f ? -128.0f : (q > 127.0f ? 127.0f : q);
                    word |= ((uint32_t)(int)q & 0xFFu) << (8 * j);
                }
                words[i] = n < rows ? word : 0u;
            }
            u32x4 rec = {words[0], words[1], words[2], words[3]};
            reinterpret_cast<u32x4*>(qdst)[(dtile * ng + g) * 64 + lane] = rec;
        }
    }
    if (l == 0) sdst[(dtile * ngp + gp) * 8 + col] = (n < rows ? (uint32_t)sb[0] : 0u) | ((n < rows && 2 * gp + 1 < ng ? (uint32_t)sb[1] : 0u) << 16);
}

void kr_launch_quant_bf16(const uint16_t* w_dev, int rows, int K, int bits, void* qdst, uint32_t* sdst, int tile0, hipStream_t st) {
    const int ng = K / 128, ngp = (ng + 1) / 2;
    dim3 grid((rows + 7) / 8, ngp);
    if (bits == 4) hipLaunchKernelGGL(kr_quant_bf16_kernel<4>, grid, dim3(64), 0, st, w_dev, rows, K, qdst, sdst, tile0, ng, ngp);
    else hipLaunchKernelGGL(kr_quant_bf16_kernel<8>, grid, dim3(64), 0, st, w_dev, rows, K, qdst, sdst, tile0, ng, ngp);
}
