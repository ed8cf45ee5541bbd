// kr_pfh_dev.h -- device code shared by the tolerance GEMM kernels (kr_prefill_h.hip: register-staged 64 x 128|256 tiles; kr_prefill_ring.hip: the
// LDS-ring form fed by LDS-DMA): the launch arguments, the INT4 / INT8 -> f16 de-quantization, the activation of the fused gate | up epilogue and the
// accumulator stores.  The A operand of both kernels is the f16 row image the row kernels of kr_prefill_h.hip write: values scaled by a power of two,
// and -- since round 6 -- the 8 values of every aligned group of 8 stored in the order (0,4,1,5,2,6,3,7), the order pfh_dq4 emits the 8 nibbles of a
// packed word in: the INT4 kernels copy rows to LDS verbatim (LDS-DMA cannot permute), the INT8 kernel un-permutes in its commit pass.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kr_device.h"
#include "kr_libm.h"
#include "kr_kernels.h"

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------
// A operand: rows -> f16 with a power-of-two row multiplier (shared by the row kernels of kr_prefill_h.hip and the norm kernel of kr_prefill_ops.hip, which writes the
// image of its own output in the tolerance pass)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void pfh_row_scale(float mx, float& scl, float& inv) {    // mx >= 0: scl = 2^-e, inv = 2^e, e = exponent of mx
    uint32_t E = __float_as_uint(mx) >> 23;
    if (E == 0 || E > 253) { scl = 1.0f; inv = 1.0f; return; }       // zero row (or not finite): unscaled
    scl = __uint_as_float((254u - E) << 23); inv = __uint_as_float(E << 23);
}
__device__ __forceinline__ uint32_t pfh_pack_h2(float a, float b) {
    const v2h h = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float pfh_block_max(float mx, uint32_t* slot) {   // slot: LDS word zeroed before the call + barrier
    mx = kr_red16_max_f32(mx);
    if ((threadIdx.x & 15) == 0) atomicMax(slot, __float_as_uint(mx));        // non-negative floats order like their bit patterns
    __syncthreads();
    return __uint_as_float(*slot);
}


struct KrPfGemmHArgs {
    KrMatDev m;
    const uint16_t* a; const float* a_mul;   // f16 rows [rows_or_tokens][K], row multipliers
    const uint16_t* a_sum;                   // G = 1 (Q4_K copy): f16 sums of every 32 consecutive values of a row, [rows_or_tokens][K / 32]
    const int* row_pair; int topk; int gather_tokens;
    const int* tile_expert; const int* tile_row0; const int* tile_rows; const int* n_tiles;
    float* out; int out_ld;
    int single_expert; int total_rows; int scatter_rows; int out_bf16;      // out_bf16: element type of out 0 f32, 1 bf16, 2 f16
    int act_fused; float act_limit, act_alpha;      // gate | up GEMM of 256-column tiles: h = act(gate, up) formed in the epilogue, out = [rows][N / 2] f32
    int run;                                 // experts: consecutive row tiles given to one XCD (> 1: the tiles of one expert share an L2)
    int sr, sc;                              // dense: super-tile of sr row tiles x sc column blocks per XCD (kr_pf_super_tile)
    int n_extra; KrMatDev mx[2]; float* outx[2]; int out_ldx[2];
};

// one packed INT4 word (nibbles k .. k + 7 of a column) -> 8 f16 in the order (0,4,1,5,2,6,3,7); sq = s / 4 (both halves), cq = -1536 * sq.
// 12 VALU: and + (shift | or) for the two low pairs, shift + and-or for the two high pairs, 4 packed fma.  M0 / M1 / M2 / Kc are held in registers by the
// caller (opaque to the compiler): VOP3 takes no literal on gfx9, with literals the compiler falls back to separate v_and / v_or (16 VALU).
__device__ __forceinline__ v8h pfh_dq4(uint32_t w, v2h sq, v2h cq, uint32_t M0, uint32_t M1, uint32_t MH, uint32_t Kc) {
    const uint32_t t0 = ((w & M0) << 6) | Kc;                  // v_and, v_lshl_or       (n0, n4) at mantissa bits 6..9
    const uint32_t t1 = ((w & M1) << 2) | Kc;                  // v_and, v_lshl_or       (n1, n5)
    const uint32_t t2 = ((w >> 2) & MH) | Kc;                  // v_lshrrev, v_and_or    (n2, n6)
    const uint32_t t3 = ((w >> 6) & MH) | Kc;                  // v_lshrrev, v_and_or    (n3, n7)
    const v2h r0 = __builtin_elementwise_fma(__builtin_bit_cast(v2h, t0), sq, cq), r1 = __builtin_elementwise_fma(__builtin_bit_cast(v2h, t1), sq, cq);
    const v2h r2 = __builtin_elementwise_fma(__builtin_bit_cast(v2h, t2), sq, cq), r3 = __builtin_elementwise_fma(__builtin_bit_cast(v2h, t3), sq, cq);
    return v8h{r0.x, r0.y, r1.x, r1.y, r2.x, r2.y, r3.x, r3.y};
}
// 8 INT8 weights (two words, natural k order) -> 8 f16: (1024 + (b ^ 0x80)) - 1152 = b, times sb = s * 16
__device__ __forceinline__ v8h pfh_dq8(uint32_t w0, uint32_t w1, v2h sb) {
    const uint32_t x0 = w0 ^ 0x80808080u, x1 = w1 ^ 0x80808080u, Kc = 0x64646464u;
    const v2h off = {(_Float16)-1152.0f, (_Float16)-1152.0f};
    // v_perm: selector bytes 0-3 pick from the second operand, 4-7 from the first
    const uint32_t t0 = __builtin_amdgcn_perm(Kc, x0, 0x04010400u), t1 = __builtin_amdgcn_perm(Kc, x0, 0x04030402u);
    const uint32_t t2 = __builtin_amdgcn_perm(Kc, x1, 0x04010400u), t3 = __builtin_amdgcn_perm(Kc, x1, 0x04030402u);
    const v2h r0 = (__builtin_bit_cast(v2h, t0) + off) * sb, r1 = (__builtin_bit_cast(v2h, t1) + off) * sb;
    const v2h r2 = (__builtin_bit_cast(v2h, t2) + off) * sb, r3 = (__builtin_bit_cast(v2h, t3) + off) * sb;
    return v8h{r0.x, r0.y, r1.x, r1.y, r2.x, r2.y, r3.x, r3.y};
}

// Stores of a 32 x 32 accumulator block set: lane (n31, khalf) holds column n31 of 16 rows (r & 3) + 8 (r >> 2) + 4 khalf.  FULL: every row and column
// of the tile is valid and rows are stored in GEMM order (no scatter): the row base is wave-uniform (scalar), the lane part one 32-bit offset, no
// exec-mask branches -- the guarded form costs ~20 instructions and two branches per store.  BF16: round to bf16 (RNE) on the way out.
// OT: element type of the output 0 f32, 1 bf16 (RNE), 2 f16 (the expert rows of the tolerance mode: half the bytes for the store and for the combine pass;
// RAW accumulators -- the kernel parks multipliers of 1 -- so the values are O(1..100) for any row magnitude; kr_pf_combine_kernel<2> applies the row multiplier).
// ACTF != 0 (NC == 2, gate | up GEMM): column block 0 holds gate columns, block 1 the matching up columns -- the epilogue forms h = act(g, u) and stores ONE
// value per pair at the hidden column (f32): the [rows, 2 I] gate | up matrix is never written or re-read.  1 = silu with the poly-5 sigmoid
// (avx2.rs:2331), 2 = GPT-OSS (moe.rs:268-287), 3 = libm silu (gguf_kernels.rs:733-737).
__device__ __forceinline__ float pfh_act(int ACTF, float g, float u, float limit, float alpha) {      // ACTF is wave-uniform
    if (ACTF == 2) {
        float gate = g, up = u;
        if (gate > limit) gate = limit;
        if (up > limit) up = limit;
        if (up < -limit) up = -limit;
        return (up + 1.0f) * (gate * kr_sigmoid_poly5_scalar(gate * alpha));
    }
    if (ACTF == 3) return (g / (1.0f + kr_expf(-g))) * u;
    return (g * kr_sigmoid_poly5(g)) * u;
}
template <int OT> __device__ __forceinline__ void pfh_put(void* base, size_t idx, float v) {
    if (OT == 1) reinterpret_cast<uint16_t*>(base)[idx] = kr_f32_to_bf16(v);
    else if (OT == 2) { const _Float16 h = (_Float16)__builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);     // saturates instead of overflowing to inf
         reinterpret_cast<uint16_t*>(base)[idx] = __builtin_bit_cast(uint16_t, h); }
    else reinterpret_cast<float*>(base)[idx] = v;
}
template <int NSB, int NC, bool FULL, int OT, bool ACT, typename ACC>
__device__ __forceinline__ void pfh_store_tile(const ACC& acc, int nsb, int rows, int row0, const float* rmul, const int* row_dst, void* out_p, int out_ld,
                                               const int (&col)[NC], int N, int lane, int ACTF = 0, float limit = 0.0f, float alpha = 0.0f) {
    const int khalf = lane >> 5;
    constexpr int NCS = ACT ? 1 : NC;       // stores per (row, lane)
    constexpr size_t ESZ = OT == 0 ? 4 : 2;
#pragma unroll
    for (int s = 0; s < NSB; s++) {
        if (s >= nsb) break;
#pragma unroll
        for (int rq = 0; rq < 4; rq++) {
            const int rowb = s * 32 + 8 * rq + 4 * khalf;
            const float4 rm4 = *reinterpret_cast<const float4*>(rmul + rowb);
            const float rm[4] = {rm4.x, rm4.y, rm4.z, rm4.w};
            if (FULL) {
                const uint32_t lane_off = (uint32_t)(4 * khalf) * (uint32_t)out_ld;      // elements; + col below
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    char* rb = reinterpret_cast<char*>(out_p) + (size_t)(row0 + s * 32 + 8 * rq + i) * out_ld * ESZ;     // wave-uniform
#pragma unroll
                    for (int c = 0; c < NCS; c++) {
                        float v = acc[s][c][rq * 4 + i] * rm[i];
                        if (ACT) v = pfh_act(ACTF, v, acc[s][NC - 1][rq * 4 + i] * rm[i], limit, alpha);
                        pfh_put<OT>(rb, (size_t)(lane_off + (uint32_t)col[c]), v);
                    }
                }
            } else {
                const int4 rd4 = *reinterpret_cast<const int4*>(row_dst + rowb);
                const int rd[4] = {rd4.x, rd4.y, rd4.z, rd4.w};
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (rowb + i < rows) {
#pragma unroll
                        for (int c = 0; c < NCS; c++)
                            if (col[c] < N) {
                                float v = acc[s][c][rq * 4 + i] * rm[i];
                                if (ACT) v = pfh_act(ACTF, v, acc[s][NC - 1][rq * 4 + i] * rm[i], limit, alpha);
                                pfh_put<OT>(out_p, (size_t)rd[i] * out_ld + col[c], v);
                            }
                    }
            }
        }
    }
}

// kr_prefill_ring.hip: the LDS-ring form of the INT4 GEMM; 0 = launched, 1 = this shape stays on the register-staged kernel.  mt64 = 64-row tiles of the experts' tile table
int kr_pfr_try_launch(const KrPfGemmHArgs& a, int mt64, hipStream_t st);
