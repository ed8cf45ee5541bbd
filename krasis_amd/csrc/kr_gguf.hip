// kr_gguf.hip -- native GGUF block experts on gfx950: Q4_K / Q8_0 / Q4_0 with INT16 activations (exact integer dots),
// Q5_0 / Q6_K through the reference's scalar f32 form.  Replaces src/gguf_kernels.rs (matvec_q4_k_avx2 :271, matvec_q8_0_avx2 :379,
// matvec_q4_0_avx2 :436, scalar fallbacks :495-635, quantize_*_to_int16 :110-172, expert_forward_gguf :690) and moe_forward_gguf
// (src/moe.rs:990) bit for bit: the 8 AVX lanes of a row are 8 GPU lanes that see exactly the bytes their AVX twin sees
// (bytes {2l, 2l+1, 16+2l, 16+2l+1} of every 32-byte chunk), run the same per-sub-block fma chain and the same hsum tree.
//
// HBM layout (built at upload from the raw row-major GGUF blocks, kr_engine.cpp::retile_gguf):
//   Q4_K  q[rows/8][blocks][8 rows][8 lanes] x 16 B  : lane record = 4 chunks j x bytes {qs[32j+2l], qs[32j+2l+1], qs[32j+16+2l], qs[32j+17+2l]}
//         h[rows/8][blocks][8 rows] x 16 B           : {f16 d, f16 dmin, 12 scale bytes}
//   Q8_0  q[rows/8][blocks/4][8][8] x 16 B (4 blocks x 4 bytes), h[rows/8][blocks/4][8] x 8 B (4 x f16 d)
//   Q4_0  q[rows/8][blocks/8][8][8] x 16 B (8 blocks x 2 bytes), h[rows/8][blocks/8][8] x 16 B (8 x f16 d)
//   Q5_0 / Q6_K: raw row-major blocks (scalar path, one lane per row)
#include "kr_gguf_dev.h"

// ---- scalar f32 rows (gguf_kernels.rs:572-635): one lane per row, raw row-major blocks ----
__device__ float gg_row_scalar(int type, const uint8_t* row, const float* x, int K) {
    float sum = 0.0f;
    if (type == GG_Q5_0) {
        for (int b = 0; b < K / 32; b++) {
            const uint8_t* blk = row + (size_t)b * 22;
            const float d = gg_f16((uint32_t)blk[0] | ((uint32_t)blk[1] << 8));
            const uint32_t qh = (uint32_t)blk[2] | ((uint32_t)blk[3] << 8) | ((uint32_t)blk[4] << 16) | ((uint32_t)blk[5] << 24);
            const uint8_t* qs = blk + 6;
            for (int j = 0; j < 32; j++) {
                const uint32_t q4 = j < 16 ? (qs[j] & 0x0Fu) : ((qs[j - 16] >> 4) & 0x0Fu);
                const int qv = (int)(q4 | (((qh >> j) & 1u) << 4)) - 16;
                sum += d * (float)qv * x[b * 32 + j];
            }
        }
    } else if (type == GG_Q6_K) {
        for (int b = 0; b < K / 256; b++) {
            const uint8_t* blk = row + (size_t)b * 210; const uint8_t* ql = blk; const uint8_t* qh = blk + 128; const uint8_t* sc = blk + 192;
            const float d = gg_f16((uint32_t)blk[208] | ((uint32_t)blk[209] << 8)); const float* in = x + b * 256;
            for (int hf = 0; hf < 2; hf++) {
                const int qlo = hf * 64, qho = hf * 32, sco = hf * 8, io = hf * 128;
                for (int l = 0; l < 32; l++) {
                    const int is = l / 16;
                    const int q0 = (int)(ql[qlo + l] & 0xF) | ((int)((qh[qho + l] >> 0) & 3) << 4);
                    const int q1 = (int)(ql[qlo + 32 + l] & 0xF) | ((int)((qh[qho + l] >> 2) & 3) << 4);
                    const int q2 = (int)((ql[qlo + l] >> 4) & 0xF) | ((int)((qh[qho + l] >> 4) & 3) << 4);
                    const int q3 = (int)((ql[qlo + 32 + l] >> 4) & 0xF) | ((int)((qh[qho + l] >> 6) & 3) << 4);
                    const float s0 = d * (float)(int8_t)sc[sco + is + 0], s1 = d * (float)(int8_t)sc[sco + is + 2];
                    const float s2 = d * (float)(int8_t)sc[sco + is + 4], s3 = d * (float)(int8_t)sc[sco + is + 6];
                    sum += s0 * (float)(q0 - 32) * in[io + l];
                    sum += s1 * (float)(q1 - 32) * in[io + 32 + l];
                    sum += s2 * (float)(q2 - 32) * in[io + 64 + l];
                    sum += s3 * (float)(q3 - 32) * in[io + 96 + l];
                }
            }
        }
    }
    return sum;
}

__device__ __forceinline__ void gg_run_rows(const GgMat& m, const GgAct& A, float* out, int row_base_out, int tile0, int ntiles_wg) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (gg_int_path(m.type)) {
        const int ntiles = (m.N + 7) / 8;
        for (int t = wave; t < ntiles_wg; t += GG_BLOCK / 64) {
            const int tile = tile0 + t;
            if (tile >= ntiles) break;
            float r;
            if (m.type == GG_Q4_K) r = gg_tile_q4k(m, tile, A, lane);
            else if (m.type == GG_Q8_0) r = gg_tile_q8_0(m, tile, A, lane);
            else r = gg_tile_q4_0(m, tile, A, lane);
            const int row = tile * 8 + (lane >> 3);
            if ((lane & 7) == 0 && row < m.N) out[row_base_out + row] = r;
        }
    } else {
        const size_t row_bytes = (size_t)(m.K / (m.type == GG_Q6_K ? 256 : 32)) * (m.type == GG_Q6_K ? 210 : 22);
        for (int r = tile0 * 8 + threadIdx.x; r < m.N && r < (tile0 + ntiles_wg) * 8; r += GG_BLOCK)
            out[row_base_out + r] = gg_row_scalar(m.type, reinterpret_cast<const uint8_t*>(m.q) + (size_t)r * row_bytes, A.f32v, m.K);
    }
}

extern __shared__ __attribute__((aligned(16))) char gg_smem[];

// stage 1: gu[b][slot] = [gate rows | up rows]      grid (row-tile groups of gate+up, n_slots, B)
__global__ void __launch_bounds__(GG_BLOCK) kr_gguf_w13_kernel(const GgMoeArgs a, int tiles_per_wg) {
    const int slot = blockIdx.y, b = blockIdx.z;
    const bool shared = slot >= a.topk;
    const int e = shared ? 0 : a.ids[(size_t)b * a.topk + slot];
    if (e < 0 || (!shared && e >= a.E)) return;
    const GgMat gate = shared ? a.sgate : gg_expert_mat(a.gate, e), up = shared ? a.sup : gg_expert_mat(a.up, e);
    const int I = gate.N, nt = (I + 7) / 8;
    const int t0 = blockIdx.x * tiles_per_wg;
    if (t0 >= 2 * nt) return;
    const bool intp = gg_int_path(gate.type) && (a.H % 32 == 0);
    const GgAct A = gg_carve(gg_smem, a.H);
    // tiles [0, nt) are gate rows, [nt, 2nt) are up rows; a workgroup's span may straddle the boundary
    const int t1 = t0 + tiles_per_wg < 2 * nt ? t0 + tiles_per_wg : 2 * nt;
    const int ng = t0 < nt ? (t1 < nt ? t1 : nt) - t0 : 0;           // gate tiles of this workgroup, then its up tiles [us, us + nu)
    const int us = t0 > nt ? t0 - nt : 0, nu = t1 > nt ? (t1 - nt) - us : 0;
    if (a.act_f32) gg_prologue_f32_as_bf16(a.act_f32 + (size_t)b * a.H, a.H, A, !intp);
    else gg_prologue_bf16(a.act + (size_t)b * a.H, a.H, A, !intp);
    __syncthreads();
    float* gu = a.gu + ((size_t)b * a.n_slots + slot) * a.gu_ld;
    if (ng > 0) gg_run_rows(gate, A, gu, 0, t0, ng);
    if (nu > 0) gg_run_rows(up, A, gu, a.gu_ld / 2, us, nu);
}

// stage 2: eo[b][slot] = down . q(silu(gate)*up)
__global__ void __launch_bounds__(GG_BLOCK) kr_gguf_w2_kernel(const GgMoeArgs a, int tiles_per_wg) {
    const int slot = blockIdx.y, b = blockIdx.z;
    const bool shared = slot >= a.topk;
    const int e = shared ? 0 : a.ids[(size_t)b * a.topk + slot];
    if (e < 0 || (!shared && e >= a.E)) return;
    const GgMat down = shared ? a.sdown : gg_expert_mat(a.down, e);
    const int I = down.K, nt = (down.N + 7) / 8;
    const int t0 = blockIdx.x * tiles_per_wg;
    if (t0 >= nt) return;
    const bool intp = gg_int_path(down.type) && (I % 32 == 0);
    const GgAct A = gg_carve(gg_smem, I);
    const float* gu = a.gu + ((size_t)b * a.n_slots + slot) * a.gu_ld;
    // gu holds gate at [0,I) and up at [gu_ld/2, gu_ld/2+I): present them contiguously to the prologue
    gg_prologue_hidden_split(gu, gu + a.gu_ld / 2, I, A, !intp, intp);
    __syncthreads();
    gg_run_rows(down, A, a.eo + ((size_t)b * a.n_slots + slot) * a.H, 0, t0, tiles_per_wg);
}

void kr_launch_gguf_moe(const GgMoeArgs& a, hipStream_t st) {
    {
        const int nt2 = 2 * ((a.I_max + 7) / 8);
        int tpw = 8; while (tpw > 4 && (nt2 + tpw - 1) / tpw < 64) tpw >>= 1;
        dim3 grid((nt2 + tpw - 1) / tpw, a.n_slots, a.B);
        hipLaunchKernelGGL(kr_gguf_w13_kernel, grid, dim3(GG_BLOCK), gg_lds_bytes(a.H, true), st, a, tpw);
    }
    {
        const int nt = (a.H + 7) / 8;
        int tpw = 8; while (tpw > 4 && (nt + tpw - 1) / tpw < 64) tpw >>= 1;
        dim3 grid((nt + tpw - 1) / tpw, a.n_slots, a.B);
        hipLaunchKernelGGL(kr_gguf_w2_kernel, grid, dim3(GG_BLOCK), gg_lds_bytes(a.I_max, true), st, a, tpw);
    }
}

size_t gg_q_bytes(int type, int K, int N) {
    const size_t nt = (size_t)(N + 7) / 8;
    switch (type) {
        case GG_Q4_K: return nt * (K / 256) * 64 * 16;
        case GG_Q8_0: return nt * (((K / 32) + 3) / 4) * 64 * 16;
        case GG_Q4_0: return nt * (((K / 32) + 7) / 8) * 64 * 16;
        case GG_Q5_0: return (size_t)N * (K / 32) * 22;
        case GG_Q6_K: return (size_t)N * (K / 256) * 210;
        default: return 0;
    }
}
size_t gg_h_bytes(int type, int K, int N) {
    const size_t nt = (size_t)(N + 7) / 8;
    switch (type) {
        case GG_Q4_K: return nt * (K / 256) * 8 * 16;
        case GG_Q8_0: return nt * (((K / 32) + 3) / 4) * 8 * 8;
        case GG_Q4_0: return nt * (((K / 32) + 7) / 8) * 8 * 16;
        default: return 16;
    }
}
