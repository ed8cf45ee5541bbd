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
#include "kr_device.h"
#include "kr_libm.h"
#include "kr_gguf.h"
#include <hip/hip_fp16.h>

#define GG_BLOCK 256

__device__ __forceinline__ float gg_f16(uint32_t bits16) { return __half2float(__ushort_as_half((uint16_t)bits16)); }
__device__ __forceinline__ float gg_hsum8(float v) {
    v = v + __shfl_xor(v, 4); v = v + __shfl_xor(v, 1); v = v + __shfl_xor(v, 2);
    return v;
}

// ---- activation image: per 32-element sub-block s and AVX lane l an 8-byte record {AH4, AL4} of elements {2l,2l+1,16+2l,17+2l} ----
struct GgAct { uint32_t* rec; float* scale; int* sum; float* f32v; };   // rec [K/32][8][2], scale/sum [K/32], f32v [K] (scalar path)
__device__ __forceinline__ GgAct gg_carve(char* smem, int K) {
    GgAct a; a.rec = reinterpret_cast<uint32_t*>(smem); a.scale = reinterpret_cast<float*>(smem + (size_t)(K / 32) * 64);
    a.sum = reinterpret_cast<int*>(a.scale + K / 32); a.f32v = reinterpret_cast<float*>(a.sum + K / 32);
    return a;
}
static inline size_t gg_lds_bytes(int K, bool want_f32) { return (size_t)(K / 32) * 64 + (size_t)(K / 32) * 8 + (want_f32 ? (size_t)K * 4 : 0) + 16; }

// quantize_bf16_to_int16 / quantize_f32_to_int16 (gguf_kernels.rs:110,143): per 32, f32::round, clamp, i32 sums.
// One thread per 8 elements, 4 consecutive lanes per sub-block.
__device__ __forceinline__ void gg_quant_store(const float (&v)[8], int c, const GgAct& A) {
    float mx = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) mx = fmaxf(mx, fabsf(v[i]));
    mx = kr_red4_max_f32(mx);
    const float scale = mx > 0.0f ? mx / 32767.0f : 1.0f, inv = mx > 0.0f ? 32767.0f / mx : 0.0f;
    int q[8]; int s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int t = (int)roundf(v[i] * inv); t = t > 32767 ? 32767 : (t < -32768 ? -32768 : t);
        q[i] = t; s += t;
    }
    s += KR_DPP(s, KR_DPP_XOR1); s += KR_DPP(s, KR_DPP_XOR2);
    const int sb = c >> 2, part = c & 3;            // chunk part: 0,1 -> first pair halves of lanes 0-3 / 4-7 ; 2,3 -> second pair halves
    uint16_t* rec16 = reinterpret_cast<uint16_t*>(A.rec + (size_t)sb * 16);
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int l = (part & 1) * 4 + p;           // AVX lane served by this pair
        const int half = part >> 1;                 // 0: elements (2l,2l+1), 1: elements (16+2l,17+2l)
        const int a0 = q[2 * p], a1 = q[2 * p + 1];
        rec16[l * 4 + half] = (uint16_t)(((a0 >> 8) & 0xFF) | (((a1 >> 8) & 0xFF) << 8));       // AH bytes
        rec16[l * 4 + 2 + half] = (uint16_t)((a0 & 0xFF) | ((a1 & 0xFF) << 8));                  // AL bytes
    }
    if (part == 0) { A.scale[sb] = scale; A.sum[sb] = s; }
}

__device__ __forceinline__ void gg_prologue_bf16(const uint16_t* x, int K, const GgAct& A, bool keep_f32) {
    for (int c = threadIdx.x; c < K / 8; c += GG_BLOCK) {
        const u32x4 r = *reinterpret_cast<const u32x4*>(x + (size_t)c * 8);
        float v[8];
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xFFFF0000u); v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xFFFF0000u);
        v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xFFFF0000u); v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xFFFF0000u);
        if (keep_f32) {
#pragma unroll
            for (int i = 0; i < 8; i++) A.f32v[c * 8 + i] = v[i];
        }
        gg_quant_store(v, c, A);
    }
}
// the decode graph's f32 hidden: the routed experts see bf16(hidden) (decode.rs:3307-3309: f32 -> bf16 RNE before moe_forward), then the bf16 path above
__device__ __forceinline__ void gg_prologue_f32_as_bf16(const float* x, int K, const GgAct& A, bool keep_f32) {
    for (int c = threadIdx.x; c < K / 8; c += GG_BLOCK) {
        float v[8];
        kr_load8(x, c, v);
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = kr_bf16_to_f32(kr_f32_to_bf16(v[i]));
        if (keep_f32) {
#pragma unroll
            for (int i = 0; i < 8; i++) A.f32v[c * 8 + i] = v[i];
        }
        gg_quant_store(v, c, A);
    }
}
// hidden = silu(gate) * up with libm exp (gguf_kernels.rs:733-737), then per-32 quantization
__device__ __forceinline__ void gg_prologue_hidden_split(const float* gate, const float* up, int n, const GgAct& A, bool keep_f32, bool do_quant) {
    for (int c = threadIdx.x; c < n / 8; c += GG_BLOCK) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float g = gate[c * 8 + i];
            const float silu = g / (1.0f + kr_expf(-g));
            v[i] = silu * up[c * 8 + i];
            if (keep_f32) A.f32v[c * 8 + i] = v[i];
        }
        if (do_quant) gg_quant_store(v, c, A);
    }
    // tail when n % 8 != 0 (scalar path only)
    if (keep_f32) for (int i = (n / 8) * 8 + threadIdx.x; i < n; i += GG_BLOCK) { const float g = gate[i]; A.f32v[i] = (g / (1.0f + kr_expf(-g))) * up[i]; }
}

__device__ __forceinline__ void gg_scale_min_k4(int j, uint32_t s0, uint32_t s1, uint32_t s2, int& sc, int& mn) {   // gguf_kernels.rs:640
    const uint32_t w[3] = {s0, s1, s2};
    auto B = [&](int i) -> uint32_t { return (w[i >> 2] >> ((i & 3) * 8)) & 0xFFu; };
    if (j < 4) { sc = (int)(B(j) & 63u); mn = (int)(B(j + 4) & 63u); }
    else { sc = (int)((B(j + 4) & 0xFu) | ((B(j - 4) >> 6) << 4)); mn = (int)((B(j + 4) >> 4) | ((B(j) >> 6) << 4)); }
}

// ---- one row tile (8 rows x 8 lanes), int path; returns the row result in every lane of the row's 8-lane group ----
// The block records of a tile are REQUESTED IN BATCHES of GG_PF before the first one is consumed (round 4): the loop used to issue one block's two loads,
// wait for them (one HBM / fabric round trip each), compute, and go on -- K / 256 dependent round trips per tile, the whole duration of the launch.  Indices
// past the last block are clamped (the load is unconditional, its value unused), the arithmetic and its order are unchanged: same bits.
#define GG_PF 8
__device__ __forceinline__ float gg_tile_q4k(const GgMat& m, int tile, const GgAct& A, int lane) {
    const int l = lane & 7, row = lane >> 3;
    const int nb = m.K / 256;
    const u32x4* q = reinterpret_cast<const u32x4*>(m.q) + (size_t)tile * nb * 64 + lane;
    const u32x4* h = reinterpret_cast<const u32x4*>(m.h) + (size_t)tile * nb * 8 + row;
    float acc = 0.0f, corr = 0.0f;
    for (int b0 = 0; b0 < nb; b0 += GG_PF) {
        u32x4 wv[GG_PF], hv[GG_PF];
#pragma unroll
        for (int u = 0; u < GG_PF; u++) { const int bb = b0 + u < nb ? b0 + u : nb - 1; wv[u] = kr_ldg_nt(q + (size_t)bb * 64); hv[u] = kr_ldg_nt(h + (size_t)bb * 8); }
#pragma unroll
        for (int u = 0; u < GG_PF; u++) {
            const int b = b0 + u;
            if (b < nb) {
                const u32x4 w = wv[u], hd = hv[u];
                const float d = gg_f16(hd.x & 0xFFFFu), dmin = gg_f16(hd.x >> 16);
                const uint32_t wj[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    int sc_lo, mn_lo, sc_hi, mn_hi;
                    gg_scale_min_k4(2 * j, hd.y, hd.z, hd.w, sc_lo, mn_lo); gg_scale_min_k4(2 * j + 1, hd.y, hd.z, hd.w, sc_hi, mn_hi);
                    const int s_lo = b * 8 + 2 * j, s_hi = s_lo + 1;
                    const u32x2 r_lo = *reinterpret_cast<const u32x2*>(A.rec + ((size_t)s_lo * 8 + l) * 2);
                    const u32x2 r_hi = *reinterpret_cast<const u32x2*>(A.rec + ((size_t)s_hi * 8 + l) * 2);
                    const uint32_t lo = wj[j] & 0x0F0F0F0Fu, hi = (wj[j] >> 4) & 0x0F0F0F0Fu;
                    const int i_lo = (__builtin_amdgcn_sdot4((int)lo, (int)r_lo.x, 0, false) << 8) + (int)__builtin_amdgcn_udot4(lo, r_lo.y, 0u, false);
                    const int i_hi = (__builtin_amdgcn_sdot4((int)hi, (int)r_hi.x, 0, false) << 8) + (int)__builtin_amdgcn_udot4(hi, r_hi.y, 0u, false);
                    const float as_lo = A.scale[s_lo], as_hi = A.scale[s_hi];
                    acc = __builtin_fmaf((float)i_lo, d * (float)sc_lo * as_lo, acc);
                    corr += dmin * (float)mn_lo * as_lo * (float)A.sum[s_lo];
                    acc = __builtin_fmaf((float)i_hi, d * (float)sc_hi * as_hi, acc);
                    corr += dmin * (float)mn_hi * as_hi * (float)A.sum[s_hi];
                }
            }
        }
    }
    return gg_hsum8(acc) - corr;
}

__device__ __forceinline__ float gg_tile_q8_0(const GgMat& m, int tile, const GgAct& A, int lane) {
    const int l = lane & 7, row = lane >> 3;
    const int nb = m.K / 32, nbg = (nb + 3) / 4;
    const u32x4* q = reinterpret_cast<const u32x4*>(m.q) + (size_t)tile * nbg * 64 + lane;
    const u32x2* h = reinterpret_cast<const u32x2*>(m.h) + (size_t)tile * nbg * 8 + row;
    float acc = 0.0f;
    for (int g0 = 0; g0 < nbg; g0 += GG_PF) {      // batched requests, as gg_tile_q4k
        u32x4 wv[GG_PF]; u32x2 hv[GG_PF];
#pragma unroll
        for (int v = 0; v < GG_PF; v++) { const int gg = g0 + v < nbg ? g0 + v : nbg - 1; wv[v] = kr_ldg_nt(q + (size_t)gg * 64); hv[v] = h[(size_t)gg * 8]; }
#pragma unroll
        for (int v = 0; v < GG_PF; v++) {
            const int bg = g0 + v;
            if (bg < nbg) {
                const u32x4 w = wv[v]; const u32x2 hd = hv[v];
                const uint32_t wb[4] = {w.x, w.y, w.z, w.w};
                const float dd[4] = {gg_f16(hd.x & 0xFFFFu), gg_f16(hd.x >> 16), gg_f16(hd.y & 0xFFFFu), gg_f16(hd.y >> 16)};
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int s = bg * 4 + u;
                    if (s < nb) {
                        const u32x2 r = *reinterpret_cast<const u32x2*>(A.rec + ((size_t)s * 8 + l) * 2);
                        const int iv = (__builtin_amdgcn_sdot4((int)wb[u], (int)r.x, 0, false) << 8) + __builtin_amdgcn_sdot4((int)wb[u], (int)(r.y ^ 0x80808080u), 0, false) +
                                       (__builtin_amdgcn_sdot4((int)wb[u], 0x01010101, 0, false) << 7);
                        acc = __builtin_fmaf((float)iv, dd[u] * A.scale[s], acc);
                    }
                }
            }
        }
    }
    return gg_hsum8(acc);
}

__device__ __forceinline__ float gg_tile_q4_0(const GgMat& m, int tile, const GgAct& A, int lane) {
    const int l = lane & 7, row = lane >> 3;
    const int nb = m.K / 32, nbg = (nb + 7) / 8;
    const u32x4* q = reinterpret_cast<const u32x4*>(m.q) + (size_t)tile * nbg * 64 + lane;
    const u32x4* h = reinterpret_cast<const u32x4*>(m.h) + (size_t)tile * nbg * 8 + row;
    float acc = 0.0f, corr = 0.0f;
    for (int bg = 0; bg < nbg; bg++) {
        const u32x4 w = kr_ldg_nt(q + (size_t)bg * 64);
        const u32x4 hd = kr_ldg_nt(h + (size_t)bg * 8);
        const uint32_t wb[4] = {w.x, w.y, w.z, w.w}, hb[4] = {hd.x, hd.y, hd.z, hd.w};
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int s = bg * 8 + u;
            if (s < nb) {
                const uint32_t two = (wb[u >> 1] >> ((u & 1) * 16)) & 0xFFFFu;        // bytes qs[2l], qs[2l+1]
                const uint32_t nib = (two & 0x0F0Fu) | (((two >> 4) & 0x0F0Fu) << 16);  // {lo(b0), lo(b1), hi(b0), hi(b1)} = elems 2l,2l+1,16+2l,17+2l
                const float d = gg_f16((hb[u >> 1] >> ((u & 1) * 16)) & 0xFFFFu);
                const u32x2 r = *reinterpret_cast<const u32x2*>(A.rec + ((size_t)s * 8 + l) * 2);
                const int iv = (__builtin_amdgcn_sdot4((int)nib, (int)r.x, 0, false) << 8) + (int)__builtin_amdgcn_udot4(nib, r.y, 0u, false);
                const float as = A.scale[s];
                acc = __builtin_fmaf((float)iv, d * as, acc);
                corr += d * 8.0f * as * (float)A.sum[s];
            }
        }
    }
    return gg_hsum8(acc) - corr;
}

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

__device__ __forceinline__ bool gg_int_path(int t) { return t == GG_Q4_K || t == GG_Q8_0 || t == GG_Q4_0; }

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

__device__ __forceinline__ GgMat gg_expert_mat(const GgMat& base, int e) {
    GgMat m = base;
    m.q = reinterpret_cast<const char*>(base.q) + (size_t)e * base.q_stride;
    m.h = reinterpret_cast<const char*>(base.h) + (size_t)e * base.h_stride;
    return m;
}

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
