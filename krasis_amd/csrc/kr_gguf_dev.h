// kr_gguf_dev.h -- device side of the native GGUF block experts, shared by kr_gguf.hip (the exact kernels: moe_forward_gguf bit for bit) and
// kr_decode_fast.hip (KR_DECODE_FAST: the same products with the row's blocks split over two waves and the select / activation / combine folded in).
#pragma once
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
__host__ __device__ static inline size_t gg_lds_bytes(int K, bool want_f32) { return (size_t)(K / 32) * 64 + (size_t)(K / 32) * 8 + (want_f32 ? (size_t)K * 4 : 0) + 16; }

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
// [part, parts): the slice of the row's blocks this call walks (tolerance decode: two waves share a row tile and add their results -- (0, 1) is the whole row,
// the reference's order)
__device__ __forceinline__ float gg_tile_q4k(const GgMat& m, int tile, const GgAct& A, int lane, int part = 0, int parts = 1) {
    const int l = lane & 7, row = lane >> 3;
    const int nb = m.K / 256, per = (nb + parts - 1) / parts, b_lo = part * per, b_hi = b_lo + per < nb ? b_lo + per : nb;
    const u32x4* q = reinterpret_cast<const u32x4*>(m.q) + (size_t)tile * nb * 64 + lane;
    const u32x4* h = reinterpret_cast<const u32x4*>(m.h) + (size_t)tile * nb * 8 + row;
    float acc = 0.0f, corr = 0.0f;
    for (int b0 = b_lo; b0 < b_hi; b0 += GG_PF) {
        u32x4 wv[GG_PF], hv[GG_PF];
#pragma unroll
        for (int u = 0; u < GG_PF; u++) { const int bb = b0 + u < b_hi ? b0 + u : (b_hi > 0 ? b_hi - 1 : 0); wv[u] = kr_ldg_nt(q + (size_t)bb * 64); hv[u] = kr_ldg_nt(h + (size_t)bb * 8); }
#pragma unroll
        for (int u = 0; u < GG_PF; u++) {
            const int b = b0 + u;
            if (b < b_hi) {
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

__device__ __forceinline__ float gg_tile_q8_0(const GgMat& m, int tile, const GgAct& A, int lane, int part = 0, int parts = 1) {
    const int l = lane & 7, row = lane >> 3;
    const int nb = m.K / 32, nbg = (nb + 3) / 4, per = (nbg + parts - 1) / parts, g_lo = part * per, g_hi = g_lo + per < nbg ? g_lo + per : nbg;
    const u32x4* q = reinterpret_cast<const u32x4*>(m.q) + (size_t)tile * nbg * 64 + lane;
    const u32x2* h = reinterpret_cast<const u32x2*>(m.h) + (size_t)tile * nbg * 8 + row;
    float acc = 0.0f;
    for (int g0 = g_lo; g0 < g_hi; g0 += GG_PF) {      // batched requests, as gg_tile_q4k
        u32x4 wv[GG_PF]; u32x2 hv[GG_PF];
#pragma unroll
        for (int v = 0; v < GG_PF; v++) { const int gg = g0 + v < g_hi ? g0 + v : (g_hi > 0 ? g_hi - 1 : 0); wv[v] = kr_ldg_nt(q + (size_t)gg * 64); hv[v] = h[(size_t)gg * 8]; }
#pragma unroll
        for (int v = 0; v < GG_PF; v++) {
            const int bg = g0 + v;
            if (bg < g_hi) {
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

__device__ __forceinline__ float gg_tile_q4_0(const GgMat& m, int tile, const GgAct& A, int lane, int part = 0, int parts = 1) {
    const int l = lane & 7, row = lane >> 3;
    const int nb = m.K / 32, nbg = (nb + 7) / 8, per = (nbg + parts - 1) / parts, g_lo = part * per, g_hi = g_lo + per < nbg ? g_lo + per : nbg;
    const u32x4* q = reinterpret_cast<const u32x4*>(m.q) + (size_t)tile * nbg * 64 + lane;
    const u32x4* h = reinterpret_cast<const u32x4*>(m.h) + (size_t)tile * nbg * 8 + row;
    float acc = 0.0f, corr = 0.0f;
    for (int bg = g_lo; bg < g_hi; bg++) {
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

__device__ __forceinline__ bool gg_int_path(int t) { return t == GG_Q4_K || t == GG_Q8_0 || t == GG_Q4_0; }

__device__ __forceinline__ GgMat gg_expert_mat(const GgMat& base, int e) {
    GgMat m = base;
    m.q = reinterpret_cast<const char*>(base.q) + (size_t)e * base.q_stride;
    m.h = reinterpret_cast<const char*>(base.h) + (size_t)e * base.h_stride;
    return m;
}

