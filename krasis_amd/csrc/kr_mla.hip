// kr_mla.hip -- MLA (multi-head latent attention, DeepSeek-V2/V3 family) decode step on gfx950.
//
// Replaces (bit-exactly) the MLA arm of decode_step (src/decode.rs:2993-3252) and its AVX2 helpers:
//   mla_absorb_wkc_avx2 (:4508)  q_abs[h][j] = fma(q_nope[h][i], w_kc[h][i][j], .) for i ascending
//   mla_attn_dot_fp16_avx2 (:4286)  two 8-lane fma accumulators over alternating 8-blocks, (acc0+acc1), hsum
//   mla_weighted_sum_fp16_avx2 (:4326)  out[j] = fma(p_t, ckv[t][j], out[j]) for t ascending
//   mla_project_wvc_avx2 (:4555)  same two-accumulator dot per output row
// and the scalar pieces in between (sequential-sum RMSNorm of the compressed KV, de-interleave + RoPE, libm softmax).
//
// Three launches per layer: prep (norm, RoPE, FP16 cache append, w_kc absorption spread over nh*klr/64 workgroups so the 4 MiB of
// w_kc streams from many CUs), attention (one workgroup per head), w_vc projection (nh*vhd/8 workgroups, 4 MiB of w_vc).
#include "kr_device.h"
#include "kr_libm.h"
#include "kr_decode_ops.h"
#include <hip/hip_fp16.h>

__device__ __forceinline__ float kr_mla_hsum8(float v) {   // lo+hi, movehdup, movehl (same tree as every hsum in decode.rs)
    v = v + __shfl_xor(v, 4);
    v = v + __shfl_xor(v, 1);
    v = v + __shfl_xor(v, 2);
    return v;
}
__device__ __forceinline__ float kr_h2f(uint16_t h) { return __half2float(__ushort_as_half(h)); }
// cache element i of a row / of the whole cache: FP16 (reference CPU decode) or E4M3 (the GPU cache dtype, extended to the latent cache)
template <bool FP8> __device__ __forceinline__ float kr_mla_ld(const void* base, size_t i) {
    if (FP8) return kr_e4m3_to_f32(reinterpret_cast<const uint8_t*>(base)[i]);
    return kr_h2f(reinterpret_cast<const uint16_t*>(base)[i]);
}
template <bool FP8> __device__ __forceinline__ void kr_mla_st(void* base, size_t i, float v) {
    if (FP8) reinterpret_cast<uint8_t*>(base)[i] = kr_f32_to_e4m3(v);
    else reinterpret_cast<uint16_t*>(base)[i] = __half_as_ushort(__float2half_rn(v));
}

// 16 cooperating lanes (c = lane & 15) evaluate mla_attn_dot_fp16_avx2 / the w_vc row dot: chain (a = c >> 3, l = c & 7) owns the
// 8-blocks i with i % 2 == a (an odd trailing block goes to accumulator 0), ascending.  Every lane of the 16 returns the result.
template <typename LoadB>
__device__ __forceinline__ float kr_dot2acc(const float* q, LoadB loadb, int dim, int c) {
    const int n8 = dim >> 3, a = c >> 3, l = c & 7, paired = (n8 >> 1) << 1;
    float acc = 0.0f;
    for (int i = a; i < paired; i += 2) acc = __builtin_fmaf(q[i * 8 + l], loadb(i * 8 + l), acc);
    if ((n8 & 1) && a == 0) acc = __builtin_fmaf(q[(n8 - 1) * 8 + l], loadb((n8 - 1) * 8 + l), acc);
    const float other = __shfl_xor(acc, 8);
    const float s8 = a == 0 ? acc + other : other + acc;   // _mm256_add_ps(acc0, acc1)
    return kr_mla_hsum8(s8);
}

// ---- launch 1: prep --------------------------------------------------------------------------------------------------
// blocks [0, nh*klr/64): absorb tile (h, jt); the jt == 0 block of each head also de-interleaves + ropes q_pe[h].
// last block: kv_a RMSNorm (sequential sum), k_pe de-interleave + RoPE, FP16 cache append at `pos`.
// per-token view of the argument block (decode: the single token; prompt pass: token blockIdx.y / .z of the chunk)
__device__ __forceinline__ int kr_mla_token(KrMlaArgs& a, int tk) {
    if (a.step) return a.step->pos;
    a.kv_out += (size_t)tk * a.ld_kv; a.q_full += (size_t)tk * a.ld_q;
    a.q_abs += (size_t)tk * a.nh * a.klr; a.q_pe += (size_t)tk * a.nh * a.rd; a.attn_lat += (size_t)tk * a.nh * a.klr; a.v_proj += (size_t)tk * a.nh * a.vhd;
    return a.pos0 + tk;
}
template <bool FP8>
__global__ void __launch_bounds__(64) kr_mla_prep_kernel(KrMlaArgs a) {
    __shared__ float sh[640];
    const int pos = kr_mla_token(a, blockIdx.y);
    const int tiles = a.klr / 64, nb_abs = a.nh * tiles, hd = a.nd + a.rd, half = a.rd / 2;
    const int t = threadIdx.x;
    if ((int)blockIdx.x < nb_abs) {
        const int h = blockIdx.x / tiles, jt = blockIdx.x % tiles, j = jt * 64 + t;
        const float* qh = a.q_full + (size_t)h * hd;
        for (int i = t; i < a.nd; i += 64) sh[i] = qh[i];
        __syncthreads();
        const float* w = a.w_kc + (size_t)h * a.nd * a.klr + j;
        float o = 0.0f;
        int i = 0;
        for (; i + 16 <= a.nd; i += 16) {
            float wv[16];
#pragma unroll
            for (int u = 0; u < 16; u++) wv[u] = __builtin_nontemporal_load(w + (size_t)(i + u) * a.klr);
#pragma unroll
            for (int u = 0; u < 16; u++) o = __builtin_fmaf(sh[i + u], wv[u], o);
        }
        for (; i < a.nd; i++) o = __builtin_fmaf(sh[i], w[(size_t)i * a.klr], o);
        a.q_abs[(size_t)h * a.klr + j] = o;
        if (jt == 0 && t < half) {   // decode.rs:3113-3128
            const float x1 = qh[a.nd + 2 * t], x2 = qh[a.nd + 2 * t + 1];
            const float c = a.rope_cos[(size_t)pos * half + t], s = a.rope_sin[(size_t)pos * half + t];
            a.q_pe[(size_t)h * a.rd + t] = x1 * c - x2 * s;
            a.q_pe[(size_t)h * a.rd + half + t] = x2 * c + x1 * s;
        }
        return;
    }
    // ---- compressed KV ----
    float* x = sh;                                  // klr <= 576 values + 1 slot for rms
    for (int i = t; i < a.klr; i += 64) x[i] = a.kv_out[i];
    __syncthreads();
    if (t == 0) {                                   // decode.rs:3025-3028: scalar sequential sum, mul and add separate
        float ss = 0.0f; int i = 0;
        for (; i + 8 <= a.klr; i += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { v[u] = x[i + u]; v[u] = v[u] * v[u]; }
#pragma unroll
            for (int u = 0; u < 8; u++) ss += v[u];
        }
        for (; i < a.klr; i++) ss += x[i] * x[i];
        sh[639] = 1.0f / sqrtf(ss / (float)a.klr + a.eps);
    }
    __syncthreads();
    const float rms = sh[639];
    for (int i = t; i < a.klr; i += 64) {
        const float v = x[i] * (rms * a.kv_a_norm[i]);                                     // x *= rms * w (decode.rs:3030)
        kr_mla_st<FP8>(a.ckv_cache, (size_t)pos * a.klr + i, v);
    }
    if (t < half) {                                                                        // decode.rs:3098-3107, 3131-3140
        const float x1 = a.kv_out[a.klr + 2 * t], x2 = a.kv_out[a.klr + 2 * t + 1];
        const float c = a.rope_cos[(size_t)pos * half + t], s = a.rope_sin[(size_t)pos * half + t];
        kr_mla_st<FP8>(a.kpe_cache, (size_t)pos * a.rd + t, x1 * c - x2 * s);
        kr_mla_st<FP8>(a.kpe_cache, (size_t)pos * a.rd + half + t, x2 * c + x1 * s);
    }
}

// ---- launch 2: attention, one workgroup (512 threads) per head.  dynamic LDS: klr + rd + seq_max + 8 floats --------------
template <bool FP8>
__global__ void __launch_bounds__(512) kr_mla_attn_kernel(KrMlaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[12];
    const int seq = kr_mla_token(a, blockIdx.y) + 1;
    float* qa = lds; float* qp = qa + a.klr; float* sc = qp + a.rd;
    const int h = blockIdx.x, t = threadIdx.x;
    for (int i = t; i < a.klr; i += 512) qa[i] = a.q_abs[(size_t)h * a.klr + i];
    for (int i = t; i < a.rd; i += 512) qp[i] = a.q_pe[(size_t)h * a.rd + i];
    __syncthreads();
    const int c = t & 15;
    for (int s = t >> 4; s < seq; s += 32) {
        const size_t cko = (size_t)s * a.klr, kpo = (size_t)s * a.rd;
        float v = kr_dot2acc(qa, [&](int i) { return kr_mla_ld<FP8>(a.ckv_cache, cko + i); }, a.klr, c);
        v += kr_dot2acc(qp, [&](int i) { return kr_mla_ld<FP8>(a.kpe_cache, kpo + i); }, a.rd, c);
        v *= a.sm_scale;
        if (c == 0) sc[s] = v;
    }
    __syncthreads();
    float mx = -__builtin_inff();
    for (int s = t; s < seq; s += 512) mx = fmaxf(mx, sc[s]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < 8; w++) mx = fmaxf(mx, red[w]);
    for (int s = t; s < seq; s += 512) sc[s] = kr_expf(sc[s] - mx);
    __syncthreads();
    if (t == 0) {
        float se = 0.0f; int s = 0;
        for (; s + 8 <= seq; s += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = sc[s + u];
#pragma unroll
            for (int u = 0; u < 8; u++) se += v[u];
        }
        for (; s < seq; s++) se += sc[s];
        red[8] = 1.0f / se;
    }
    __syncthreads();
    const float inv = red[8];
    for (int s = t; s < seq; s += 512) sc[s] *= inv;
    __syncthreads();
    for (int j = t; j < a.klr; j += 512) {
        float o = 0.0f;
        int s = 0;
        for (; s + 16 <= seq; s += 16) {      // 16 independent cache loads in flight, then the dependent fmas in position order
            float vv[16];
#pragma unroll
            for (int u = 0; u < 16; u++) vv[u] = kr_mla_ld<FP8>(a.ckv_cache, (size_t)(s + u) * a.klr + j);
#pragma unroll
            for (int u = 0; u < 16; u++) o = __builtin_fmaf(sc[s + u], vv[u], o);
        }
        for (; s < seq; s++) o = __builtin_fmaf(sc[s], kr_mla_ld<FP8>(a.ckv_cache, (size_t)s * a.klr + j), o);
        a.attn_lat[(size_t)h * a.klr + j] = o;
    }
}

// ---- launch 3: v_projected[h][o] = w_vc[h][o][:] . attn_lat[h][:]   grid (vhd/8, nh), 128 threads = 8 outputs x 16 lanes ----
__global__ void __launch_bounds__(128) kr_mla_wvc_kernel(KrMlaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    (void)kr_mla_token(a, blockIdx.z);
    const int h = blockIdx.y, t = threadIdx.x, o = blockIdx.x * 8 + (t >> 4);
    for (int i = t; i < a.klr; i += 128) lds[i] = a.attn_lat[(size_t)h * a.klr + i];
    __syncthreads();
    if (o >= a.vhd) return;   // whole 16-lane groups leave together
    const float* wr = a.w_vc + ((size_t)h * a.vhd + o) * a.klr;
    const float v = kr_dot2acc(wr, [&](int i) { return lds[i]; }, a.klr, t & 15);
    if ((t & 15) == 0) a.v_proj[(size_t)h * a.vhd + o] = v;
}

// plain sequential RMSNorm (decode.rs:3053-3062, q_a_layernorm of the LoRA query path); one workgroup, in place
__global__ void __launch_bounds__(256) kr_rmsnorm_seq_kernel(float* __restrict__ x, const float* __restrict__ w, int n, float eps, int ld) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    x += (size_t)blockIdx.x * ld;                // prompt pass: one workgroup per token row
    for (int i = t; i < n; i += 256) lds[i] = x[i];
    __syncthreads();
    if (t == 0) {
        float ss = 0.0f; int i = 0;
        for (; i + 8 <= n; i += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { v[u] = lds[i + u]; v[u] = v[u] * v[u]; }
#pragma unroll
            for (int u = 0; u < 8; u++) ss += v[u];
        }
        for (; i < n; i++) ss += lds[i] * lds[i];
        lds[n] = 1.0f / sqrtf(ss / (float)n + eps);
    }
    __syncthreads();
    const float rms = lds[n];
    for (int i = t; i < n; i += 256) x[i] = lds[i] * (rms * w[i]);
}

void kr_launch_mla(const KrMlaArgs& a, int max_seq, hipStream_t s, int n_tok) {
    if (a.kv_fp8) {
        hipLaunchKernelGGL(kr_mla_prep_kernel<true>, dim3(a.nh * (a.klr / 64) + 1, n_tok), dim3(64), 0, s, a);
        hipLaunchKernelGGL(kr_mla_attn_kernel<true>, dim3(a.nh, n_tok), dim3(512), (size_t)(a.klr + a.rd + max_seq + 8) * 4, s, a);
    } else {
        hipLaunchKernelGGL(kr_mla_prep_kernel<false>, dim3(a.nh * (a.klr / 64) + 1, n_tok), dim3(64), 0, s, a);
        hipLaunchKernelGGL(kr_mla_attn_kernel<false>, dim3(a.nh, n_tok), dim3(512), (size_t)(a.klr + a.rd + max_seq + 8) * 4, s, a);
    }
    hipLaunchKernelGGL(kr_mla_wvc_kernel, dim3((a.vhd + 7) / 8, a.nh, n_tok), dim3(128), (size_t)a.klr * 4, s, a);
}
void kr_launch_rmsnorm_seq(float* x, const float* w, int n, float eps, hipStream_t s, int rows, int ld) {
    hipLaunchKernelGGL(kr_rmsnorm_seq_kernel, dim3(rows), dim3(256), (size_t)(n + 4) * 4, s, x, w, n, eps, ld);
}
