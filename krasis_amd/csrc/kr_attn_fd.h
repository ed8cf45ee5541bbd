// kr_attn_fd.h -- FAST (tolerance) mode of decode attention over long caches, first form: exact scores + split-KV softmax / weighted sum.
// Since the flash-decode kernels (kr_attn_flash.hip for GQA, the SPLIT form of kr_mla_flash.hip for MLA) this is the FALLBACK of the MLA path
// for geometries those do not cover (kr_mla.hip: the latent cache [position][klr] is the value stream of ALL heads, nkv = 1, G = nh <= 16).
//
// The exact kernels keep the reference's sequential order (sum of exponentials and p.v one position after the other, decode.rs:4236-4270 /
// :4326-4351), so their second launch is ONE workgroup per head walking the whole cache.  north_star asks for fp TOLERANCE outside the
// router, so a second numerics mode (kr_decode_set_attention_mode) keeps the exact per-position scores (the scores launch is parallel
// already) and replaces that launch by
//   kr_fd_partial_kernel : one workgroup per (256-position chunk, KV head): local maximum, exponentials, local sum and the chunk's
//                          sum_p p * v for ALL query heads that share the value rows (a row is read once per G heads), f32 throughout;
//   kr_fd_merge_kernel   : per head, the log-sum-exp merge of the chunk partials (+ the sigmoid gate and the o-projection image for GQA).
// Same products, same exp (the libm twin), a different summation order: the attention output differs in its last bits; after the INT16
// re-quantisation of the o-projection input the logits move by ~1e-4 relative (tests/test_attn_fast_gpu.py states 5e-4).
#pragma once
#include "kr_lds_optin.h"
#include "kr_device.h"
#include "kr_libm.h"
#include "kr_decode_ops.h"

#define KR_FD_CH 256

struct KrFdArgs {
    const KrStep* step; const float* sc_g;     // scores [nh][max_seq] (already scaled)
    const void* v_cache; int v_ld;             // value rows: v_ld elements per position, head group kvh reads columns [kvh * HD, +HD)
    float *fd_o, *fd_ml;                       // partials [nkv][chunks][G][HD], (max, sum) [nh][chunks][2]
    int nh, nkv;
    const float* gate; int gated; float* out; void* img_out;   // merge tail: out [nh][HD]; GQA: * sigmoid(gate), INT16 image of the o-projection input
};

template <int HD, bool FP8, int GMAX>
__global__ void __launch_bounds__(256) kr_fd_partial_kernel(const KrFdArgs a, int max_seq, int n_chunks) {
    constexpr int NDG = HD / 4, NPL = 256 / NDG;                       // 4 dims per thread; NPL position lanes
    __shared__ __attribute__((aligned(16))) float P[KR_FD_CH][GMAX];   // probabilities (relative to the chunk maximum), [position][head of the group]
    __shared__ float red[4][GMAX];
    extern __shared__ __attribute__((aligned(16))) float ored[];       // [NPL][G][HD] partial outputs of the position lanes
    const int c = blockIdx.x, kvh = blockIdx.y, t = threadIdx.x, G = a.nh / a.nkv;
    const int seq = a.step->pos + 1, p0 = c * KR_FD_CH;
    if (p0 >= seq) return;
    const int n = min(KR_FD_CH, seq - p0);
    // ---- probabilities of the chunk: thread t = position p0 + t, all G heads
    float sv[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; g++) sv[g] = (g < G && t < n) ? a.sc_g[(size_t)(kvh * G + g) * max_seq + p0 + t] : -__builtin_inff();
#pragma unroll
    for (int g = 0; g < GMAX; g++) {
        float m = sv[g];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if ((t & 63) == 0) red[t >> 6][g] = m;
    }
    __syncthreads();
    float mx[GMAX], ls[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; g++) {
        mx[g] = fmaxf(fmaxf(red[0][g], red[1][g]), fmaxf(red[2][g], red[3][g]));
        const float pv = (g < G && t < n) ? kr_expf(sv[g] - mx[g]) : 0.0f;
        P[t][g] = pv; ls[g] = pv;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GMAX; g++) {
        float v = ls[g];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if ((t & 63) == 0) red[t >> 6][g] = v;
    }
    __syncthreads();
    if (t < G) {
        const float l = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
        float* ml = a.fd_ml + ((size_t)(kvh * G + t) * n_chunks + c) * 2;
        float mt = mx[0];
#pragma unroll
        for (int g = 1; g < GMAX; g++) if (g == t) mt = mx[g];        // (mx[] is uniform across threads; no dynamic register indexing)
        ml[0] = mt; ml[1] = l;
    }
    // ---- sum_p p * v: thread = (4 dims dg, position lane pl); positions pl, pl + NPL, ...
    const int dg = t % NDG, pl = t / NDG, esz = FP8 ? 1 : 2;
    const unsigned char* vb = reinterpret_cast<const unsigned char*>(a.v_cache) + ((size_t)p0 * a.v_ld + (size_t)kvh * HD + dg * 4) * esz;
    float acc[GMAX][4];
#pragma unroll
    for (int g = 0; g < GMAX; g++) { acc[g][0] = 0.0f; acc[g][1] = 0.0f; acc[g][2] = 0.0f; acc[g][3] = 0.0f; }
    constexpr int UN = 4;
    for (int i0 = pl; i0 < n; i0 += NPL * UN) {
        float v[UN][4];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int i = i0 + u * NPL;
            if (i < n) {
                if (FP8) {
                    const uint32_t w = *reinterpret_cast<const uint32_t*>(vb + (size_t)i * a.v_ld);
                    v[u][0] = __builtin_amdgcn_cvt_f32_fp8((int)w, 0); v[u][1] = __builtin_amdgcn_cvt_f32_fp8((int)w, 1);
                    v[u][2] = __builtin_amdgcn_cvt_f32_fp8((int)w, 2); v[u][3] = __builtin_amdgcn_cvt_f32_fp8((int)w, 3);
                } else {
                    const u32x2 w = *reinterpret_cast<const u32x2*>(vb + (size_t)i * a.v_ld * 2);
                    v[u][0] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w.x & 0xFFFFu)); v[u][1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w.x >> 16));
                    v[u][2] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w.y & 0xFFFFu)); v[u][3] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w.y >> 16));
                }
            } else { v[u][0] = 0.0f; v[u][1] = 0.0f; v[u][2] = 0.0f; v[u][3] = 0.0f; }
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int i = min(i0 + u * NPL, KR_FD_CH - 1);
#pragma unroll
            for (int g4 = 0; g4 < GMAX / 4; g4++) {
                if (g4 * 4 < G) {
                    const float4 pq = *reinterpret_cast<const float4*>(&P[i][g4 * 4]);
                    const float pp[4] = {pq.x, pq.y, pq.z, pq.w};
#pragma unroll
                    for (int gg = 0; gg < 4; gg++)
#pragma unroll
                        for (int d = 0; d < 4; d++) acc[g4 * 4 + gg][d] = __builtin_fmaf(pp[gg], v[u][d], acc[g4 * 4 + gg][d]);
                }
            }
        }
    }
    // ---- position lanes -> one partial per (head, dim)
#pragma unroll
    for (int g = 0; g < GMAX; g++)
        if (g < G) *reinterpret_cast<float4*>(ored + ((size_t)(pl * G + g) * HD + dg * 4)) = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
    __syncthreads();
    float* po = a.fd_o + ((size_t)(kvh * n_chunks + c) * G) * HD;
    for (int o = t; o < G * HD; o += 256) {
        float sum = 0.0f;
#pragma unroll
        for (int q = 0; q < NPL; q++) sum += ored[(size_t)q * G * HD + o];
        po[o] = sum;
    }
}

template <int HD>
__global__ void __launch_bounds__(1024) kr_fd_merge_kernel(const KrFdArgs a, int n_chunks) {
    // thread = (dim d, chunk phase): 1024 / HD phases walk interleaved chunks with independent partial sums (the first version's single
    // dependent fma per chunk made this launch as long as the partial launch: 38 us at 128 chunks)
    __shared__ float wc[1024]; __shared__ float qs[1024]; __shared__ float red[32];
    const int h = blockIdx.x, t = threadIdx.x, G = a.nh / a.nkv, kvh = h / G, g = h % G;
    const int seq = a.step->pos + 1, nc = min((seq + KR_FD_CH - 1) / KR_FD_CH, 1024);
    const float* ml = a.fd_ml + (size_t)h * n_chunks * 2;
    const float mv = t < nc ? ml[t * 2] : -__builtin_inff(), lv = t < nc ? ml[t * 2 + 1] : 0.0f;
    float mx = mv;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int i = 1; i < 16; i++) mx = fmaxf(mx, red[i]);
    const float wv = t < nc ? kr_expf(mv - mx) : 0.0f;
    wc[t] = wv;
    float l = wv * lv;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) l += __shfl_xor(l, off);
    if ((t & 63) == 0) red[16 + (t >> 6)] = l;
    __syncthreads();
    float lt = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) lt += red[16 + i];
    const float inv = 1.0f / lt;
    constexpr int NPH = 1024 / HD;
    const int d = t % HD, ph = t / HD;
    const float* ob = a.fd_o + ((size_t)kvh * n_chunks * G + g) * HD + d;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int c = ph;
    for (; c + 3 * NPH < nc; c += 4 * NPH) {
        const float v0 = ob[(size_t)c * G * HD], v1 = ob[(size_t)(c + NPH) * G * HD], v2 = ob[(size_t)(c + 2 * NPH) * G * HD], v3 = ob[(size_t)(c + 3 * NPH) * G * HD];
        s0 = __builtin_fmaf(wc[c], v0, s0); s1 = __builtin_fmaf(wc[c + NPH], v1, s1); s2 = __builtin_fmaf(wc[c + 2 * NPH], v2, s2); s3 = __builtin_fmaf(wc[c + 3 * NPH], v3, s3);
    }
    for (; c < nc; c += NPH) s0 = __builtin_fmaf(wc[c], ob[(size_t)c * G * HD], s0);
    float o = (s0 + s1) + (s2 + s3);
    if (NPH > 1) {
        __syncthreads();
        qs[t] = o;
        __syncthreads();
        o = 0.0f;
        if (t < HD) for (int p = 0; p < NPH; p++) o += qs[p * HD + t];
        __syncthreads();
    }
    if (t < HD) {
        o *= inv;
        if (a.gated) { const float gt = a.gate[(size_t)h * HD + t]; o *= 1.0f / (1.0f + kr_expf(-gt)); }
        a.out[(size_t)h * HD + t] = o;
        if (a.img_out) qs[t] = o;
    }
    if (a.img_out) {   // hd % 128 == 0: the head's output is hd/128 whole quantization groups of the o-projection's input
        __syncthreads();
        const KrActLds Lg = kr_carve_lds(reinterpret_cast<u32x4*>(a.img_out), a.nh * HD, false);
        constexpr int nch = HD / 8;
        if (t < nch) {
            float v8[8];
            kr_load8(qs, t, v8);
            float mx8 = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; i++) mx8 = fmaxf(mx8, fabsf(v8[i]));
            float scale, qinv;
            kr_group_scale(mx8, scale, qinv);
            int q8[8];
            kr_quant8<false>(v8, qinv, q8);
            const int gc = h * nch + t;
            kr_store_chunk<false>(Lg, gc, q8);
            if ((gc & 15) == 0) Lg.ascale[gc >> 4] = scale;
        }
    }
}

// opt-in to more than 64 KiB of dynamic LDS (G = 16 heads: 64 KiB of partials + 16 KiB of probabilities): a per-device function attribute,
// set OUTSIDE graph capture (kr_mla_attn_prepare)
template <int HD, int GMAX>
static void kr_fd_prepare() {
    (void)kr_lds_optin((const void*)kr_fd_partial_kernel<HD, false, GMAX>, 96 * 1024);
    (void)kr_lds_optin((const void*)kr_fd_partial_kernel<HD, true, GMAX>, 96 * 1024);
}
// G <= GMAX is the caller's check
template <int HD, int GMAX>
static void kr_launch_fd(const KrFdArgs& a, int fp8, int max_seq, hipStream_t s) {
    const int nchunks = (max_seq + KR_FD_CH - 1) / KR_FD_CH, G = a.nh / a.nkv;
    const size_t lds = (size_t)256 * G * 4 * 4;      // [NPL][G][HD] floats with NPL * HD == 1024
    dim3 grid(nchunks, a.nkv);
    if (fp8) hipLaunchKernelGGL((kr_fd_partial_kernel<HD, true, GMAX>), grid, dim3(256), lds, s, a, max_seq, nchunks);
    else hipLaunchKernelGGL((kr_fd_partial_kernel<HD, false, GMAX>), grid, dim3(256), lds, s, a, max_seq, nchunks);
    hipLaunchKernelGGL(kr_fd_merge_kernel<HD>, dim3(a.nh), dim3(1024), 0, s, a, nchunks);
}
