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
#include "kr_lds_optin.h"
#include "kr_router.h"
#include "kr_device.h"
#include "kr_libm.h"
#include "kr_decode_ops.h"
#include "kr_decode_fast.h"
#include "kr_prefill_ops.h"
#include "kr_attn_fd.h"
#include <hip/hip_fp16.h>

#ifdef KR_TIMING   // tools/probes/mla_timing.hip: wall-clock stamps (10 ns units) of thread 0 of workgroup 0, no-op in the product build
__device__ unsigned long long kr_mstamps[32];
#define KR_MSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) kr_mstamps[i] = wall_clock64(); } while (0)
#else
#define KR_MSTAMP(i) do { } while (0)
#endif

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
template <bool FP8> __device__ __forceinline__ float kr_stage_val(const unsigned char* row, int i) {   // element i of a staged (LDS) row
    if (FP8) return kr_e4m3_to_f32(row[i]);
    _Float16 hv; __builtin_memcpy(&hv, row + 2 * i, 2);
    return (float)hv;
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
    const int tiles = a.absorb_done ? 1 : a.klr / 64, nb_abs = a.nh * tiles, hd = a.nd + a.rd, half = a.rd / 2;      // absorb_done: one workgroup per head (rope of q_pe only)
    const int t = threadIdx.x;
    if ((int)blockIdx.x < nb_abs) {
        const int h = blockIdx.x / tiles, jt = blockIdx.x % tiles, j = jt * 64 + t;
        const float* qh = a.q_full + (size_t)h * hd;
        if (!a.absorb_done) {
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
        }
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

// ---- launch 2, specialised: latent-cache rows staged through LDS ---------------------------------------------------------------
// Same arithmetic and order as kr_mla_attn_kernel above; what changes is where the operands wait.  klr / 8, rd / 8 and the cache dtype
// are template parameters (no run-time guards around the per-lane reads -- each would end a basic block with a wait), and the
// compressed-KV + rope rows of 64 positions at a time are fetched by all 512 threads with 16-byte buffer loads (num_records = the
// current length: rows past it read as zero) while the previous 64 are consumed from LDS.  The generic kernel issued one 2-byte global
// load per element of every dot product and of every weighted-sum step: 215 ns per cached position per layer on a V2-Lite shape.
#define KR_MLA_ROWS 64
#define KR_MLA_HG 4          // heads that share a staged row in the decode scores launch
// Scores + softmax + weighted sum of one head in one workgroup: short caches of the decode step (the score row is LDS-resident) and the prompt
// pass (grid y = token).  Long decode caches split the work: kr_mla_scores_kernel + kr_mla_pv_kernel below.
template <bool FP8, int NBC, int NBR>
__global__ void __launch_bounds__(512) kr_mla_attn_staged_kernel(KrMlaArgs a, int max_seq) {
    const int lds_seq = max_seq;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[12];
    constexpr int klr = NBC * 8, rd = NBR * 8, esz = FP8 ? 1 : 2;
    constexpr int CPR_C = klr * esz / 16, CPR_R = rd * esz / 16;   // 16-byte chunks per staged row (latent part, rope part)
    constexpr int pitch = (klr + rd) * esz + 16;
    constexpr int NLC = KR_MLA_ROWS * CPR_C / 512;                                     // latent-row chunks per thread per stage (8 at klr 512 / FP16)
    static_assert(KR_MLA_ROWS * CPR_C % 512 == 0 && KR_MLA_ROWS * CPR_R <= 512, "chunk split");
    const int seq = kr_mla_token(a, blockIdx.y) + 1;
    float* qa = lds; float* qp = qa + klr; float* sc = qp + rd;
    unsigned char* stage = reinterpret_cast<unsigned char*>(sc) + ((((size_t)lds_seq + 40) * 4 + 15) & ~(size_t)15);
    const int h = blockIdx.x, t = threadIdx.x;
    for (int i = t; i < klr; i += 512) qa[i] = a.q_abs[(size_t)h * klr + i];
    for (int i = t; i < rd; i += 512) qp[i] = a.q_pe[(size_t)h * rd + i];
    const __amdgpu_buffer_rsrc_t srd_c = __builtin_amdgcn_make_buffer_rsrc(a.ckv_cache, 0, seq * klr * esz, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_r = __builtin_amdgcn_make_buffer_rsrc(a.kpe_cache, 0, seq * rd * esz, 0x00020000);
    // Every thread fetches NLC chunks of the latent rows and (threads below ROWS * CPR_R) one chunk of the rope rows: UNCONDITIONAL loads
    // (a per-thread guard would put each load behind an exec-mask branch whose join waits for it) -- rows at or past the current length
    // are outside the descriptors and read as zero.  Two stages are in flight (register sets r0 / r1): one stage's compute (~1.2 us)
    // does not cover a memory round trip.
    struct Regs { u32x4 c[NLC]; u32x4 r; };
    const int rrow = t / CPR_R, rcol = t % CPR_R;
    const bool has_r = t < KR_MLA_ROWS * CPR_R;
    auto issue = [&](Regs& R, int s0) {
#pragma unroll
        for (int i = 0; i < NLC; i++) {
            const int c = t + 512 * i, r = c / CPR_C, col = c % CPR_C;
            R.c[i] = __builtin_amdgcn_raw_buffer_load_b128(srd_c, (s0 + r) * klr * esz + col * 16, 0, 0);
        }
        R.r = __builtin_amdgcn_raw_buffer_load_b128(srd_r, has_r ? (s0 + rrow) * rd * esz + rcol * 16 : 0x7FFFFFF0, 0, 0);
    };
    auto commit = [&](const Regs& R) {
#pragma unroll
        for (int i = 0; i < NLC; i++) {
            const int c = t + 512 * i, r = c / CPR_C, col = c % CPR_C;
            *reinterpret_cast<u32x4*>(stage + r * pitch + col * 16) = R.c[i];
        }
        if (has_r) *reinterpret_cast<u32x4*>(stage + rrow * pitch + klr * esz + rcol * 16) = R.r;
    };
    Regs r0, r1;
    const int nst = (seq + KR_MLA_ROWS - 1) / KR_MLA_ROWS;
    KR_MSTAMP(0);
    issue(r0, 0); issue(r1, KR_MLA_ROWS);
    __syncthreads();
    KR_MSTAMP(1);
    // ---- scores: 16 lanes per position (mla_attn_dot_fp16_avx2): lane c = (accumulator a2 = c >> 3, AVX lane l = c & 7) owns the 8-blocks
    // i % 2 == a2 (an odd trailing block goes to accumulator 0), ascending; (acc0 + acc1) then the 8-lane hsum; latent dot + rope dot
    const int c16 = t & 15, a2 = c16 >> 3, l = c16 & 7, g = t >> 4;
    constexpr int NQC = (NBC + 1) / 2, NQR = (NBR + 1) / 2;
    float qc[NQC], qr[NQR];
#pragma unroll
    for (int u = 0; u < NQC; u++) { const int i = 2 * u + a2; qc[u] = (i < NBC && (i < (NBC & ~1) || a2 == 0)) ? qa[i * 8 + l] : 0.0f; }
#pragma unroll
    for (int u = 0; u < NQR; u++) { const int i = 2 * u + a2; qr[u] = (i < NBR && (i < (NBR & ~1) || a2 == 0)) ? qp[i * 8 + l] : 0.0f; }
    static_assert(NBC % 2 == 0 && NBR % 2 == 0, "odd block counts take the generic kernel (a zero query element is not a skipped fma)");
    for (int st = 0; st < nst; st++) {
        if (st) __syncthreads();
        commit(r0);
        r0 = r1;
        issue(r1, (st + 2) * KR_MLA_ROWS);           // past the end: zeros, never committed
        __syncthreads();
        KR_MSTAMP(2);
        const int s0 = st * KR_MLA_ROWS;
#pragma unroll
        for (int k2 = 0; k2 < KR_MLA_ROWS / 32; k2++) {
            const int r = g + 32 * k2;
            if (s0 + r < seq) {
                const unsigned char* row = stage + r * pitch;
                float kc[NQC], kr[NQR];
#pragma unroll
                for (int u = 0; u < NQC; u++) kc[u] = kr_stage_val<FP8>(row, (2 * u + a2) * 8 + l);
#pragma unroll
                for (int u = 0; u < NQR; u++) kr[u] = kr_stage_val<FP8>(row + klr * esz, (2 * u + a2) * 8 + l);
                float acc = 0.0f;
#pragma unroll
                for (int u = 0; u < NQC; u++) acc = __builtin_fmaf(qc[u], kc[u], acc);
                float oth = __shfl_xor(acc, 8);
                float v = kr_mla_hsum8(a2 == 0 ? acc + oth : oth + acc);
                acc = 0.0f;
#pragma unroll
                for (int u = 0; u < NQR; u++) acc = __builtin_fmaf(qr[u], kr[u], acc);
                oth = __shfl_xor(acc, 8);
                v += kr_mla_hsum8(a2 == 0 ? acc + oth : oth + acc);
                v *= a.sm_scale;
                if (c16 == 0) sc[s0 + r] = v;
            }
        }
    }
    KR_MSTAMP(3);
    __syncthreads();
    KR_MSTAMP(4);
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
    const int seq32 = (seq + 31) & ~31;              // zero padding: the exponentials are >= +0, so s + 0.0f == s bit for bit
    if (t < seq32 - seq) sc[seq + t] = 0.0f;
    issue(r0, 0);                                    // the first stages of the weighted sum ride under the sequential sum
    issue(r1, KR_MLA_ROWS);
    __syncthreads();
    KR_MSTAMP(5);
    if (t == 0) red[8] = 1.0f / kr_seq_sum(sc, seq32);
    KR_MSTAMP(6);
    __syncthreads();
    const float inv = red[8];
    for (int s = t; s < seq; s += 512) sc[s] *= inv;
    // ---- weighted sum (mla_weighted_sum_fp16_avx2): thread j owns latent element j, one fma per position in ascending order
    float o = 0.0f;
    for (int st = 0; st < nst; st++) {
        __syncthreads();
        commit(r0);
        r0 = r1;
        issue(r1, (st + 2) * KR_MLA_ROWS);
        __syncthreads();
        KR_MSTAMP(7);
        if (t < klr) {
            const int s0 = st * KR_MLA_ROWS, n = min(KR_MLA_ROWS, seq - s0);
            for (int r = 0; r < n; r += 16) {        // n <= 64 and r % 16 == 0: rows r .. r + 15 are inside the stage
                float vv[16], pp[16];
#pragma unroll
                for (int u = 0; u < 16; u++) { vv[u] = kr_stage_val<FP8>(stage + (r + u) * pitch, t); pp[u] = sc[s0 + r + u]; }
#pragma unroll
                for (int u = 0; u < 16; u++) if (r + u < n) o = __builtin_fmaf(pp[u], vv[u], o);
            }
        }
    }
    KR_MSTAMP(8);
    if (t < klr) a.attn_lat[(size_t)h * klr + t] = o;
}

// ---- decode, long caches: scores of 8 heads x 64 positions per workgroup ---------------------------------------------------------
// The latent cache is shared by ALL heads, so a staged row should serve many of them: workgroup (position block x, head group y) stages
// 64 rows once and evaluates them against KR_MLA_HG heads (16 lanes per (position, head) pair; a lane group keeps its head, so its
// query slice stays in registers).  grid = (max_seq / 64, nh / KR_MLA_HG); blocks past the current length leave at once.  Same dot products,
// same order as the per-head kernels.
template <bool FP8, int NBC, int NBR>
__global__ void __launch_bounds__(512) kr_mla_scores_kernel(KrMlaArgs a, int max_seq) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int klr = NBC * 8, rd = NBR * 8, esz = FP8 ? 1 : 2, HG = KR_MLA_HG;
    constexpr int CPR_C = klr * esz / 16, CPR_R = rd * esz / 16, pitch = (klr + rd) * esz + 16, NLC = KR_MLA_ROWS * CPR_C / 512;
    const int seq = a.step->pos + 1, s0 = blockIdx.x * KR_MLA_ROWS, t = threadIdx.x;
    if (s0 >= seq) return;
    const int hg0 = blockIdx.y * HG, nhg = min(HG, a.nh - hg0);
    float* qa = lds; float* qp = qa + HG * klr;
    unsigned char* stage = reinterpret_cast<unsigned char*>(qp + HG * rd);
    const __amdgpu_buffer_rsrc_t srd_c = __builtin_amdgcn_make_buffer_rsrc(a.ckv_cache, 0, seq * klr * esz, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_r = __builtin_amdgcn_make_buffer_rsrc(a.kpe_cache, 0, seq * rd * esz, 0x00020000);
    u32x4 rc[NLC];
#pragma unroll
    for (int i = 0; i < NLC; i++) { const int c = t + 512 * i, r = c / CPR_C, col = c % CPR_C; rc[i] = __builtin_amdgcn_raw_buffer_load_b128(srd_c, (s0 + r) * klr * esz + col * 16, 0, 0); }
    const int rrow = t / CPR_R, rcol = t % CPR_R;
    const bool has_r = t < KR_MLA_ROWS * CPR_R;
    const u32x4 rr = __builtin_amdgcn_raw_buffer_load_b128(srd_r, has_r ? (s0 + rrow) * rd * esz + rcol * 16 : 0x7FFFFFF0, 0, 0);
    for (int i = t; i < nhg * klr; i += 512) qa[i] = a.q_abs[(size_t)hg0 * klr + i];
    for (int i = t; i < nhg * rd; i += 512) qp[i] = a.q_pe[(size_t)hg0 * rd + i];
#pragma unroll
    for (int i = 0; i < NLC; i++) { const int c = t + 512 * i, r = c / CPR_C, col = c % CPR_C; *reinterpret_cast<u32x4*>(stage + r * pitch + col * 16) = rc[i]; }
    if (has_r) *reinterpret_cast<u32x4*>(stage + rrow * pitch + klr * esz + rcol * 16) = rr;
    __syncthreads();
    const int c16 = t & 15, a2 = c16 >> 3, l = c16 & 7, g = t >> 4, hh = g % HG, prow = g / HG;
    if (hh >= nhg) return;
    constexpr int NQC = NBC / 2, NQR = NBR / 2;
    float qc[NQC], qr[NQR];
#pragma unroll
    for (int u = 0; u < NQC; u++) qc[u] = qa[hh * klr + (2 * u + a2) * 8 + l];
#pragma unroll
    for (int u = 0; u < NQR; u++) qr[u] = qp[hh * rd + (2 * u + a2) * 8 + l];
    float* out = a.sc_g + (size_t)(hg0 + hh) * max_seq + s0;
#pragma unroll 1
    for (int pass = 0; pass < KR_MLA_ROWS / (32 / HG); pass++) {
        const int r = prow + (32 / HG) * pass;
        if (s0 + r >= seq) break;
        const unsigned char* row = stage + r * pitch;
        float kc[NQC], kr[NQR];
#pragma unroll
        for (int u = 0; u < NQC; u++) kc[u] = kr_stage_val<FP8>(row, (2 * u + a2) * 8 + l);
#pragma unroll
        for (int u = 0; u < NQR; u++) kr[u] = kr_stage_val<FP8>(row + klr * esz, (2 * u + a2) * 8 + l);
        float acc = 0.0f;
#pragma unroll
        for (int u = 0; u < NQC; u++) acc = __builtin_fmaf(qc[u], kc[u], acc);
        float oth = __shfl_xor(acc, 8);
        float v = kr_mla_hsum8(a2 == 0 ? acc + oth : oth + acc);
        acc = 0.0f;
#pragma unroll
        for (int u = 0; u < NQR; u++) acc = __builtin_fmaf(qr[u], kr[u], acc);
        oth = __shfl_xor(acc, 8);
        v += kr_mla_hsum8(a2 == 0 ? acc + oth : oth + acc);
        v *= a.sm_scale;
        if (c16 == 0) out[r] = v;
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

// ---- KR_DECODE_FAST forms of launches 1 and 3 (decode only): the same products, tree sums ------------------------------------------------------------
// The exact prep launch gives an output ONE thread that walks nd = 128 weights with a stride of klr (the reference's chain), 128 waves in all for 4 MB of
// f32 weights per layer, and its latent RMSNorm is a 512-term sum on one lane; the exact w_vc launch reads 2-KB weight rows 64 bytes at a time with 16 lanes
// per output.  8.3 + 9.4 us per layer, more than the mode's expert launches.  Here: absorption -- a workgroup per (head, 64 outputs), its four waves take a
// quarter of the k range each (32 coalesced row reads in flight per lane), partial sums meet in LDS; the norm's sum of squares is a workgroup tree; w_vc --
// a wave per TWO output rows, a row is one 2-KB sweep of the wave (two 16-byte reads per lane), wave tree per row.
__global__ void __launch_bounds__(256) kr_mla_prep_fast_kernel(KrMlaArgs a) {
    __shared__ float sh[4][64];
    __shared__ float xs[640];
    const int pos = a.step->pos;
    const int tiles = a.klr / 64, nb_abs = a.nh * tiles, hd = a.nd + a.rd, half = a.rd / 2;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    if ((int)blockIdx.x < nb_abs) {
        const int h = blockIdx.x / tiles, jt = blockIdx.x % tiles, j = jt * 64 + lane;
        const float* qh = a.q_full + (size_t)h * hd;
        const int per = (a.nd + 3) / 4, i0 = wave * per, i1 = i0 + per < a.nd ? i0 + per : a.nd;
        const float* w = a.w_kc + (size_t)h * a.nd * a.klr + j;
        float o = 0.0f;
        for (int i = i0; i < i1; i += 32) {                  // 32 row reads in flight per lane: the whole slice of nd = 128
            float wv[32];
#pragma unroll
            for (int u = 0; u < 32; u++) wv[u] = __builtin_nontemporal_load(w + (size_t)(i + u < i1 ? i + u : i1 - 1) * a.klr);
#pragma unroll
            for (int u = 0; u < 32; u++) if (i + u < i1) o = __builtin_fmaf(qh[i + u], wv[u], o);
        }
        sh[wave][lane] = o;
        __syncthreads();
        if (wave == 0) a.q_abs[(size_t)h * a.klr + j] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
        if (jt == 0 && t < half) {   // decode.rs:3113-3128
            const float x1 = qh[a.nd + 2 * t], x2 = qh[a.nd + 2 * t + 1];
            const float c = a.rope_cos[(size_t)pos * half + t], s = a.rope_sin[(size_t)pos * half + t];
            a.q_pe[(size_t)h * a.rd + t] = x1 * c - x2 * s;
            a.q_pe[(size_t)h * a.rd + half + t] = x2 * c + x1 * s;
        }
        return;
    }
    // ---- compressed KV: RMSNorm (tree sum of squares), k_pe de-interleave + RoPE, cache append at `pos`
    float ss = 0.0f;
    for (int i = t; i < a.klr; i += 256) { const float v = a.kv_out[i]; xs[i] = v; ss += v * v; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) sh[0][wave] = ss;
    __syncthreads();
    const float rms = 1.0f / sqrtf(((sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3])) / (float)a.klr + a.eps);
    for (int i = t; i < a.klr; i += 256) {
        const float v = xs[i] * (rms * a.kv_a_norm[i]);
        if (a.kv_fp8) kr_mla_st<true>(a.ckv_cache, (size_t)pos * a.klr + i, v); else kr_mla_st<false>(a.ckv_cache, (size_t)pos * a.klr + i, v);
    }
    if (t < half) {
        const float x1 = a.kv_out[a.klr + 2 * t], x2 = a.kv_out[a.klr + 2 * t + 1];
        const float c = a.rope_cos[(size_t)pos * half + t], s = a.rope_sin[(size_t)pos * half + t];
        if (a.kv_fp8) { kr_mla_st<true>(a.kpe_cache, (size_t)pos * a.rd + t, x1 * c - x2 * s); kr_mla_st<true>(a.kpe_cache, (size_t)pos * a.rd + half + t, x2 * c + x1 * s); }
        else { kr_mla_st<false>(a.kpe_cache, (size_t)pos * a.rd + t, x1 * c - x2 * s); kr_mla_st<false>(a.kpe_cache, (size_t)pos * a.rd + half + t, x2 * c + x1 * s); }
    }
}
// v_projected[h][o] = w_vc[h][o][:] . attn_lat[h][:]     grid (vhd / 8, nh), 256 threads: wave w takes rows 2 w, 2 w + 1 of the workgroup's eight
__global__ void __launch_bounds__(256) kr_mla_wvc_fast_kernel(KrMlaArgs a) {
    const int h = blockIdx.y, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const float* lat = a.attn_lat + (size_t)h * a.klr;
    const int nchunk = a.klr / 4;                            // float4 chunks of a row
    float acc[2] = {0.0f, 0.0f};
    const int o0 = blockIdx.x * 8 + wave * 2;
    for (int c = lane; c < nchunk; c += 64) {
        const float4 x = *reinterpret_cast<const float4*>(lat + 4 * c);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int o = o0 + r < a.vhd ? o0 + r : a.vhd - 1;
            const float4 w = *reinterpret_cast<const float4*>(a.w_vc + ((size_t)h * a.vhd + o) * a.klr + 4 * c);
            acc[r] = __builtin_fmaf(w.x, x.x, __builtin_fmaf(w.y, x.y, __builtin_fmaf(w.z, x.z, __builtin_fmaf(w.w, x.w, acc[r]))));
        }
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        float v = acc[r];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0 && o0 + r < a.vhd) a.v_proj[(size_t)h * a.vhd + o0 + r] = v;
    }
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

// ---- decode, long caches (FP16 or E4M3): softmax + weighted sum of one head with PRODUCER and CONSUMER waves (the MLA twin of kr_gqa_pv_kernel) --------
// The weighted sum (mla_weighted_sum_fp16_avx2, decode.rs:4326) is one fma per position in position order for every latent element, and a
// wave issues roughly one instruction per 9 cycles whatever it does -- so the chain's waves should issue nothing but the chain.  Threads
// 0 .. klr-1 (consumers, thread j owns latent element j) read a COLUMN-major stage [j][64 positions] from LDS: one 16-byte read = 8 positions,
// probabilities four per broadcast read, the half -> float widening inside v_fma_mix_f32.  The last 256 threads (producers) fetch the next
// 64 latent rows a stage ahead, transpose 8 x 8 halves in registers (v_perm_b32) and write the other half of the double-buffered stage;
// they also scale the stage's 64 probabilities into a small window.  One workgroup barrier per 64 positions.  The score row stays in
// a.sc_g (written by kr_mla_scores_kernel): max, exp in place, position-ordered sum over 1024-value tiles with the running sum carried.
// Same operations in the same order as the softmax / weighted-sum half of kr_mla_attn_staged_kernel.
#define KR_MPV_ROWS 64
#define KR_MPV_EPT 1        // latent elements per consumer thread; 2 (half the probability reads, two chains per thread) measured 5 % slower
template <int NBC, bool FP8>
__global__ void __launch_bounds__(NBC * 8 / KR_MPV_EPT + 256) kr_mla_pv_kernel(KrMlaArgs a, int max_seq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char stage_mem[];
    __shared__ float red[16];
    __shared__ __attribute__((aligned(16))) float pw[2][KR_MPV_ROWS];
    __shared__ __attribute__((aligned(16))) float tile[1024 + 32];
    constexpr int klr = NBC * 8, esz = FP8 ? 1 : 2, pitchT = KR_MPV_ROWS * esz + 16, stage_bytes = klr * pitchT;
    constexpr int EPT = KR_MPV_EPT, NC = klr / EPT, NT = NC + 256, NW = NT / 64;     // NC consumer threads, thread j owns elements j + e * NC
    // producer pieces.  FP16: 8 rows x one 16-byte column (NBC columns x 8 row blocks).  E4M3: 16 rows x one 4-byte column (klr / 4 columns x 4
    // row blocks; 8 v_perm per four rows turn byte j of 16 rows into a 16-position group; position groups swizzled by (j / 8) % 4).
    constexpr int NCOL = FP8 ? klr / 4 : NBC, NRB = FP8 ? 4 : 8, RPB = KR_MPV_ROWS / NRB, CB = FP8 ? 4 : 16;
    constexpr int NBLK = NCOL * NRB / 256, RBS = 256 / NCOL, NRG = FP8 ? 4 : 8;       // blocks per producer thread, row-block step, 16-byte registers per block
    const int h = blockIdx.x, t = threadIdx.x, seq = a.step->pos + 1;
    const int nst = (seq + KR_MPV_ROWS - 1) / KR_MPV_ROWS;
    float* row = a.sc_g + (size_t)h * max_seq;
    const bool producer = t >= NC;
    const int pt = t - NC, col = pt & (NCOL - 1), rb0 = pt / NCOL;
    const __amdgpu_buffer_rsrc_t srd_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.ckv_cache), 0, seq * klr * esz, 0x00020000);
    const int voff = producer ? rb0 * RPB * klr * esz + col * CB : 0x7FFFFFF0;
    u32x4 rg[NBLK * NRG];        // one register set: the rows of stage k+2 are requested while stage k is consumed
    auto issue_v = [&](u32x4 (&R)[NBLK * NRG], int s0) {    // unguarded: rows at or past the current length are outside the descriptor and read as zero
#pragma unroll
        for (int b = 0; b < NBLK; b++) {
            if constexpr (FP8) {
#pragma unroll
                for (int i = 0; i < 16; i++) R[b * 4 + (i >> 2)][i & 3] = __builtin_amdgcn_raw_buffer_load_b32(srd_c, voff + (s0 + b * RBS * 16 + i) * klr, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) R[b * 8 + i] = __builtin_amdgcn_raw_buffer_load_b128(srd_c, voff + (s0 + b * RBS * 8 + i) * klr * 2, 0, 0);
            }
        }
    };
    auto commit_v = [&](const u32x4 (&R)[NBLK * NRG], int buf) {
#pragma unroll
        for (int b = 0; b < NBLK; b++) {
            const int rb = rb0 + b * RBS;                    // position group inside the stage
            if constexpr (FP8) {
                u32x4 o4[4];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const u32x4 q = R[b * 4 + m];
                    const uint32_t ta = __builtin_amdgcn_perm(q.y, q.x, 0x05010400u), tb = __builtin_amdgcn_perm(q.y, q.x, 0x07030602u);
                    const uint32_t tc = __builtin_amdgcn_perm(q.w, q.z, 0x05010400u), td = __builtin_amdgcn_perm(q.w, q.z, 0x07030602u);
                    o4[0][m] = __builtin_amdgcn_perm(tc, ta, 0x05040100u); o4[1][m] = __builtin_amdgcn_perm(tc, ta, 0x07060302u);
                    o4[2][m] = __builtin_amdgcn_perm(td, tb, 0x05040100u); o4[3][m] = __builtin_amdgcn_perm(td, tb, 0x07060302u);
                }
                unsigned char* base = stage_mem + buf * stage_bytes + (size_t)(col * 4) * pitchT + ((rb ^ ((col >> 1) & 3)) << 4);
#pragma unroll
                for (int j = 0; j < 4; j++) *reinterpret_cast<u32x4*>(base + j * pitchT) = o4[j];
            } else {
                unsigned char* base = stage_mem + buf * stage_bytes + (size_t)(col * 8) * pitchT + ((rb ^ (col & 7)) << 4);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
                    u32x4 o4;
                    o4.x = __builtin_amdgcn_perm(R[b * 8 + 1][j >> 1], R[b * 8 + 0][j >> 1], sel);
                    o4.y = __builtin_amdgcn_perm(R[b * 8 + 3][j >> 1], R[b * 8 + 2][j >> 1], sel);
                    o4.z = __builtin_amdgcn_perm(R[b * 8 + 5][j >> 1], R[b * 8 + 4][j >> 1], sel);
                    o4.w = __builtin_amdgcn_perm(R[b * 8 + 7][j >> 1], R[b * 8 + 6][j >> 1], sel);
                    *reinterpret_cast<u32x4*>(base + j * pitchT) = o4;
                }
            }
        }
    };
    if (producer) issue_v(rg, 0);
    // ---- softmax over the streamed score row
    float mx = -__builtin_inff();
    for (int s2 = t; s2 < seq; s2 += NT) mx = fmaxf(mx, row[s2]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < NW; w++) mx = fmaxf(mx, red[w]);
    for (int s2 = t; s2 < seq; s2 += NT) row[s2] = kr_expf(row[s2] - mx);
    if (producer) { commit_v(rg, 0); issue_v(rg, KR_MPV_ROWS); }
    __syncthreads();
    float se = 0.0f;
    for (int s0 = 0; s0 < seq; s0 += 1024) {
        const int n = min(1024, seq - s0), n32 = (n + 31) & ~31;
        for (int i = t; i < n32; i += NT) tile[i] = i < n ? row[s0 + i] : 0.0f;        // zero padding: s + 0.0f == s for sums of exponentials
        __syncthreads();
        if (t == 0) { se = kr_seq_sum(tile, n32, se); red[15] = se; }
        __syncthreads();
    }
    // workgroup-uniform: kept in a scalar register (as a lane value it was the one register the <64, FP16> form spilled at 3 waves per SIMD)
    float inv;
    // (the s_nop pair: gfx950 wants one wait state between a VALU write of a VGPR and a lane read of it, two between a VALU write of an SGPR and a VALU
    //  read -- the compiler's hazard pass does not look inside an asm statement)
    { const float inv_lane = 1.0f / red[15]; asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 1" : "=s"(inv) : "v"(inv_lane)); }
    if (t < KR_MPV_ROWS) pw[0][t] = t < seq ? row[t] * inv : 0.0f;                      // sc[s] *= inv, stage 0
    __syncthreads();
    // ---- weighted sum
    const unsigned char* rowT = stage_mem + (size_t)(t & (NC - 1)) * pitchT;
    const int swz = FP8 ? (t >> 3) & 3 : (t >> 3) & 7;
    float o[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) o[e] = 0.0f;
    auto lo = [](uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xFFFFu)); };
    auto hi = [](uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); };
    auto chain8 = [&](float acc, const u32x4 v, const float4 pa, const float4 pb) {
        acc = __builtin_fmaf(pa.x, lo(v.x), acc); acc = __builtin_fmaf(pa.y, hi(v.x), acc);
        acc = __builtin_fmaf(pa.z, lo(v.y), acc); acc = __builtin_fmaf(pa.w, hi(v.y), acc);
        acc = __builtin_fmaf(pb.x, lo(v.z), acc); acc = __builtin_fmaf(pb.y, hi(v.z), acc);
        acc = __builtin_fmaf(pb.z, lo(v.w), acc); acc = __builtin_fmaf(pb.w, hi(v.w), acc);
        return acc;
    };
    auto chain4f8 = [&](float acc, const uint32_t w, const float4 p4) {     // one dword = 4 positions, hardware E4M3 widening
        acc = __builtin_fmaf(p4.x, __builtin_amdgcn_cvt_f32_fp8((int)w, 0), acc); acc = __builtin_fmaf(p4.y, __builtin_amdgcn_cvt_f32_fp8((int)w, 1), acc);
        acc = __builtin_fmaf(p4.z, __builtin_amdgcn_cvt_f32_fp8((int)w, 2), acc); acc = __builtin_fmaf(p4.w, __builtin_amdgcn_cvt_f32_fp8((int)w, 3), acc);
        return acc;
    };
    for (int st0 = 0; st0 < nst; st0 += 2) {
#pragma unroll
      for (int u = 0; u < 2; u++) {                  // stage st0 + u: LDS buffer u
        const int st = st0 + u;
        if (st >= nst) break;
        const int s0 = st * KR_MPV_ROWS;
        if (producer) {
            if (st + 1 < nst) {
                commit_v(rg, u ^ 1);
                issue_v(rg, (st + 2) * KR_MPV_ROWS);
                if (pt < KR_MPV_ROWS) { const int sn = s0 + KR_MPV_ROWS + pt; pw[u ^ 1][pt] = sn < seq ? row[sn] * inv : 0.0f; }
            }
        } else {
            const float* P = pw[u];
            const unsigned char* rT = rowT + u * stage_bytes;
            const int n = min(KR_MPV_ROWS, seq - s0);
            if (FP8) {
                static_assert(!FP8 || EPT == 1, "E4M3 form: one element per consumer thread");
                if (n == KR_MPV_ROWS) {
#pragma unroll
                    for (int hf = 0; hf < 2; hf++) {
                        u32x4 v[2]; float4 pq[8];
#pragma unroll
                        for (int g = 0; g < 2; g++) v[g] = *reinterpret_cast<const u32x4*>(rT + (((hf * 2 + g) ^ swz) << 4));
#pragma unroll
                        for (int g = 0; g < 8; g++) pq[g] = *reinterpret_cast<const float4*>(P + hf * 32 + g * 4);
#pragma unroll
                        for (int g = 0; g < 8; g++) o[0] = chain4f8(o[0], v[g >> 2][g & 3], pq[g]);
                    }
                } else {
                    for (int k2 = 0; k2 < n; k2++) {
                        const uint32_t w = *reinterpret_cast<const uint32_t*>(rT + (((k2 >> 4) ^ swz) << 4) + ((k2 >> 2) & 3) * 4);
                        const int bs = k2 & 3;
                        const float vv = bs == 0 ? __builtin_amdgcn_cvt_f32_fp8((int)w, 0) : bs == 1 ? __builtin_amdgcn_cvt_f32_fp8((int)w, 1)
                                       : bs == 2 ? __builtin_amdgcn_cvt_f32_fp8((int)w, 2) : __builtin_amdgcn_cvt_f32_fp8((int)w, 3);
                        o[0] = __builtin_fmaf(P[k2], vv, o[0]);
                    }
                }
            } else if (n == KR_MPV_ROWS) {
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {     // 32 positions at a time: the values up front, the probabilities 16 positions ahead of their use
                    u32x4 v[EPT][4]; float4 pq[2][4];
#pragma unroll
                    for (int e = 0; e < EPT; e++)
#pragma unroll
                        for (int g = 0; g < 4; g++) v[e][g] = *reinterpret_cast<const u32x4*>(rT + (size_t)e * NC * pitchT + (((hf * 4 + g) ^ swz) << 4));
#pragma unroll
                    for (int q = 0; q < 2; q++)
#pragma unroll
                        for (int g = 0; g < 4; g++) pq[q][g] = *reinterpret_cast<const float4*>(P + hf * 32 + q * 16 + g * 4);
#pragma unroll
                    for (int g = 0; g < 4; g++)
#pragma unroll
                        for (int e = 0; e < EPT; e++) o[e] = chain8(o[e], v[e][g], pq[g >> 1][2 * (g & 1)], pq[g >> 1][2 * (g & 1) + 1]);
                }
            } else {                                 // last, partial stage (once per launch)
                const int nfull = n >> 3, rem = n & 7;
#pragma unroll
                for (int e = 0; e < EPT; e++) {
                    const unsigned char* rE = rT + (size_t)e * NC * pitchT;
                    for (int g0 = 0; g0 < nfull; g0++)
                        o[e] = chain8(o[e], *reinterpret_cast<const u32x4*>(rE + ((g0 ^ swz) << 4)), *reinterpret_cast<const float4*>(P + g0 * 8), *reinterpret_cast<const float4*>(P + g0 * 8 + 4));
                    if (rem) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(rE + ((nfull ^ swz) << 4));
                        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                        for (int k2 = 0; k2 < rem; k2++) {
                            const uint16_t hb = (uint16_t)((k2 & 1) ? (w[k2 >> 1] >> 16) : (w[k2 >> 1] & 0xFFFFu));
                            o[e] = __builtin_fmaf(P[nfull * 8 + k2], (float)__builtin_bit_cast(_Float16, hb), o[e]);
                        }
                    }
                }
            }
        }
        __syncthreads();
      }
    }
    if (!producer) {
#pragma unroll
        for (int e = 0; e < EPT; e++) a.attn_lat[(size_t)h * klr + e * NC + t] = o[e];
    }
}
template <int NBC, bool FP8> static size_t kr_mla_pv_lds() { return 2 * (size_t)NBC * 8 * (KR_MPV_ROWS * (FP8 ? 1 : 2) + 16); }

static size_t kr_mla_staged_lds(const KrMlaArgs& a, int lds_seq) {
    const size_t esz = a.kv_fp8 ? 1 : 2;
    return (size_t)(a.klr + a.rd) * 4 + ((((size_t)lds_seq + 40) * 4 + 15) & ~(size_t)15) + (size_t)KR_MLA_ROWS * ((size_t)(a.klr + a.rd) * esz + 16);
}
int kr_launch_mla_flash_decode(const KrMlaArgs& a, int max_seq, hipStream_t st);   // kr_mla_flash.hip
void kr_mla_flash_prepare();
template <bool FP8, int NBC>
static bool kr_launch_mla_staged(const KrMlaArgs& a, int max_seq, hipStream_t s, int n_tok) {
    const bool split = a.sc_g && n_tok == 1;          // decode with a long cache: head-shared scores launch, then softmax + weighted sum per head
    const size_t lds = kr_mla_staged_lds(a, max_seq);
    const bool prep = !s;                             // prepare-only call (outside graph capture): raise the windows of both forms
    if (!prep && !split && lds > 160 * 1024) return false;     // one-launch form: the score row is LDS-resident (the generic kernel reports the limit)
    const size_t esz = a.kv_fp8 ? 1 : 2;
    const size_t lds_sc = (size_t)KR_MLA_HG * (a.klr + a.rd) * 4 + (size_t)KR_MLA_ROWS * ((size_t)(a.klr + a.rd) * esz + 16);
    const size_t lds_pv = kr_mla_pv_lds<NBC, FP8>();
    // windows are per (kernel, device) and raised outside graph capture by kr_mla_attn_prepare; inside a capture the table already covers them
    if ((prep || !split) && lds <= 160 * 1024 && kr_lds_optin((const void*)kr_mla_attn_staged_kernel<FP8, NBC, 8>, lds)) return false;
    if (prep || split) {
        if (kr_lds_optin((const void*)kr_mla_scores_kernel<FP8, NBC, 8>, lds_sc)) return false;
        if (kr_lds_optin((const void*)kr_mla_pv_kernel<NBC, FP8>, lds_pv)) return false;
    }
    if (prep) { kr_fd_prepare<NBC * 8, 16>(); kr_mla_flash_prepare(); return true; }
    if (split && a.fast && kr_launch_mla_flash_decode(a, max_seq, s) == 0) return true;      // FAST: split-KV flash-decode on the f16 MFMA + merge
    if (split) {
        hipLaunchKernelGGL((kr_mla_scores_kernel<FP8, NBC, 8>), dim3((max_seq + KR_MLA_ROWS - 1) / KR_MLA_ROWS, (a.nh + KR_MLA_HG - 1) / KR_MLA_HG), dim3(512), lds_sc, s, a, max_seq);
        // the merge launch has one thread per chunk (1024): longer caches keep the exact softmax + weighted sum (silently dropping the tail chunks was ADVICE r2)
        if (a.fast && a.fd_o && a.fd_ml && a.nh <= 16 && (max_seq + KR_FD_CH - 1) / KR_FD_CH <= 1024) {     // tolerance mode: split-KV softmax + weighted sum over (chunk) workgroups, all heads share a latent row
            KrFdArgs f{};
            f.step = a.step; f.sc_g = a.sc_g; f.v_cache = a.ckv_cache; f.v_ld = a.klr; f.fd_o = a.fd_o; f.fd_ml = a.fd_ml; f.nh = a.nh; f.nkv = 1;
            f.gate = nullptr; f.gated = 0; f.out = a.attn_lat; f.img_out = nullptr;
            kr_launch_fd<NBC * 8, 16>(f, FP8 ? 1 : 0, max_seq, s);
        } else
        hipLaunchKernelGGL((kr_mla_pv_kernel<NBC, FP8>), dim3(a.nh), dim3(NBC * 8 / KR_MPV_EPT + 256), lds_pv, s, a, max_seq);
    } else hipLaunchKernelGGL((kr_mla_attn_staged_kernel<FP8, NBC, 8>), dim3(a.nh, n_tok), dim3(512), lds, s, a, max_seq);
    return true;
}
static bool kr_mla_staged(const KrMlaArgs& a, int max_seq, hipStream_t s, int n_tok) {
    if (a.rd != 64) return false;
    if (a.klr == 512) return a.kv_fp8 ? kr_launch_mla_staged<true, 64>(a, max_seq, s, n_tok) : kr_launch_mla_staged<false, 64>(a, max_seq, s, n_tok);
    if (a.klr == 256) return a.kv_fp8 ? kr_launch_mla_staged<true, 32>(a, max_seq, s, n_tok) : kr_launch_mla_staged<false, 32>(a, max_seq, s, n_tok);
    return false;
}
// raises the staged kernel's dynamic-LDS window; called outside graph capture (hipFuncSetAttribute is not a stream operation)
void kr_mla_attn_prepare(const KrMlaArgs& a, int max_seq) { (void)kr_mla_staged(a, max_seq, nullptr, 1); }
int kr_launch_mla_flash(const KrMlaArgs& a, int n_tok, hipStream_t st);   // kr_mla_flash.hip
static bool mla_no_mfma() { static const bool v = getenv("KR_EXACT_ATTN_VALU") != nullptr; return v; }     // A-B hook: keep the per-token exact launches
void kr_launch_mla(const KrMlaArgs& a_in, int max_seq, hipStream_t s, int n_tok) {
    KrMlaArgs a = a_in;
    // prompt pass: the w_kc absorption of the whole chunk on the f32 MFMA (one fma chain per output, bit-identical); the prep launch then only
    // ropes q_pe (one workgroup per head) and appends the latent / rope rows
    if (!a.step && n_tok >= 32 && kr_launch_mla_absorb_mfma(a.q_full, a.ld_q, a.nd + a.rd, a.nd, a.w_kc, a.klr, a.q_abs, n_tok, a.nh, s) == 0) a.absorb_done = 1;
    const int prep_blocks = a.nh * (a.absorb_done ? 1 : a.klr / 64) + 1;
    const bool dfast = a.step && a.decode_fast && n_tok == 1 && a.klr % 64 == 0 && a.klr <= 640 && a.klr % 4 == 0 && a.rd / 2 <= 256;      // tree-sum forms of the prep / w_vc launches
    if (dfast) hipLaunchKernelGGL(kr_mla_prep_fast_kernel, dim3(prep_blocks), dim3(256), 0, s, a);
    else if (a.kv_fp8) hipLaunchKernelGGL(kr_mla_prep_kernel<true>, dim3(prep_blocks, n_tok), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(kr_mla_prep_kernel<false>, dim3(prep_blocks, n_tok), dim3(64), 0, s, a);
    if (a.fast && !a.step && kr_launch_mla_flash(a, n_tok, s) == 0) {
        // prompt pass, tolerance mode: one flash-attention launch streams the latent cache once per 64 (token, head) rows
    } else if (!a.step && n_tok >= 32 && a.pf_sc && !mla_no_mfma() && kr_mla_exact_mfma_ok(a.nh, a.klr, a.rd)) {
        // prompt pass, exact mode: scores and the weighted sum of the latent rows on the f32 matrix cores, bit-identical to the per-token launches
        // (kr_attn_exact_mfma.hip: each kr_dot2acc is 16 accumulators with the AVX2 fold tree, the weighted sum one accumulator chain over positions);
        // the softmax pass is the GQA one (max, libm exp, sum in position order)
        const int rows = n_tok * a.nh;
        float* inv = a.pf_sc + (size_t)rows * a.pf_sc_ld; float* tmax = inv + rows;
        (void)kr_launch_mla_scores_mfma(a.q_abs, a.q_pe, a.ckv_cache, a.kpe_cache, a.kv_fp8, a.nh, a.klr, a.rd, a.pos0, n_tok, a.sm_scale, a.pf_sc, a.pf_sc_ld, tmax, s);
        kr_launch_pfm_softmax_rows(a.pf_sc, a.pf_sc_ld, inv, a.nh, a.pos0, rows, tmax, s);
        KrPfmGqaArgs g{};
        g.v_cache = a.ckv_cache; g.kv_fp8 = a.kv_fp8; g.nh = a.nh; g.nkv = 1; g.hd = a.klr; g.pos0 = a.pos0; g.gated = 0; g.attn_out = a.attn_lat;
        kr_launch_pfm_gqa_pv_mfma(g, n_tok, a.pf_sc, a.pf_sc_ld, inv, s);
    } else if (dfast && s && a.decode_fused && kr_launch_fmla(a, max_seq, s) == 0) {
        // KR_DECODE_FAST over a short cache: tree-sum scores / softmax / weighted sum (kr_decode_fast.hip)
    } else if (!kr_mla_staged(a, max_seq, s, n_tok)) {      // other geometries: the generic kernel
        if (a.kv_fp8) hipLaunchKernelGGL(kr_mla_attn_kernel<true>, dim3(a.nh, n_tok), dim3(512), (size_t)(a.klr + a.rd + max_seq + 8) * 4, s, a);
        else hipLaunchKernelGGL(kr_mla_attn_kernel<false>, dim3(a.nh, n_tok), dim3(512), (size_t)(a.klr + a.rd + max_seq + 8) * 4, s, a);
    }
    // prompt pass: the w_vc projection of the whole chunk on the f32 MFMA (its two-accumulator dot is the router's 16-chain structure:
    // kr_route_mfma.hip), bit-identical to the per-token launch below
    if (!a.step && n_tok >= 32 && kr_launch_mla_wvc_mfma(a.w_vc, a.attn_lat, a.v_proj, n_tok, a.nh, a.vhd, a.klr, s) == 0) return;
    if (dfast) { hipLaunchKernelGGL(kr_mla_wvc_fast_kernel, dim3((a.vhd + 7) / 8, a.nh), dim3(256), 0, s, a); return; }
    hipLaunchKernelGGL(kr_mla_wvc_kernel, dim3((a.vhd + 7) / 8, a.nh, n_tok), dim3(128), (size_t)a.klr * 4, s, a);
}
void kr_launch_rmsnorm_seq(float* x, const float* w, int n, float eps, hipStream_t s, int rows, int ld) {
    hipLaunchKernelGGL(kr_rmsnorm_seq_kernel, dim3(rows), dim3(256), (size_t)(n + 4) * 4, s, x, w, n, eps, ld);
}
