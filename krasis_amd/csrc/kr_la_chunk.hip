// kr_la_chunk.hip -- FAST-mode prompt pass of the gated delta rule (linear-attention layers): sub-chunks of 64 tokens in closed form.
//
// The reference recurrence (src/decode.rs:1293 restated per token in kr_pfm_la_recur_kernel, which stays the exact / default path) is
//     S <- a_t S;  d = b_t (v_t - S^T k_t);  S <- S + k_t d^T;  o_t = S^T q_t            (a_t = e^{g_t}, b_t = beta_t, S is [dk][dv])
// i.e. S_t = (I - b_t k_t k_t^T) a_t S_{t-1} + b_t k_t v_t^T: two dk-long dependent fma chains per token, 1.67 ms per 1024-token launch on
// MI355X (52.6 % of the FAST prompt pass, profiles/r02_prefill_fast_8192_kernel_stats.txt).  Over a sub-chunk of T = 64 tokens with
// G_t = sum_{r<=t} g_r the same map is (WY form):
//     (I + L) [W | Y] = [diag(b e^G) K | diag(b) V],   L[t][j] = b_t e^{G_t - G_j} (k_t . k_j)   (j < t)          -> forward substitution
//     U  = Y - W S_0                                                                                            (the 64 "delta" rows)
//     O  = (diag(e^G) Q - B W) S_0 + B Y,          B[t][j] = e^{G_t - G_j} (q_t . k_j)   (j <= t)
//     S' = e^{G_T} S_0 + (diag(e^{G_T - G}) K)^T U
// Only the two products with S_0 are sequential over sub-chunks.  Two launches:
//   kr_lac_prep_kernel  grid (sub-chunks, heads): everything that does not need the state -- K K^T, Q K^T on the f32 MFMA, the triangular
//                       solve in registers (one column of [W | Y] per thread, L rows broadcast from LDS), Q' = e^G Q - B W and O_0 = B Y.
//   kr_lac_scan_kernel  grid (heads x 4 column slices of the state): walks the sub-chunks; per step U = Y - W S, O = O_0 + Q' S (four 32x32
//                       blocks, one per wave) and S <- e^{G_T} S + K'^T U (four blocks), the next sub-chunk's tiles in flight in registers.
// Every product runs on the bf16 MFMA with its f32 operands split into two bf16 values (x = hi + lo, hi = bf16(x), lo = bf16(x - hi)) and
// acc += hi.hi + hi.lo + lo.hi: three v_mfma_f32_32x32x16_bf16 (96 matrix-pipe cycles per 16 k) instead of eight v_mfma_f32_32x32x2_f32 (512),
// products exact to ~2^-17 relative, f32 accumulation, f32 exponent range.  The prep kernel splits in registers (lc_blk_s); W, Q' and K^T leave it
// already split into planes (and K transposed), so the SCAN -- the sequential part -- stages them with plain 16-byte copies (lc_blk3); only the state
// slice and the delta rows are split inside the scan (16 values per lane and step).  Round 2 ran everything on v_mfma_f32_32x32x2_f32.
// (tolerance mode: tests/test_attn_fast_gpu.py states the bound.)  All exponents are differences G_t - G_j <= 0: nothing overflows, and a
// decay that underflows gives 0, not NaN (g is clamped at -80 per token).
#include <hip/hip_runtime.h>
#include "kr_lds_optin.h"
#include "kr_device.h"
#include "kr_libm.h"
#include "kr_prefill_ops.h"

#ifdef KR_TIMING   // tools/probes/lac_timing.hip: wall-clock stamps (10 ns units) by thread 0 of workgroup (0, 0); no-op in the product build
__device__ unsigned long long kr_lstamps[64];
#define LC_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) kr_lstamps[i] = wall_clock64(); } while (0)
#else
#define LC_STAMP(i) do { } while (0)
#endif
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
// e^x for x <= 0 on the transcendental unit (v_exp_f32 of x * log2 e, ~1 ulp of 2^-22 relative): the precise expf costs ~40 VALU instructions,
// 1.1 us per 16 values on a lone wave -- more than the epilogue it sits in
__device__ __forceinline__ float lc_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
typedef __bf16 lc_b8 __attribute__((ext_vector_type(8)));
// sum of squares of x[0..n) with 8 fma lanes (lane l owns elements 8 b + l, b ascending) folded by the reference's hsum tree: kr_pfm_sumsq8 of kr_prefill_ops.hip,
// callable by ANY aligned group of 8 lanes (the fold uses xor shuffles inside the group)
__device__ __forceinline__ float kr_lc_sumsq8(const float* x, int n, int l) {
    float acc = 0.0f;
    for (int b = 0; b < n / 8; b++) { const float v = x[b * 8 + l]; acc = __builtin_fmaf(v, v, acc); }
    acc = acc + __shfl_xor(acc, 4); acc = acc + __shfl_xor(acc, 1); acc = acc + __shfl_xor(acc, 2);
    return acc;
}
typedef uint32_t lc_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t lc_u2 __attribute__((ext_vector_type(2)));
// x = hi + lo in bf16 (round to nearest even, v_cvt_pk_bf16_f32): the pair carries x to ~2^-17 relative
__device__ __forceinline__ void lc_split(float x, uint16_t& hi, uint16_t& lo) {
    const __bf16 h = (__bf16)x;
    const __bf16 l = (__bf16)(x - (float)h);
    hi = __builtin_bit_cast(uint16_t, h); lo = __builtin_bit_cast(uint16_t, l);
}
#define LC_T 64
#define LC_D 128
#define LC_LD 132      // LDS row stride (floats) of the 64 x 128 tiles: rows 4 banks apart -> the float4 k-contiguous reads are conflict-free
#define LC_LS 68       // row stride of the 64 x 64 matrices
#define LC_SS 40       // row stride of the 32-column state / delta slices: rows k and k+4 are 32 banks apart
#define LC_PREP_LDS ((3 * LC_T * LC_LD + 2 * LC_T * LC_LS + 3 * LC_T + 2 * LC_T) * 4)      // K, Q, V tiles, L, B, G / b / b e^G, (fused) the 2 x 64 inverse norms
#define LC_PA 272      // bytes per row of a [rows][128 k] bf16 plane in LDS (256 + 16: the 16-byte fragment reads of 32 consecutive rows spread over the banks)
#define LC_PT 144      // bytes per row of a [rows][64 k] bf16 plane
#define LC_SCAN_LDS (4 * LC_T * LC_PA + 2 * LC_D * LC_PT + 2 * 32 * LC_PA + 2 * 32 * LC_PT + LC_T * 4)

struct KrLacArgs {
    const float *q, *k, *v, *gexp, *beta;   // [C][nv*128] x3, [C][nv] x2   (unused when `fused`)
    // fused != 0 (round 6): the prep launch forms q / k / v / e^g / beta itself from the in-projection's output -- causal conv over the carried conv slots + the
    // chunk's rows, SiLU, the two L2 norms, the gates: the arithmetic of kr_pfm_la_conv_kernel (kr_prefill_ops.hip), value for value -- so that launch, its
    // 540 MB of q / k / v / z stores per 8192 tokens and the prep's re-read of them disappear from the tolerance pass
    int fused; const float* qkvz; int ld_qkvz; const float* ba; int ld_ba; const float *conv_state, *conv_w, *a_log, *dt_bias; float scale; int nk, hr;
    float *Y, *G;                           // [n_sub*nv][64][128], [n_sub*nv][64]
    uint16_t *Wh, *Wl, *Qh, *Ql;            // bf16 planes [n_sub*nv][64][128] of W and Q' (hi, lo)
    uint16_t *Kh, *Kl;                      // bf16 planes [n_sub*nv][128][64] of K^T
    float *out, *state;                     // [C][nv*128]; [nv][128][128]
    int nv, C, n_sub;
};

// acc(32x32) += A(32 x K) . B(K x 32) on the bf16 MFMA with split operands (x = hi + lo in bf16: ~2^-17 relative per product, f32 accumulation, f32 range).
//   A_ROWS: A[m][k] = A[m*lda + k] (k contiguous) else A[m][k] = A[k*lda + m];   B_ROWS: B[k][n] = B[n*ldb + k] (k contiguous) else B[k*ldb + n]
struct LcNone { __device__ __forceinline__ void operator()(int) const {} };
// f32 operands in LDS, split ON THE FLY (the prep kernel has no LDS left for planes): lane (r, h) takes
// k = kb + 8 h .. + 8 of row / column r of both operands, splits the 16 values (3 VALU each) and issues hi.hi + hi.lo + lo.hi on the bf16 MFMA --
// 96 matrix-pipe cycles + ~200 VALU cycles per 16 k against 512 cycles of f32 MFMA.
template <bool A_ROWS, bool B_ROWS, int K>
__device__ __forceinline__ void lc_blk_s(v16f& acc, const float* A, int lda, const float* B, int ldb, int lane) {
    const int r = lane & 31, h = lane >> 5;
    float a[2][8], b[2][8];
    auto fetch = [&](int kb, float* a_, float* b_) {
        if (A_ROWS) {
            const f4 t0 = *reinterpret_cast<const f4*>(A + r * lda + kb + 8 * h), t1 = *reinterpret_cast<const f4*>(A + r * lda + kb + 8 * h + 4);
            a_[0] = t0.x; a_[1] = t0.y; a_[2] = t0.z; a_[3] = t0.w; a_[4] = t1.x; a_[5] = t1.y; a_[6] = t1.z; a_[7] = t1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) a_[j] = A[(kb + 8 * h + j) * lda + r];
        }
        if (B_ROWS) {
            const f4 t0 = *reinterpret_cast<const f4*>(B + r * ldb + kb + 8 * h), t1 = *reinterpret_cast<const f4*>(B + r * ldb + kb + 8 * h + 4);
            b_[0] = t0.x; b_[1] = t0.y; b_[2] = t0.z; b_[3] = t0.w; b_[4] = t1.x; b_[5] = t1.y; b_[6] = t1.z; b_[7] = t1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) b_[j] = B[(kb + 8 * h + j) * ldb + r];
        }
    };
    auto split8 = [&](const float* v, lc_b8& hi, lc_b8& lo) {
#pragma unroll
        for (int j = 0; j < 8; j++) { const __bf16 x = (__bf16)v[j]; hi[j] = x; lo[j] = (__bf16)(v[j] - (float)x); }
    };
    fetch(0, a[0], b[0]);
#pragma unroll
    for (int it = 0; it < K / 16; it++) {
        if (it + 1 < K / 16) fetch((it + 1) * 16, a[(it + 1) & 1], b[(it + 1) & 1]);
        lc_b8 ah, al, bh, bl;
        split8(a[it & 1], ah, al); split8(b[it & 1], bh, bl);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    }
}
__device__ __forceinline__ int lc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }   // accumulator register r -> block row

#define LC_PTH 512     // threads of a prep workgroup: its LDS (134 KB) allows one per CU, so the workgroup itself brings the second wave per SIMD that overlaps
                       // the split arithmetic of one wave with the LDS / MFMA latency of another (256 threads: 24.7 us per workgroup, 512: see lac_timing)
__global__ void __launch_bounds__(LC_PTH) kr_lac_prep_kernel(KrLacArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Kt = lds; float* Qt = Kt + LC_T * LC_LD; float* Vt = Qt + LC_T * LC_LD; float* Lm = Vt + LC_T * LC_LD; float* Bm = Lm + LC_T * LC_LS;
    float* Gs = Bm + LC_T * LC_LS; float* bs = Gs + LC_T; float* bg = bs + LC_T;
    const int sub = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = sub * LC_T, n = min(LC_T, a.C - c0);
    const size_t ld = (size_t)a.nv * LC_D, tile = ((size_t)sub * a.nv + h) * LC_T;
    LC_STAMP(0);
    float* nrm = bg + LC_T;       // [64][2] inverse L2 norms of the q / k rows (fused form)
    if (a.fused) {
        // ---- the tiles from the in-projection's output: thread = (4 consecutive tokens, 4 consecutive channels), three passes (q, k, v); the 4-tap window of the four
        // tokens is 7 rows of 16 bytes.  Tap j of token t is X(t - 3 + j): a chunk row for >= 0, carried conv slot 4 + i for i < 0 (kr_pfm_la_conv_kernel).
        const int kh = h / a.hr, rr = h % a.hr, gd = 2 * LC_D + 2 * LC_D * a.hr, key_dim = a.nk * LC_D;
        const int tg = tid >> 5, c = (tid & 31) * 4;
#pragma unroll 1
        for (int arr = 0; arr < 3; arr++) {
            const int ch = arr == 0 ? kh * LC_D + c : (arr == 1 ? key_dim + kh * LC_D + c : 2 * key_dim + h * LC_D + c);       // conv channel of column c
            const int off = arr == 0 ? c : (arr == 1 ? LC_D + c : 2 * LC_D + rr * LC_D + c);                                     // its column inside the key head's group
            const float* col = a.qkvz + (size_t)kh * gd + off;
            const float* cs = a.conv_state + (size_t)ch * 4;
            float4 cw[4];
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) cw[q4] = *reinterpret_cast<const float4*>(a.conv_w + (size_t)(ch + q4) * 4);
            float4 x[7];
#pragma unroll
            for (int j = 0; j < 7; j++) {
                const int i = c0 + 4 * tg - 3 + j;
                if (i < 0) x[j] = float4{cs[4 + i], cs[8 + i], cs[12 + i], cs[16 + i]};
                else x[j] = i < a.C ? *reinterpret_cast<const float4*>(col + (size_t)i * a.ld_qkvz) : float4{0.0f, 0.0f, 0.0f, 0.0f};
            }
            float* T = arr == 0 ? Qt : (arr == 1 ? Kt : Vt);
#pragma unroll
            for (int tt = 0; tt < 4; tt++) {
                const int t = 4 * tg + tt;
                float4 co = float4{0.0f, 0.0f, 0.0f, 0.0f};
                if (t < n) {
                    co.x = x[tt].x * cw[0].x + x[tt + 1].x * cw[0].y + x[tt + 2].x * cw[0].z + x[tt + 3].x * cw[0].w;
                    co.y = x[tt].y * cw[1].x + x[tt + 1].y * cw[1].y + x[tt + 2].y * cw[1].z + x[tt + 3].y * cw[1].w;
                    co.z = x[tt].z * cw[2].x + x[tt + 1].z * cw[2].y + x[tt + 2].z * cw[2].z + x[tt + 3].z * cw[2].w;
                    co.w = x[tt].w * cw[3].x + x[tt + 1].w * cw[3].y + x[tt + 2].w * cw[3].z + x[tt + 3].w * cw[3].w;
                    co.x = co.x * kr_sigmoid_poly5(co.x); co.y = co.y * kr_sigmoid_poly5(co.y); co.z = co.z * kr_sigmoid_poly5(co.z); co.w = co.w * kr_sigmoid_poly5(co.w);
                }
                *reinterpret_cast<float4*>(T + t * LC_LD + c) = co;
            }
        }
        if (wave == 0) {      // gates of the 64 tokens (decode.rs:3891-3901), then the running sum of g as below
            const int t = lane;
            float g = 0.0f, b = 0.0f;
            if (t < n) {
                const float* ba = a.ba + (size_t)(c0 + t) * a.ld_ba;
                const float b_raw = ba[kh * 2 * a.hr + rr], a_p = ba[kh * 2 * a.hr + a.hr + rr];
                b = 1.0f / (1.0f + kr_expf(-b_raw));
                const float ap_dt = a_p + a.dt_bias[h];
                const float softplus = ap_dt > 20.0f ? ap_dt : kr_logf(1.0f + kr_expf(ap_dt));
                const float ge = kr_expf(-(kr_expf(a.a_log[h])) * softplus);
                g = fmaxf(logf(fmaxf(ge, 1e-37f)), -80.0f);
            }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const float y = __shfl_up(g, o); if (lane >= o) g += y; }
            Gs[t] = g; bs[t] = b; bg[t] = b * lc_exp(g);
            a.G[tile + t] = g;
        }
        __syncthreads();
        // L2 norms of the 64 q rows and 64 k rows: 8 lanes per chain, the reference's order (kr_pfm_sumsq8); 128 chains on 512 threads in two rounds
        for (int u = tid; u < 2 * LC_T * 8; u += LC_PTH) {
            const int rowi = u >> 3, l = u & 7, t = rowi >> 1, which = rowi & 1;
            const float ss = kr_lc_sumsq8((which ? Kt : Qt) + t * LC_LD, LC_D, l);
            if (l == 0) nrm[t * 2 + which] = ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f;
        }
        __syncthreads();
        for (int u = tid; u < LC_T * 32; u += LC_PTH) {
            const int t = u >> 5, c4 = (u & 31) * 4;
            const float inv_q = nrm[t * 2] * a.scale, inv_k = nrm[t * 2 + 1] * 1.0f;
            float4 qv = *reinterpret_cast<float4*>(Qt + t * LC_LD + c4), kv = *reinterpret_cast<float4*>(Kt + t * LC_LD + c4);
            *reinterpret_cast<float4*>(Qt + t * LC_LD + c4) = float4{qv.x * inv_q, qv.y * inv_q, qv.z * inv_q, qv.w * inv_q};
            *reinterpret_cast<float4*>(Kt + t * LC_LD + c4) = float4{kv.x * inv_k, kv.y * inv_k, kv.z * inv_k, kv.w * inv_k};
        }
    } else {
    // ---- tiles (rows past the chunk end are zero: b = 0, g = 0 make them inert)
    for (int u = tid; u < LC_T * 32; u += LC_PTH) {
        const int t = u >> 5, c4 = (u & 31) * 4;
        float4 kk = make_float4(0, 0, 0, 0), qq = kk, vv = kk;
        if (t < n) {
            const size_t o = (size_t)(c0 + t) * ld + (size_t)h * LC_D + c4;
            kk = *reinterpret_cast<const float4*>(a.k + o); qq = *reinterpret_cast<const float4*>(a.q + o); vv = *reinterpret_cast<const float4*>(a.v + o);
        }
        *reinterpret_cast<float4*>(Kt + t * LC_LD + c4) = kk; *reinterpret_cast<float4*>(Qt + t * LC_LD + c4) = qq; *reinterpret_cast<float4*>(Vt + t * LC_LD + c4) = vv;
    }
    if (wave == 0) {
        const int t = lane;
        float g = t < n ? fmaxf(logf(fmaxf(a.gexp[(size_t)(c0 + t) * a.nv + h], 1e-37f)), -80.0f) : 0.0f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const float y = __shfl_up(g, o); if (lane >= o) g += y; }
        const float b = t < n ? a.beta[(size_t)(c0 + t) * a.nv + h] : 0.0f;
        Gs[t] = g; bs[t] = b; bg[t] = b * lc_exp(g);
        a.G[tile + t] = g;
    }
    }
    __syncthreads();
    LC_STAMP(1);
    // ---- K^T for the scan, split into bf16 planes [d][t] (the solve below overwrites the K tile with W): thread = (d, a quarter of the tokens)
    {
        const int d = tid & (LC_D - 1), tb = (tid >> 7) * 16;
        uint32_t hw[8], lw[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint16_t h0, l0, h1, l1;
            lc_split(Kt[(tb + 2 * i) * LC_LD + d], h0, l0); lc_split(Kt[(tb + 2 * i + 1) * LC_LD + d], h1, l1);
            hw[i] = (uint32_t)h0 | ((uint32_t)h1 << 16); lw[i] = (uint32_t)l0 | ((uint32_t)l1 << 16);
        }
        lc_u4* kh = reinterpret_cast<lc_u4*>(a.Kh + (tile * LC_D + (size_t)d * LC_T + tb));      // tile * LC_D = first element of this tile's [128][64] plane
        lc_u4* kl = reinterpret_cast<lc_u4*>(a.Kl + (tile * LC_D + (size_t)d * LC_T + tb));
#pragma unroll
        for (int i = 0; i < 2; i++) { kh[i] = lc_u4{hw[4 * i], hw[4 * i + 1], hw[4 * i + 2], hw[4 * i + 3]}; kl[i] = lc_u4{lw[4 * i], lw[4 * i + 1], lw[4 * i + 2], lw[4 * i + 3]}; }
    }
    // ---- K K^T -> L (strictly lower, decayed, row-scaled by b) and Q K^T -> B (lower incl. diagonal, decayed); the (0,1) blocks are zero
    if (wave < 6) {
        // one 32 x 32 block per wave: waves 0-2 L(0,0), L(1,1), L(1,0); waves 3-5 B(0,0), B(1,1), B(1,0)
        const bool isL = wave < 3;
        const int wb = wave % 3, rb = wb == 0 ? 0 : 1, cb = wb == 1 ? 1 : 0;
        v16f acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.0f;
        lc_blk_s<true, true, LC_D>(acc, (isL ? Kt : Qt) + rb * 32 * LC_LD, LC_LD, Kt + cb * 32 * LC_LD, LC_LD, lane);
        float* M = isL ? Lm : Bm;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = rb * 32 + lc_row(r, lane), col = cb * 32 + (lane & 31);
            const float d = lc_exp(Gs[row] - Gs[col]) * acc[r];
            M[row * LC_LS + col] = isL ? (col < row ? bs[row] * d : 0.0f) : (col <= row ? d : 0.0f);
        }
    } else {
        for (int u = tid - 6 * 64; u < 32 * 32; u += 128) { const int row = u >> 5, col = 32 + (u & 31); Lm[row * LC_LS + col] = 0.0f; Bm[row * LC_LS + col] = 0.0f; }
    }
    __syncthreads();
    LC_STAMP(2);
    // ---- (I + L) X = R,  R = [diag(b e^G) K | diag(b) V]  (X = [W | Y], 64 x 256), by 32-row blocks on the matrix cores:
    //          X_top = T11 R_top,   X_bot = T22 (R_bot - L21 X_top),   T_ii = (I + L_ii)^-1
    // Wave 0 inverts the two unit-lower-triangular diagonal blocks: lane (i, c) carries column c of T_ii through a 32-step forward substitution in
    // registers (rows of L_ii are broadcast reads) -- 496 fma per lane instead of the 2016 of the 64-step column solve this replaces, whose chains
    // were the longest phase of the kernel (8.5 of 28 us per workgroup) -- and parks T11 / T22 in the (0,1) blocks of Lm / Bm, which hold zeros nothing reads.
    // Waves 1-3 scale the K and V rows in place meanwhile (K is not needed unscaled any more: its products and the K^T planes are done).
    if (wave == 0) {
        const int blk = lane >> 5, c = lane & 31;
        const float* Lb = Lm + blk * 32 * LC_LS + blk * 32;
        float x[32];
#pragma unroll
        for (int t = 0; t < 32; t++) x[t] = 0.0f;
        x[0] = c == 0 ? 1.0f : 0.0f;
#pragma unroll
        for (int t = 1; t < 32; t++) {
            f2 sa = {0.0f, 0.0f}, sb = {0.0f, 0.0f};
#pragma unroll
            for (int j4 = 0; j4 < (t + 3) / 4; j4++) {        // entries at j >= t are stored zeros, x[j >= t] still 0
                const f4 l = *reinterpret_cast<const f4*>(Lb + t * LC_LS + 4 * j4);
                sa = __builtin_elementwise_fma((f2){l.x, l.y}, (f2){x[4 * j4], x[4 * j4 + 1]}, sa);
                sb = __builtin_elementwise_fma((f2){l.z, l.w}, (f2){x[4 * j4 + 2], x[4 * j4 + 3]}, sb);
            }
            x[t] = t == c ? 1.0f : -((sa.x + sa.y) + (sb.x + sb.y));
        }
        float* Td = (blk ? Bm : Lm) + 32 + c;
#pragma unroll
        for (int t = 0; t < 32; t++) Td[t * LC_LS] = x[t];
    } else {
        for (int u = tid - 64; u < LC_T * 64; u += LC_PTH - 64) {      // 16-byte units: row u >> 6; 32 units of K, 32 of V
            const int t = u >> 6, j = u & 63;
            float* p = j < 32 ? Kt + t * LC_LD + 4 * j : Vt + t * LC_LD + 4 * (j - 32);
            const float sc = j < 32 ? bg[t] : bs[t];
            f4 v = *reinterpret_cast<f4*>(p);
            v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
            *reinterpret_cast<f4*>(p) = v;
        }
    }
    __syncthreads();
    LC_STAMP(3);
    // wave w owns 32 of the 256 columns of [W | Y] through all three products: what it reads of R / X it wrote itself, so the hand-over between the
    // products is LDS order inside one wave (a wave's LDS operations execute in issue order), not a workgroup barrier
    {
        float* X = wave < 4 ? Kt + wave * 32 : Vt + (wave - 4) * 32;
        const float* T11 = Lm + 32; const float* T22 = Bm + 32; const float* L21 = Lm + 32 * LC_LS;
        const int n31 = lane & 31;
        v16f x0, x1;
#pragma unroll
        for (int i = 0; i < 16; i++) x0[i] = 0.0f;
        lc_blk_s<true, false, 32>(x0, T11, LC_LS, X, LC_LD, lane);
#pragma unroll
        for (int r = 0; r < 16; r++) X[lc_row(r, lane) * LC_LD + n31] = x0[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 16; i++) x1[i] = 0.0f;
        lc_blk_s<true, false, 32>(x1, L21, LC_LS, X, LC_LD, lane);
#pragma unroll
        for (int r = 0; r < 16; r++) { float* e = X + (32 + lc_row(r, lane)) * LC_LD + n31; *e = *e - x1[r]; }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 16; i++) x1[i] = 0.0f;
        lc_blk_s<true, false, 32>(x1, T22, LC_LS, X + 32 * LC_LD, LC_LD, lane);
#pragma unroll
        for (int r = 0; r < 16; r++) X[(32 + lc_row(r, lane)) * LC_LD + n31] = x1[r];
    }
    __syncthreads();
    LC_STAMP(4);
    // ---- Q' = diag(e^G) Q - B W (waves 0-3)  and  O_0 = B Y (waves 4-7): a wave takes 32 columns, both row halves (row half 0 only needs j < 32)
    {
        const bool isQ = wave < 4;
        const int col = (wave & 3) * 32 + (lane & 31);
        const float* Rt = (isQ ? Kt : Vt) + (wave & 3) * 32;
        v16f p0, p1;
#pragma unroll
        for (int i = 0; i < 16; i++) { p0[i] = 0.0f; p1[i] = 0.0f; }
        lc_blk_s<true, false, 32>(p0, Bm, LC_LS, Rt, LC_LD, lane);
        lc_blk_s<true, false, 64>(p1, Bm + 32 * LC_LS, LC_LS, Rt, LC_LD, lane);
        LC_STAMP(5);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int r0 = lc_row(r, lane), r1 = 32 + r0;
            if (isQ) {
                Qt[r0 * LC_LD + col] = lc_exp(Gs[r0]) * Qt[r0 * LC_LD + col] - p0[r];      // Q' over Q in LDS: element (row, col) is touched by this lane only
                Qt[r1 * LC_LD + col] = lc_exp(Gs[r1]) * Qt[r1 * LC_LD + col] - p1[r];
            } else {
                if (r0 < n) a.out[(size_t)(c0 + r0) * ld + (size_t)h * LC_D + col] = p0[r];
                if (r1 < n) a.out[(size_t)(c0 + r1) * ld + (size_t)h * LC_D + col] = p1[r];
            }
        }
        LC_STAMP(6);
    }
    __syncthreads();
    // ---- W (K tile), Q' (Q tile) -> their bf16 planes, Y (V tile) -> f32, all with 16-byte stores: chunk u = 8 consecutive columns of row u >> 4
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int u = tid + LC_PTH * i, t = u >> 4, c8 = (u & 15) * 8;
        const size_t o = (tile + t) * LC_D + c8;
        auto planes = [&](const float* src, uint16_t* ph, uint16_t* pl) {
            const f4 v0 = *reinterpret_cast<const f4*>(src + t * LC_LD + c8), v1 = *reinterpret_cast<const f4*>(src + t * LC_LD + c8 + 4);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            uint16_t hh[8], ll[8];
#pragma unroll
            for (int j = 0; j < 8; j++) lc_split(v[j], hh[j], ll[j]);
            *reinterpret_cast<lc_u4*>(ph + o) = lc_u4{(uint32_t)hh[0] | ((uint32_t)hh[1] << 16), (uint32_t)hh[2] | ((uint32_t)hh[3] << 16), (uint32_t)hh[4] | ((uint32_t)hh[5] << 16), (uint32_t)hh[6] | ((uint32_t)hh[7] << 16)};
            *reinterpret_cast<lc_u4*>(pl + o) = lc_u4{(uint32_t)ll[0] | ((uint32_t)ll[1] << 16), (uint32_t)ll[2] | ((uint32_t)ll[3] << 16), (uint32_t)ll[4] | ((uint32_t)ll[5] << 16), (uint32_t)ll[6] | ((uint32_t)ll[7] << 16)};
        };
        planes(Kt, a.Wh, a.Wl);
        planes(Qt, a.Qh, a.Ql);
        *reinterpret_cast<f4*>(a.Y + o) = *reinterpret_cast<const f4*>(Vt + t * LC_LD + c8);
        *reinterpret_cast<f4*>(a.Y + o + 4) = *reinterpret_cast<const f4*>(Vt + t * LC_LD + c8 + 4);
    }
}

// acc(32x32) += A(32 x K) . B(K x 32) with both operands given as bf16 (hi, lo) planes whose rows hold k contiguously: A row = output row, B row =
// output COLUMN (B is stored transposed).  Lane (r = lane & 31, h = lane >> 5) supplies k = 16 ks + 8 h .. + 8 of row r of both operands (one 16-byte
// read per plane); three MFMAs per 16 k: hi.hi + hi.lo + lo.hi.  `between(ks)` issues the caller's global prefetch under the MFMAs of step ks.
template <int K, typename F = LcNone>
__device__ __forceinline__ void lc_blk3(v16f& acc, const char* Ah, const char* Al, int lda, const char* Bh, const char* Bl, int ldb, int lane, F&& between = LcNone()) {
    const int r = lane & 31, h = lane >> 5;
    const char* ah = Ah + r * lda + h * 16; const char* al = Al + r * lda + h * 16;
    const char* bh = Bh + r * ldb + h * 16; const char* bl = Bl + r * ldb + h * 16;
    lc_b8 fa[2][2], fb[2][2];
    auto fetch = [&](int ks, int buf) {
        fa[buf][0] = *reinterpret_cast<const lc_b8*>(ah + ks * 32); fa[buf][1] = *reinterpret_cast<const lc_b8*>(al + ks * 32);
        fb[buf][0] = *reinterpret_cast<const lc_b8*>(bh + ks * 32); fb[buf][1] = *reinterpret_cast<const lc_b8*>(bl + ks * 32);
    };
    fetch(0, 0);
#pragma unroll
    for (int ks = 0; ks < K / 16; ks++) {
        if (ks + 1 < K / 16) fetch(ks + 1, (ks + 1) & 1);
        between(ks);
        const int b = ks & 1;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][0], fb[b][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][0], fb[b][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][1], fb[b][0], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// 16 accumulator values of a lane (rows lc_row(r), column lane & 31) -> the two bf16 planes of the TRANSPOSED matrix [col][row]: registers 4 q .. 4 q + 3
// are four consecutive rows -> one 8-byte store per plane and q
__device__ __forceinline__ void lc_store_t(const float (&v)[16], char* Ph, char* Pl, int ld, int row0, int lane) {
    const int n = lane & 31, h = lane >> 5;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint16_t hh[4], ll[4];
#pragma unroll
        for (int i = 0; i < 4; i++) lc_split(v[4 * q + i], hh[i], ll[i]);
        const int off = n * ld + (row0 + 8 * q + 4 * h) * 2;
        *reinterpret_cast<lc_u2*>(Ph + off) = lc_u2{(uint32_t)hh[0] | ((uint32_t)hh[1] << 16), (uint32_t)hh[2] | ((uint32_t)hh[3] << 16)};
        *reinterpret_cast<lc_u2*>(Pl + off) = lc_u2{(uint32_t)ll[0] | ((uint32_t)ll[1] << 16), (uint32_t)ll[2] | ((uint32_t)ll[3] << 16)};
    }
}

__global__ void __launch_bounds__(256) kr_lac_scan_kernel(KrLacArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lsm[];
    char* Whs = lsm; char* Wls = Whs + LC_T * LC_PA; char* Qhs = Wls + LC_T * LC_PA; char* Qls = Qhs + LC_T * LC_PA;      // [64 t][128 k]
    char* Khs = Qls + LC_T * LC_PA; char* Kls = Khs + LC_D * LC_PT;                                                       // K^T [128 d][64 t]
    char* Shs = Kls + LC_D * LC_PT; char* Sls = Shs + 32 * LC_PA;                                                         // S^T slice [32 cols][128 d]
    char* Uhs = Sls + 32 * LC_PA; char* Uls = Uhs + 32 * LC_PT;                                                           // U'^T [32 cols][64 t]
    float* Gs = reinterpret_cast<float*>(Uls + 32 * LC_PT);
    const int bid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the 4 column slices of one head read the same W / Q' / K tiles: keep them on one XCD (workgroup id % 8) so the re-reads hit its L2
    int h, slice;
    if (a.nv % 8 == 0) { h = (bid & 7) + 8 * (bid >> 5); slice = (bid >> 3) & 3; } else { h = bid >> 2; slice = bid & 3; }
    const size_t ld = (size_t)a.nv * LC_D;
    const int col = slice * 32 + (lane & 31);
    float* Sg = a.state + (size_t)h * LC_D * LC_D;
    float S[16];
#pragma unroll
    for (int r = 0; r < 16; r++) S[r] = Sg[(size_t)(wave * 32 + lc_row(r, lane)) * LC_D + col];
    lc_store_t(S, Shs, Sls, LC_PA, wave * 32, lane);
    // plane chunk u = tid + 256 i (i < 4) of 16 bytes: W / Q' planes [64][128] -> row u >> 4, chunk u & 15;  K^T planes [128][64] -> row u >> 3, chunk u & 7
    const size_t tile_elems = (size_t)LC_T * LC_D, tile_stride = (size_t)a.nv * tile_elems;      // elements per tile / between sub-chunks
    const lc_u4* pWh = reinterpret_cast<const lc_u4*>(a.Wh + (size_t)h * tile_elems) + tid; const lc_u4* pWl = reinterpret_cast<const lc_u4*>(a.Wl + (size_t)h * tile_elems) + tid;
    const lc_u4* pQh = reinterpret_cast<const lc_u4*>(a.Qh + (size_t)h * tile_elems) + tid; const lc_u4* pQl = reinterpret_cast<const lc_u4*>(a.Ql + (size_t)h * tile_elems) + tid;
    const lc_u4* pKh = reinterpret_cast<const lc_u4*>(a.Kh + (size_t)h * tile_elems) + tid; const lc_u4* pKl = reinterpret_cast<const lc_u4*>(a.Kl + (size_t)h * tile_elems) + tid;
    lc_u4 pf[6][4]; float pg = 0.0f;
#define LC_FETCH3(SUB, P, I) { const size_t o_ = (size_t)(SUB) * (tile_stride / 8) + 256 * (I); \
        pf[P][I] = (P) == 0 ? pWh[o_] : (P) == 1 ? pWl[o_] : (P) == 2 ? pQh[o_] : (P) == 3 ? pQl[o_] : (P) == 4 ? pKh[o_] : pKl[o_]; }
#pragma unroll
    for (int p = 0; p < 6; p++)
#pragma unroll
        for (int i = 0; i < 4; i++) LC_FETCH3(0, p, i)
    if (tid < LC_T) pg = a.G[(size_t)h * LC_T + tid];
    // the addend of the first product (Y rows for waves 0,1; O_0 rows, written by the prep kernel, for waves 2,3) is requested ONE STEP AHEAD as well:
    // with the products at bf16 rate a step is shorter than a memory round trip
    const int rbw = wave & 1;
    float yn[16];
    auto fetch_y = [&](int sb) {
        const size_t tl = ((size_t)sb * a.nv + h) * LC_T;
        const int cc = sb * LC_T, nn = min(LC_T, a.C - cc);
        if (wave < 2) {
#pragma unroll
            for (int r = 0; r < 16; r++) yn[r] = a.Y[(tl + rbw * 32 + lc_row(r, lane)) * LC_D + col];
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) { const int row = min(rbw * 32 + lc_row(r, lane), nn - 1); yn[r] = a.out[(size_t)(cc + row) * ld + (size_t)h * LC_D + col]; }
        }
    };
    fetch_y(0);
    for (int sub = 0; sub < a.n_sub; sub++) {
        const int c0 = sub * LC_T, n = min(LC_T, a.C - c0);
        const int nxt = min(sub + 1, a.n_sub - 1);     // the last step re-requests its own tiles: no branch in the product loop
        if (sub == 1) LC_STAMP(10);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int u = tid + 256 * i, ro = (u >> 4) * LC_PA + (u & 15) * 16, ko = (u >> 3) * LC_PT + (u & 7) * 16;
            *reinterpret_cast<lc_u4*>(Whs + ro) = pf[0][i]; *reinterpret_cast<lc_u4*>(Wls + ro) = pf[1][i];
            *reinterpret_cast<lc_u4*>(Qhs + ro) = pf[2][i]; *reinterpret_cast<lc_u4*>(Qls + ro) = pf[3][i];
            *reinterpret_cast<lc_u4*>(Khs + ko) = pf[4][i]; *reinterpret_cast<lc_u4*>(Kls + ko) = pf[5][i];
        }
        if (tid < LC_T) Gs[tid] = pg;
        __syncthreads();
        if (sub == 1) LC_STAMP(11);
        // ---- waves 0,1: U = Y - W S (row halves);  waves 2,3: O = O_0 + Q' S.  The Y / O_0 values are requested FIRST (the memory counter is in
        // order: waiting for them must not wait for the prefetch), then the next sub-chunk's planes are requested from inside the product loop
        {
            const int rb = rbw;
            float y[16];
#pragma unroll
            for (int r = 0; r < 16; r++) y[r] = yn[r];
            fetch_y(nxt);
            if (tid < LC_T) pg = a.G[((size_t)nxt * a.nv + h) * LC_T + tid];
            v16f acc;
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = 0.0f;
            if (sub == 1) LC_STAMP(12);
            lc_blk3<LC_D>(acc, (wave < 2 ? Whs : Qhs) + rb * 32 * LC_PA, (wave < 2 ? Wls : Qls) + rb * 32 * LC_PA, LC_PA, Shs, Sls, LC_PA, lane, [&](int ks) {
                // 24 requests over 8 steps: plane ks >> 1 ... two planes per pair of steps, in plane order
                if (ks < 6) {
#pragma unroll
                    for (int i = 0; i < 4; i++) LC_FETCH3(nxt, ks, i)
                }
            });
            if (sub == 1) LC_STAMP(13);
            const float gT = Gs[LC_T - 1];
            if (wave < 2) {
                float u[16];
#pragma unroll
                for (int r = 0; r < 16; r++) { const int row = rb * 32 + lc_row(r, lane); u[r] = (y[r] - acc[r]) * lc_exp(gT - Gs[row]); }
                lc_store_t(u, Uhs, Uls, LC_PT, rb * 32, lane);
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++) { const int row = rb * 32 + lc_row(r, lane); if (row < n) a.out[(size_t)(c0 + row) * ld + (size_t)h * LC_D + col] = y[r] + acc[r]; }
            }
        }
        if (sub == 1) LC_STAMP(14);
        __syncthreads();
        if (sub == 1) LC_STAMP(15);
        // ---- S <- e^{G_T} S + K^T U'   (U' rows already carry e^{G_T - G_t}); wave w owns state rows [32w, 32w+32)
        {
            const float gam = lc_exp(Gs[LC_T - 1]);
            v16f sa;
#pragma unroll
            for (int r = 0; r < 16; r++) sa[r] = S[r] * gam;
            lc_blk3<LC_T>(sa, Khs + wave * 32 * LC_PT, Kls + wave * 32 * LC_PT, LC_PT, Uhs, Uls, LC_PT, lane);
#pragma unroll
            for (int r = 0; r < 16; r++) S[r] = sa[r];
        }
        if (sub == 1) LC_STAMP(16);
        __syncthreads();              // every wave has read S^T (first product of this step) and U'^T before they are rewritten
        if (sub == 1) LC_STAMP(17);
        lc_store_t(S, Shs, Sls, LC_PA, wave * 32, lane);
        if (sub == 1) LC_STAMP(18);
        if (sub == 2) LC_STAMP(19);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) Sg[(size_t)(wave * 32 + lc_row(r, lane)) * LC_D + col] = S[r];
}

// floats of scratch per padded token (a multiple of 64 tokens) and head: Y rows, the bf16 plane pairs of W, Q', K^T (2 + 2 bytes per value each), one G
size_t kr_pfm_la_chunk_scratch_floats(int C, int nv) { const size_t c64 = ((size_t)C + LC_T - 1) / LC_T * LC_T; return c64 * nv * (4 * LC_D + 1) + 4; }   // Y (f32), three pairs of bf16 planes, G
bool kr_pfm_la_chunk_ok(int dk, int dv, int C) { return dk == LC_D && dv == LC_D && C >= LC_T; }

// > 64 KB of dynamic LDS is an opt-in per device (and not allowed inside a stream capture: the prompt pass is never captured)
static int lac_prepare() {
    return kr_lds_optin(reinterpret_cast<const void*>(kr_lac_prep_kernel), LC_PREP_LDS) || kr_lds_optin(reinterpret_cast<const void*>(kr_lac_scan_kernel), LC_SCAN_LDS);
}

// fused != 0: the prep launch forms q / k / v and the gates from the in-projection's output itself (no conv launch ran); `between` (may be null) is called between the
// two launches, on the stream -- the carried conv slots are advanced there, AFTER the prep launch has read the old ones
int kr_launch_pfm_la_chunked(const KrPfmLaArgs& p, float* state, float* out, float* scratch, int C, hipStream_t st, const KrPfSync* sy, int fused, void (*between)(const KrPfmLaArgs&, int, hipStream_t, const KrPfSync*)) {
    if (!kr_pfm_la_chunk_ok(p.dk, p.dv, C) || !scratch) return 1;
    if (fused && (p.nv != p.nk * p.hr || p.ld_qkvz % 4)) return 1;
    if (lac_prepare()) return 1;
    KrLacArgs a{};
    a.fused = fused; a.qkvz = p.qkvz; a.ld_qkvz = p.ld_qkvz; a.ba = p.ba; a.ld_ba = p.ld_ba; a.conv_state = p.conv_state; a.conv_w = p.conv_w; a.a_log = p.a_log; a.dt_bias = p.dt_bias;
    a.scale = p.scale; a.nk = p.nk; a.hr = p.hr;
    a.q = p.q; a.k = p.k; a.v = p.v; a.gexp = p.gexp; a.beta = p.beta; a.nv = p.nv; a.C = C; a.n_sub = (C + LC_T - 1) / LC_T;
    const size_t tiles = (size_t)a.n_sub * p.nv * LC_T;
    a.Y = scratch; a.G = a.Y + tiles * LC_D;
    uint16_t* pl = reinterpret_cast<uint16_t*>(a.G + ((tiles + 3) / 4) * 4);        // planes start 16-byte aligned
    a.Wh = pl; a.Wl = a.Wh + tiles * LC_D; a.Qh = a.Wl + tiles * LC_D; a.Ql = a.Qh + tiles * LC_D; a.Kh = a.Ql + tiles * LC_D; a.Kl = a.Kh + tiles * LC_D;
    a.out = out; a.state = state;
    hipLaunchKernelGGL(kr_lac_prep_kernel, dim3(a.n_sub, p.nv), dim3(LC_PTH), LC_PREP_LDS, st, a);
    if (between) between(p, C, st, sy);
    if (sy) kr_pf_wait(st, sy->wait_b);      // only the scan reads the state the previous chunk of the prompt leaves
    hipLaunchKernelGGL(kr_lac_scan_kernel, dim3(p.nv * 4), dim3(256), LC_SCAN_LDS, st, a);
    return 0;
}
