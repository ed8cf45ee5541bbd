// kr_attn_exact_mfma.hip -- passes A (scores) and C (P.V) of the EXACT prompt-pass attention (decode.rs:4194-4281 order) on the f32 matrix
// cores, BIT-IDENTICAL to the vector-ALU kernels kr_pfm_gqa_scores*_kernel / kr_pfm_gqa_pv_kernel of kr_prefill_ops.hip (which stay as the
// fallback for group sizes that do not divide 32, and are what these were checked against: every prompt-pass == decode test runs them).
//
// v_mfma_f32_32x32x2_f32 computes D = fma(A[.][1], B[1][.], fma(A[.][0], B[0][.], C)): two fused multiply-adds in k order, no intermediate
// rounding beyond the fma's own -- a sequential f32 fma chain IS an accumulator fed with consecutive pairs (the router logits and the MLA
// projections of this round use the same fact, kr_route_mfma.hip).
//   * scores: the reference keeps 8 fma lanes per (query, position): lane l walks q[8b + l] * k[8b + l] over b ascending, then
//     ((l0 + l4) + (l1 + l5)) + ((l2 + l6) + (l3 + l7)), then * sm_scale.  Lane l of a 32-query x 32-position block is accumulator l fed with
//     the pairs (b, b + 1): A[q][k] = q[8 (2m + k) + l], B[k][p] = K[p][8 (2m + k) + l] -- 8 accumulators per block, hd / 16 MFMAs each.
//   * P.V: out[q][d] = chain over positions ascending of fma(p[q][pos], V[pos][d]): ONE accumulator per 32 x 32 (query, dim) block with the
//     positions as k.  A position a query must not see enters with p = 0: fma(0, v, acc) = acc exactly (v is finite: cache rows past the
//     last position of the tile are never read).
// Both run at the f32-MFMA rate, which is the vector fma rate on CDNA4 -- the gain is structural: operands staged once in LDS for 32 x 32
// outputs, no per-lane broadcast traffic (the vector P.V pass was bound by the LDS return path, the scores pass by 4 LDS reads per 4 fma).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "kr_device.h"
#include "kr_lds_optin.h"
#include "kr_libm.h"
#include "kr_prefill_ops.h"

typedef float xm_v16f __attribute__((ext_vector_type(16)));
typedef float xm_f4 __attribute__((ext_vector_type(4)));

#define XM_LDF 68          // floats per 64-float LDS row (272 B: consecutive rows 4 banks apart)

// ------------------------------------------------------------------------------------------------------------------------------------
// pass A.  grid (position tiles of 64, row tiles of 64 queries, nkv), 4 waves = 2 x 2 blocks of 32 queries x 32 positions.
// A query row r of the tile is (token t0 + r / group, head kvh * group + r % group), t0 = tile * (64 / group).
// ------------------------------------------------------------------------------------------------------------------------------------
template <bool FP8>
__global__ void __launch_bounds__(256, 2) kr_pfm_gqa_scores_mfma_kernel(const KrPfmGqaArgs a, float* __restrict__ sc, int sc_ld, int C, float* __restrict__ tmax) {
    __shared__ __attribute__((aligned(16))) float Qs[64 * XM_LDF];
    __shared__ __attribute__((aligned(16))) float Ks[64 * XM_LDF];
    // group divides 32: a power of two -- tokens and heads of a row by shift and mask (a run-time integer division is ~50 instructions, and a lone wave
    // issues one per ~8 cycles: 16 rows x 2 divisions per lane were a third of a workgroup's life)
    const int hd = a.hd, group = a.nh / a.nkv, lg = __builtin_ctz(group), gm = group - 1, kvh = blockIdx.z, TT = 64 >> lg, t0 = blockIdx.y * TT, kvs = a.nkv * hd;
    const int tn = min(TT, C - t0), R = group * tn;
    const int p_lo = blockIdx.x * 64, p_max = a.pos0 + t0 + tn - 1;          // last position any query of the tile may see
    if (p_lo > p_max) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r31 = lane & 31, kh = lane >> 5;
    const int rb = (wave >> 1) * 32, pb = (wave & 1) * 32;                    // this wave's block inside the tile
    xm_v16f acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[j][i] = 0.0f;
    // staging: q 64 rows x 64 floats per stage (16 float4 per row, 4 per thread), K 64 positions x 64 values (positions past p_max re-read p_max:
    // their scores are never stored).  The next stage's requests are in flight during the MFMAs of the current one.
    const float* qsrc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
        const int rr = row < R ? row : R - 1, tt = rr >> lg, hh = kvh * group + (rr & gm);
        qsrc[i] = a.q_out + (size_t)(t0 + tt) * a.nh * hd + (size_t)hh * hd + c4;
    }
    constexpr int KCH = FP8 ? 1 : 2;
    const char* ksrc[KCH];
#pragma unroll
    for (int i = 0; i < KCH; i++) {
        const int u = tid + 256 * i, prow = FP8 ? (u >> 2) : (u >> 3), off = FP8 ? (u & 3) * 16 : (u & 7) * 16;       // bytes inside the 64-value segment
        ksrc[i] = reinterpret_cast<const char*>(a.k_cache) + ((size_t)min(p_lo + prow, p_max) * kvs + (size_t)kvh * hd) * (FP8 ? 1 : 2) + off;
    }
    xm_f4 qreg[4]; u32x4 kreg[KCH];
    auto load_stage = [&](int d0) {
#pragma unroll
        for (int i = 0; i < 4; i++) qreg[i] = *reinterpret_cast<const xm_f4*>(qsrc[i] + d0);
#pragma unroll
        for (int i = 0; i < KCH; i++) kreg[i] = *reinterpret_cast<const u32x4*>(ksrc[i] + (size_t)d0 * (FP8 ? 1 : 2));
    };
    auto commit_stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; i++) { const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4; *reinterpret_cast<xm_f4*>(Qs + row * XM_LDF + c4) = qreg[i]; }
        if (FP8) {
            const uint32_t ww[4] = {kreg[0].x, kreg[0].y, kreg[0].z, kreg[0].w};
            float* dst = Ks + (tid >> 2) * XM_LDF + (tid & 3) * 16;
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++)
                *reinterpret_cast<xm_f4*>(dst + 4 * q4) = xm_f4{kr_e4m3_to_f32((uint8_t)ww[q4]), kr_e4m3_to_f32((uint8_t)(ww[q4] >> 8)),
                                                              kr_e4m3_to_f32((uint8_t)(ww[q4] >> 16)), kr_e4m3_to_f32((uint8_t)(ww[q4] >> 24))};
        } else {
#pragma unroll
            for (int i = 0; i < KCH; i++) {
                const int u = tid + 256 * i, prow = u >> 3, c8 = (u & 7) * 8;
                const uint32_t ww[4] = {kreg[i].x, kreg[i].y, kreg[i].z, kreg[i].w};
                float f[8];
#pragma unroll
                for (int j = 0; j < 4; j++) { f[2 * j] = __half2float(__ushort_as_half((uint16_t)(ww[j] & 0xFFFFu))); f[2 * j + 1] = __half2float(__ushort_as_half((uint16_t)(ww[j] >> 16))); }
                *reinterpret_cast<xm_f4*>(Ks + prow * XM_LDF + c8) = xm_f4{f[0], f[1], f[2], f[3]};
                *reinterpret_cast<xm_f4*>(Ks + prow * XM_LDF + c8 + 4) = xm_f4{f[4], f[5], f[6], f[7]};
            }
        }
    };
    load_stage(0);
    for (int d0 = 0; d0 < hd; d0 += 64) {
        __syncthreads();                                                      // the previous stage's readers are done
        commit_stage();
        if (d0 + 64 < hd) load_stage(d0 + 64);
        __syncthreads();
        const float* qp = Qs + (rb + r31) * XM_LDF + 8 * kh;
        const float* kp = Ks + (pb + r31) * XM_LDF + 8 * kh;
#pragma unroll
        for (int m = 0; m < 4; m++) {                                         // 16 consecutive d: the (b, b + 1) pair of every lane chain
            const xm_f4 q0 = *reinterpret_cast<const xm_f4*>(qp + 16 * m), q1 = *reinterpret_cast<const xm_f4*>(qp + 16 * m + 4);
            const xm_f4 k0 = *reinterpret_cast<const xm_f4*>(kp + 16 * m), k1 = *reinterpret_cast<const xm_f4*>(kp + 16 * m + 4);
            const float qa[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w}, ka[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[j], ka[j], acc[j], 0, 0, 0);
        }
    }
    // hsum8 of the reference (kr_pfm_hsum8: xor 4, xor 1, xor 2), * sm_scale, causal store; the maximum of every row over this block's 32
    // positions goes to tmax[row][32-position block] so that pass B finds the row maximum without another walk over the score scratch (the
    // scratch is the traffic of the exact attention: a 20 k-token context is 1.3 GB of scores per chunk and layer)
    const int pos = p_lo + pb + r31;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int r = rb + (i & 3) + 8 * (i >> 2) + 4 * kh;
        const float sv = ((acc[0][i] + acc[4][i]) + (acc[1][i] + acc[5][i])) + ((acc[2][i] + acc[6][i]) + (acc[3][i] + acc[7][i]));
        float mv = -__builtin_inff();
        size_t rowi = 0;
        if (r < R) {
            const int tt = r >> lg, hh = kvh * group + (r & gm);
            rowi = (size_t)(t0 + tt) * a.nh + hh;
            if (pos <= a.pos0 + t0 + tt) { mv = sv * a.sm_scale; sc[rowi * sc_ld + pos] = mv; }
        }
        if (tmax) {        // maximum over the 32 lanes of this lane half (the other half holds other rows)
            mv = kr_red16_max_f32(mv);                            // 4 DPP steps inside each 16-lane row, then one exchange between the two rows of the lane half
            mv = fmaxf(mv, __shfl_xor(mv, 16));
            if (r31 == 0 && r < R) tmax[rowi * (size_t)(sc_ld >> 5) + ((p_lo + pb) >> 5)] = mv;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// pass C.  grid (row tiles of 32 queries, nkv), 4 waves; wave w owns the 32-dim blocks w * NB .. w * NB + NB - 1 of the head (NB = hd / 128,
// head_dim 64: waves 0 and 1 one block each).  64 positions per stage: probabilities (masked) as f32 and the raw V rows in LDS.
// ------------------------------------------------------------------------------------------------------------------------------------
template <bool FP8, int HD, int PS>      // PS = positions per stage (64; 32 for the 512-wide latent rows of MLA: the V tile must fit the static LDS window)
__global__ void __launch_bounds__(256) kr_pfm_gqa_pv_mfma_kernel(const KrPfmGqaArgs a, const float* __restrict__ sc, int sc_ld, int C, const float* __restrict__ inv) {
    constexpr int ESZ = FP8 ? 1 : 2, VROW = HD * ESZ + 64;                    // bytes per V row in LDS (+ 64: the two k halves land 16 banks apart)
    constexpr int NB = HD >= 128 ? HD / 128 : 1, NW = HD >= 128 ? 4 : HD / 32;   // blocks per wave, waves that own blocks
    constexpr int VCH = PS * HD * ESZ / 16 / 256;                             // 16-byte V chunks per thread per stage
    constexpr int PLD = PS + 4, PPT = PS / 8;                                 // floats per P row in LDS; probabilities per thread (8 threads per row)
    static_assert(VCH >= 1 && (PPT == 4 || PPT == 8), "stage shape");
    __shared__ __attribute__((aligned(16))) float Ps[32 * PLD];
    __shared__ __attribute__((aligned(16))) char Vs[PS * VROW];
    const int group = a.nh / a.nkv, lg = __builtin_ctz(group), gm = group - 1, kvh = blockIdx.y, TT = 32 >> lg, t0 = blockIdx.x * TT, kvs = a.nkv * HD;      // group: a power of two
    const int tn = min(TT, C - t0), R = group * tn, p_max = a.pos0 + t0 + tn - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r31 = lane & 31, kh = lane >> 5;
    xm_v16f acc[NB];
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[b][i] = 0.0f;
    // staging maps: P row = tid >> 3 (32 rows), PPT positions per thread; V chunk u = tid + 256 i: row u / CPR, 16-byte chunk u % CPR
    const int prow = tid >> 3, pseg = (tid & 7) * PPT;
    const int ptt = (prow < R ? prow : 0) >> lg, pg = (prow < R ? prow : 0) & gm, qpos = a.pos0 + t0 + ptt;
    const float* prow_p = sc + ((size_t)(t0 + ptt) * a.nh + (size_t)kvh * group + pg) * sc_ld;
    // pass B left the exponentials unscaled (inv != nullptr): p = e * (1 / sum) is formed here, the multiply the reference does in place (decode.rs:4260)
    const float iv = inv ? inv[(size_t)(t0 + ptt) * a.nh + (size_t)kvh * group + pg] : 1.0f;
    constexpr int CPR = HD * ESZ / 16;                                        // chunks per V row
    xm_f4 pp[PPT / 4]; u32x4 pv[VCH];
    auto load_stage = [&](int p0) {
#pragma unroll
        for (int q = 0; q < PPT / 4; q++) pp[q] = *reinterpret_cast<const xm_f4*>(prow_p + p0 + pseg + 4 * q);
#pragma unroll
        for (int i = 0; i < VCH; i++) {
            const int u = tid + 256 * i, vr = u / CPR, vc = u % CPR;
            const int pos = min(p0 + vr, p_max);
            pv[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.v_cache) + ((size_t)pos * kvs + (size_t)kvh * HD) * ESZ + vc * 16);
        }
    };
    auto commit_stage = [&](int p0) {
#pragma unroll
        for (int q = 0; q < PPT / 4; q++) {
            const float e[4] = {pp[q].x, pp[q].y, pp[q].z, pp[q].w};
            float m[4];
#pragma unroll
            for (int j = 0; j < 4; j++) m[j] = (prow < R && p0 + pseg + 4 * q + j <= qpos) ? (inv ? e[j] * iv : e[j]) : 0.0f;      // select first: the scratch past qpos is not initialised
            *reinterpret_cast<xm_f4*>(Ps + prow * PLD + pseg + 4 * q) = xm_f4{m[0], m[1], m[2], m[3]};
        }
#pragma unroll
        for (int i = 0; i < VCH; i++) { const int u = tid + 256 * i, vr = u / CPR, vc = u % CPR; *reinterpret_cast<u32x4*>(Vs + vr * VROW + vc * 16) = pv[i]; }
    };
    load_stage(0);
    for (int p0 = 0; p0 <= p_max; p0 += PS) {
        __syncthreads();                                                      // the previous stage's readers are done
        commit_stage(p0);
        if (p0 + PS <= p_max) load_stage(p0 + PS);                            // in flight during the MFMAs below
        __syncthreads();
        if (wave < NW) {
            const int np = min(PS, p_max + 1 - p0);                           // positions of this stage any query may see
            const float* pr = Ps + r31 * PLD + kh;
            const char* vb = Vs + kh * VROW + (size_t)(wave * NB * 32 + r31) * ESZ;
#pragma unroll 4
            for (int s2 = 0; s2 < PS / 2; s2++) {                             // k = positions 2 s2 (lane half 0) and 2 s2 + 1 (lane half 1), ascending
                if (2 * s2 >= np) break;
                const float pa = pr[2 * s2];
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    float v;
                    if (FP8) v = kr_e4m3_to_f32(*reinterpret_cast<const uint8_t*>(vb + 2 * s2 * VROW + b * 32));
                    else v = __half2float(__ushort_as_half(*reinterpret_cast<const uint16_t*>(vb + 2 * s2 * VROW + b * 64)));
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa, v, acc[b], 0, 0, 0);
                }
            }
        }
    }
    if (wave >= NW) return;
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int r = (i & 3) + 8 * (i >> 2) + 4 * kh;
            if (r < R) {
                const int tt = r >> lg, hh = kvh * group + (r & gm);
                float o = acc[b][i];
                const size_t oi = (size_t)(t0 + tt) * a.nh * HD + (size_t)hh * HD + (size_t)(wave * NB + b) * 32 + r31;
                if (a.gated) { const float gt = a.gate[oi]; o *= 1.0f / (1.0f + kr_expf(-gt)); }
                a.attn_out[oi] = o;
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// MLA prompt pass, exact scores (decode.rs mla_attn_dot_fp16_avx2 twice: kr_dot2acc of kr_mla.hip): per (token, head) row and position
//     v = dot16(q_abs[row], ckv[pos]) + dot16(q_pe[row], kpe[pos]);  v *= sm_scale
// where dot16 keeps 16 fma chains (chain j over elements 16 s + j, s ascending) and folds them with the AVX2 tree -- the router's structure
// (kr_route_mfma.hip): 16 accumulators of a 32 x 32 block fed with (s, s + 1) pairs, tree (c0 + c1) + (c2 + c3), c0 = (a0 + a8) + (a4 + a12) ...
// grid (position tiles of 64, row tiles of 64, 1); rows are (token, head) pairs in token-major order; 64 k per LDS stage.
// ------------------------------------------------------------------------------------------------------------------------------------
// Two launches, one dot each (holding the first dot's 16 folded values across a second 256-register accumulation made the register allocator spill
// accumulators to scratch, and a kernel with a scratch segment runs with a fraction of the waves: 7.8 ms instead of ~1): ROPE = true writes
// dot16(q_pe, kpe) of every visible (row, position) to the scratch; ROPE = false reads it back as the second addend: v = dot16(q_abs, ckv) + that; v *= sm_scale.
template <bool FP8, bool ROPE>
__global__ void __launch_bounds__(256) kr_mla_scores_mfma_kernel(const float* __restrict__ qb, const void* __restrict__ kbv, int nh, int K, int pos0, int rows, float sm_scale,
                                                                 float* __restrict__ sc, int sc_ld, float* __restrict__ tmax) {
    __shared__ __attribute__((aligned(16))) float As[64 * XM_LDF];
    __shared__ __attribute__((aligned(16))) float Bs[64 * XM_LDF];
    const int row0 = blockIdx.y * 64, R = min(64, rows - row0);
    const int lnh = __builtin_ctz(nh);                                        // nh divides 32: a power of two
    const int p_lo = blockIdx.x * 64, p_max = pos0 + ((row0 + R - 1) >> lnh);  // last position any row of the tile may see
    if (p_lo > p_max) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r31 = lane & 31, kh = lane >> 5;
    const int rb = (wave >> 1) * 32, pb = (wave & 1) * 32;                    // scalars: as lane values they were spilled across the 256-register epilogue (ROPE form)
    constexpr int ESZ = FP8 ? 1 : 2, KCH = FP8 ? 1 : 2;                       // 16-byte cache chunks per thread per 64-k stage
    // staging maps of a 64-k stage: q 64 rows x 64 floats (4 float4 per thread), cache 64 positions x 64 values (rows past p_max re-read p_max)
    int qrow[4], qc4[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { const int u = tid + 256 * i, row = u >> 4; qc4[i] = (u & 15) * 4; qrow[i] = row0 + (row < R ? row : R - 1); }
    int kpos[KCH], koff[KCH];
#pragma unroll
    for (int i = 0; i < KCH; i++) { const int u = tid + 256 * i, prow = FP8 ? (u >> 2) : (u >> 3); koff[i] = FP8 ? (u & 3) * 16 : (u & 7) * 16; kpos[i] = min(p_lo + prow, p_max); }
    xm_f4 qreg[4]; u32x4 kreg[KCH];
    auto load_stage = [&](const float* qb, const char* kb, int K, int k0) {
#pragma unroll
        for (int i = 0; i < 4; i++) qreg[i] = *reinterpret_cast<const xm_f4*>(qb + (size_t)qrow[i] * K + k0 + qc4[i]);
#pragma unroll
        for (int i = 0; i < KCH; i++) kreg[i] = *reinterpret_cast<const u32x4*>(kb + ((size_t)kpos[i] * K + k0) * ESZ + koff[i]);
    };
    auto commit_stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; i++) { const int u = tid + 256 * i, row = u >> 4; *reinterpret_cast<xm_f4*>(As + row * XM_LDF + qc4[i]) = qreg[i]; }
        if (FP8) {
            const uint32_t ww[4] = {kreg[0].x, kreg[0].y, kreg[0].z, kreg[0].w};
            float* dst = Bs + (tid >> 2) * XM_LDF + (tid & 3) * 16;
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++)
                *reinterpret_cast<xm_f4*>(dst + 4 * q4) = xm_f4{kr_e4m3_to_f32((uint8_t)ww[q4]), kr_e4m3_to_f32((uint8_t)(ww[q4] >> 8)),
                                                              kr_e4m3_to_f32((uint8_t)(ww[q4] >> 16)), kr_e4m3_to_f32((uint8_t)(ww[q4] >> 24))};
        } else {
#pragma unroll
            for (int i = 0; i < KCH; i++) {
                const int u = tid + 256 * i, prow = u >> 3, c8 = (u & 7) * 8;
                const uint32_t ww[4] = {kreg[i].x, kreg[i].y, kreg[i].z, kreg[i].w};
                float f[8];
#pragma unroll
                for (int j = 0; j < 4; j++) { f[2 * j] = __half2float(__ushort_as_half((uint16_t)(ww[j] & 0xFFFFu))); f[2 * j + 1] = __half2float(__ushort_as_half((uint16_t)(ww[j] >> 16))); }
                *reinterpret_cast<xm_f4*>(Bs + prow * XM_LDF + c8) = xm_f4{f[0], f[1], f[2], f[3]};
                *reinterpret_cast<xm_f4*>(Bs + prow * XM_LDF + c8 + 4) = xm_f4{f[4], f[5], f[6], f[7]};
            }
        }
    };
    const char* kb = reinterpret_cast<const char*>(kbv);
    xm_v16f acc[16];
#pragma unroll
    for (int j = 0; j < 16; j++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[j][i] = 0.0f;
    // 16 chains, 64 k per stage, next stage requested under the MFMAs
    load_stage(qb, kb, K, 0);
    for (int k0 = 0; k0 < K; k0 += 64) {
        __syncthreads();
        commit_stage();
        if (k0 + 64 < K) load_stage(qb, kb, K, k0 + 64);
        __syncthreads();
        const float* ap = As + (rb + r31) * XM_LDF + 16 * kh;
        const float* bp = Bs + (pb + r31) * XM_LDF + 16 * kh;
#pragma unroll
        for (int m = 0; m < 2; m++) {                                         // 32 consecutive k: the (s, s + 1) pair of every chain
            float av[16], bv[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const xm_f4 x = *reinterpret_cast<const xm_f4*>(ap + 32 * m + 4 * q), y = *reinterpret_cast<const xm_f4*>(bp + 32 * m + 4 * q);
                av[4 * q] = x.x; av[4 * q + 1] = x.y; av[4 * q + 2] = x.z; av[4 * q + 3] = x.w;
                bv[4 * q] = y.x; bv[4 * q + 1] = y.y; bv[4 * q + 2] = y.z; bv[4 * q + 3] = y.w;
            }
#pragma unroll
            for (int j = 0; j < 16; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc[j], 0, 0, 0);
        }
    }
    // the lane id is formed again here (v_mbcnt needs no live register): kept across the k loop, the lane half was the one value the 256-register tree spilled
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int r31e = lane_e & 31, khe = lane_e >> 5;
    const int pos = p_lo + pb + r31e;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int r = rb + (i & 3) + 8 * (i >> 2) + 4 * khe;
        const float c0 = (acc[0][i] + acc[8][i]) + (acc[4][i] + acc[12][i]);
        const float c1 = (acc[1][i] + acc[9][i]) + (acc[5][i] + acc[13][i]);
        const float c2 = (acc[2][i] + acc[10][i]) + (acc[6][i] + acc[14][i]);
        const float c3 = (acc[3][i] + acc[11][i]) + (acc[7][i] + acc[15][i]);
        const float d = (c0 + c1) + (c2 + c3);
        float mv = -__builtin_inff();
        const size_t rowi = (size_t)row0 + (r < R ? r : 0);
        if (r < R && pos <= pos0 + (int)(rowi >> lnh)) {
            if (ROPE) sc[rowi * sc_ld + pos] = d;                             // the second addend, picked up by the latent launch
            else { mv = (d + sc[rowi * sc_ld + pos]) * sm_scale; sc[rowi * sc_ld + pos] = mv; }      // v = dot(latent) + dot(rope); v *= sm_scale
        }
        if (!ROPE && tmax) {
            mv = kr_red16_max_f32(mv);                            // 4 DPP steps inside each 16-lane row, then one exchange between the two rows of the lane half
            mv = fmaxf(mv, __shfl_xor(mv, 16));
            if (r31e == 0 && r < R) tmax[rowi * (size_t)(sc_ld >> 5) + ((p_lo + pb) >> 5)] = mv;
        }
    }
}

// scores + P.V of the exact prompt pass on the matrix cores; non-zero = geometry not covered (the caller keeps the vector kernels)
int kr_pfm_gqa_exact_mfma_ok(const KrPfmGqaArgs& a) {
    const int group = a.nkv > 0 ? a.nh / a.nkv : 0;
    return group >= 1 && group <= 32 && (32 % group) == 0 && a.nh % a.nkv == 0 && (a.hd == 64 || a.hd == 128 || a.hd == 256);
}
void kr_launch_pfm_gqa_scores_mfma(const KrPfmGqaArgs& a, int C, float* sc, int sc_ld, float* tmax, hipStream_t st) {
    const int group = a.nh / a.nkv, TT = 64 / group;
    const dim3 grid((a.pos0 + C + 63) / 64, (C + TT - 1) / TT, a.nkv);
    if (a.kv_fp8) hipLaunchKernelGGL(kr_pfm_gqa_scores_mfma_kernel<true>, grid, dim3(256), 0, st, a, sc, sc_ld, C, tmax);
    else hipLaunchKernelGGL(kr_pfm_gqa_scores_mfma_kernel<false>, grid, dim3(256), 0, st, a, sc, sc_ld, C, tmax);
}
void kr_launch_pfm_gqa_pv_mfma(const KrPfmGqaArgs& a, int C, const float* sc, int sc_ld, const float* inv, hipStream_t st) {
    const int group = a.nh / a.nkv, TT = 32 / group;
    const dim3 grid((C + TT - 1) / TT, a.nkv);
#define KR_XPV(F_, H_, P_) hipLaunchKernelGGL((kr_pfm_gqa_pv_mfma_kernel<F_, H_, P_>), grid, dim3(256), 0, st, a, sc, sc_ld, C, inv)
    if (a.hd == 512) { if (a.kv_fp8) KR_XPV(true, 512, 32); else KR_XPV(false, 512, 32); }      // MLA latent rows (kr_launch_mla_exact_mfma)
    else if (a.hd == 256) { if (a.kv_fp8) KR_XPV(true, 256, 64); else KR_XPV(false, 256, 64); }
    else if (a.hd == 128) { if (a.kv_fp8) KR_XPV(true, 128, 64); else KR_XPV(false, 128, 64); }
    else { if (a.kv_fp8) KR_XPV(true, 64, 64); else KR_XPV(false, 64, 64); }
#undef KR_XPV
}
// MLA prompt pass: scores of n_tok tokens x nh heads against the latent + rope caches -> score scratch sc[n_tok * nh][sc_ld] + row maxima per 32
// positions; non-zero = geometry not covered (klr % 128, rd != 64, nh does not divide 32)
int kr_mla_exact_mfma_ok(int nh, int klr, int rd) { return nh >= 1 && nh <= 32 && (32 % nh) == 0 && (klr == 512 || klr == 256) && rd == 64; }
int kr_launch_mla_scores_mfma(const float* q_abs, const float* q_pe, const void* ckv, const void* kpe, int kv_fp8, int nh, int klr, int rd, int pos0, int n_tok,
                              float sm_scale, float* sc, int sc_ld, float* tmax, hipStream_t st) {
    if (!kr_mla_exact_mfma_ok(nh, klr, rd)) return 1;
    const int rows = n_tok * nh;
    const dim3 grid((pos0 + n_tok + 63) / 64, (rows + 63) / 64);
    if (klr % 64 || rd % 64) return 1;
#define KR_MS(F_, R_, Q_, C_, K_) hipLaunchKernelGGL((kr_mla_scores_mfma_kernel<F_, R_>), grid, dim3(256), 0, st, Q_, C_, nh, K_, pos0, rows, sm_scale, sc, sc_ld, tmax)
    if (kv_fp8) { KR_MS(true, true, q_pe, kpe, rd); KR_MS(true, false, q_abs, ckv, klr); }
    else { KR_MS(false, true, q_pe, kpe, rd); KR_MS(false, false, q_abs, ckv, klr); }
#undef KR_MS
    return 0;
}
