// kr_route_mfma.hip -- router logits of the prompt pass (rule DECODE, moe_route_matmul_avx2, src/decode.rs:1385) on the f32 matrix cores,
// BIT-IDENTICAL to the 16-chain GEMV of kr_router.hip (the ids that follow must stay bit-exact).
//
// The reference keeps two 8-lane fma accumulators per expert row: logit(t, e) is 16 independent chains
//     a_j <- fma(w[e][16 s + j], x[t][16 s + j], a_j),   s = 0 .. H/16 - 1 in order,   j = 0 .. 15
// folded by a fixed tree ((a0+a8)+(a4+a12) + (a1+a9)+(a5+a13)) + ((a2+a10)+(a6+a14) + (a3+a11)+(a7+a15)), then + bias.
// v_mfma_f32_32x32x2_f32 computes D = fma(A[.][1], B[1][.], fma(A[.][0], B[0][.], C)) -- two fused steps in k order (checked bit for bit
// against the GEMV by tests/test_router_gpu.py and by every prompt-pass == decode test) -- so chain j over a 32-token x 32-expert block is ONE
// accumulator fed with (s, s+1) pairs: A[t][k] = x[t][16 (2m + k) + j], B[k][e] = w[e][16 (2m + k) + j].  A wave owns a block and all 16
// chains (16 accumulators = 256 AGPRs); per 32 consecutive k it reads 16 floats of x and 16 gate values per lane (4 + 2 LDS reads) for 16
// MFMAs.  The kernel runs at the f32-MFMA rate (2.1 GFLOP per 1024-token chunk and layer): 131 us as a GEMV -> see DESIGN.md 5b.
// Workgroup: 4 waves = 64 tokens x 64 experts, 128-k stages double-buffered in LDS (next stage's global loads in flight during the MFMAs).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>

#include "kr_lds_optin.h"
#include "kr_router.h"

typedef float rm_v16f __attribute__((ext_vector_type(16)));
typedef float rm_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t rm_u4 __attribute__((ext_vector_type(4)));
#define RM_KS 128                  // k per stage
#define RM_LDA (RM_KS + 4)         // floats per x row in LDS
#define RM_LDB16 (RM_KS + 8)       // bf16 per gate row in LDS (272 B: rows 4 banks apart)
#define RM_LDB32 (RM_KS + 4)       // f32 gate row

// Batched form (blockIdx.z): the same 16-chain dot + fold tree is the reference's w_vc projection of MLA (mla_project_wvc_avx2, decode.rs:4555: two
// 8-lane accumulators over alternating 8-blocks = chains j = 8 a + l over elements 16 m + j, then acc0 + acc1 and the hsum tree) -- head h reads
// x + h * x_bs with row stride ldx, gate + h * g_bs, and writes logits + h * o_bs with row stride ldo.
// SPLIT: the problem has too few 64 x 64 tiles to fill the chip (the router of a 1024-token chunk: 128 workgroups of 256-register waves on 256 CUs).
// A workgroup then takes 32 tokens x 64 experts and its four waves are (expert block, CHAIN HALF): wave half c owns the 8 chains
// j = {0,1,4,5,8,9,12,13} + 2 c, i.e. the sub-trees (c0 + c1) or (c2 + c3) of the fold; the halves meet through LDS and the last add is the
// tree's root -- the same 15 adds in the same order.  Twice the workgroups, half the accumulators (two waves per SIMD).
template <bool GATE_BF16, bool SPLIT>
__global__ void __launch_bounds__(256, SPLIT ? 2 : 1) kr_route_logits_mfma_kernel(const void* __restrict__ gate_row_, const float* __restrict__ x_, const float* __restrict__ bias,
                                                                   float* __restrict__ logits_, int T, int E, int H, int ldx, int ldo, size_t x_bs, size_t g_bs, size_t o_bs) {
    const float* x = x_ + (size_t)blockIdx.z * x_bs; float* logits = logits_ + (size_t)blockIdx.z * o_bs;
    const void* gate_row = GATE_BF16 ? (const void*)(reinterpret_cast<const uint16_t*>(gate_row_) + (size_t)blockIdx.z * g_bs)
                                     : (const void*)(reinterpret_cast<const float*>(gate_row_) + (size_t)blockIdx.z * g_bs);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TT = SPLIT ? 32 : 64, NJ = SPLIT ? 8 : 16, NA = TT / 8;
    constexpr int A_BYTES = TT * RM_LDA * 4, B_BYTES = GATE_BF16 ? 64 * RM_LDB16 * 2 : 64 * RM_LDB32 * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int t0 = blockIdx.y * TT, e0 = blockIdx.x * 64;
    const int tb = SPLIT ? 0 : (wave >> 1) * 32, eb = (wave & 1) * 32;         // this wave's block inside the tile
    const int ch = SPLIT ? (wave >> 1) : 0;                                       // its chain half
    // ---- global -> register staging: x tile TT rows x 128 floats, gate tile 64 rows x 128 values
    rm_f4 pa[NA]; rm_u4 pb[GATE_BF16 ? 4 : 8];
    auto load_stage = [&](int st) {
        const int k0 = st * RM_KS;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int u = tid + 256 * i, row = u >> 5, c4 = (u & 31) * 4;
            const int tr = min(t0 + row, T - 1);
            pa[i] = *reinterpret_cast<const rm_f4*>(x + (size_t)tr * ldx + k0 + c4);
        }
        if (GATE_BF16) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int u = tid + 256 * i, row = u >> 4, c8 = (u & 15) * 8;
                const int er = min(e0 + row, E - 1);
                pb[i] = *reinterpret_cast<const rm_u4*>(reinterpret_cast<const uint16_t*>(gate_row) + (size_t)er * H + k0 + c8);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int u = tid + 256 * i, row = u >> 5, c4 = (u & 31) * 4;
                const int er = min(e0 + row, E - 1);
                pb[i] = *reinterpret_cast<const rm_u4*>(reinterpret_cast<const float*>(gate_row) + (size_t)er * H + k0 + c4);
            }
        }
    };
    auto commit_stage = [&](int buf) {
        float* As = reinterpret_cast<float*>(smem + buf * (A_BYTES + B_BYTES));
        char* Bs = smem + buf * (A_BYTES + B_BYTES) + A_BYTES;
#pragma unroll
        for (int i = 0; i < NA; i++) { const int u = tid + 256 * i, row = u >> 5, c4 = (u & 31) * 4; *reinterpret_cast<rm_f4*>(As + row * RM_LDA + c4) = pa[i]; }
        if (GATE_BF16) {
#pragma unroll
            for (int i = 0; i < 4; i++) { const int u = tid + 256 * i, row = u >> 4, c8 = (u & 15) * 8; *reinterpret_cast<rm_u4*>(Bs + (row * RM_LDB16 + c8) * 2) = pb[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) { const int u = tid + 256 * i, row = u >> 5, c4 = (u & 31) * 4; *reinterpret_cast<rm_u4*>(Bs + (row * RM_LDB32 + c4) * 4) = pb[i]; }
        }
    };
    rm_v16f acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[j][i] = 0.0f;
    const int nst = H / RM_KS;
    // the bias is fetched here: a conditional load between the k loop and the tree made the compiler copy all 256 accumulators out in front of it (one spill)
    const int e = e0 + eb + r;
    const float bv = (bias && e < E) ? bias[e] : 0.0f;
    load_stage(0);
    commit_stage(0);
    __syncthreads();
    for (int st = 0; st < nst; st++) {
        const int buf = st & 1;
        if (st + 1 < nst) load_stage(st + 1);
        const float* As = reinterpret_cast<const float*>(smem + buf * (A_BYTES + B_BYTES)) + (tb + r) * RM_LDA + 16 * h;
        const char* Bs = smem + buf * (A_BYTES + B_BYTES) + A_BYTES;
#pragma unroll
        for (int m = 0; m < RM_KS / 32; m++) {            // 32 consecutive k: the (s, s+1) pair of every chain
            float a[NJ], b[NJ];
            if (SPLIT) {      // local chain 2 q + i = chain 4 q + 2 ch + i: two consecutive values per group of four
                typedef float rm_f2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int q = 0; q < 4; q++) { const rm_f2 v = *reinterpret_cast<const rm_f2*>(As + 32 * m + 4 * q + 2 * ch); a[2 * q] = v.x; a[2 * q + 1] = v.y; }
                if (GATE_BF16) {
                    const uint16_t* bp = reinterpret_cast<const uint16_t*>(Bs) + (eb + r) * RM_LDB16 + 32 * m + 16 * h + 2 * ch;
#pragma unroll
                    for (int q = 0; q < 4; q++) { const uint32_t w = *reinterpret_cast<const uint32_t*>(bp + 4 * q); b[2 * q] = __uint_as_float(w << 16); b[2 * q + 1] = __uint_as_float(w & 0xFFFF0000u); }
                } else {
                    const float* bp = reinterpret_cast<const float*>(Bs) + (eb + r) * RM_LDB32 + 32 * m + 16 * h + 2 * ch;
#pragma unroll
                    for (int q = 0; q < 4; q++) { const rm_f2 v = *reinterpret_cast<const rm_f2*>(bp + 4 * q); b[2 * q] = v.x; b[2 * q + 1] = v.y; }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++) { const rm_f4 v = *reinterpret_cast<const rm_f4*>(As + 32 * m + 4 * q); a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w; }
                if (GATE_BF16) {
                    const uint16_t* bp = reinterpret_cast<const uint16_t*>(Bs) + (eb + r) * RM_LDB16 + 32 * m + 16 * h;
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const rm_u4 v = *reinterpret_cast<const rm_u4*>(bp + 8 * q);
                        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int p = 0; p < 4; p++) { b[8 * q + 2 * p] = __uint_as_float(w4[p] << 16); b[8 * q + 2 * p + 1] = __uint_as_float(w4[p] & 0xFFFF0000u); }
                    }
                } else {
                    const float* bp = reinterpret_cast<const float*>(Bs) + (eb + r) * RM_LDB32 + 32 * m + 16 * h;
#pragma unroll
                    for (int q = 0; q < 4; q++) { const rm_f4 v = *reinterpret_cast<const rm_f4*>(bp + 4 * q); b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w; }
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[j], 0, 0, 0);
        }
        if (st + 1 < nst) commit_stage(buf ^ 1);          // the other buffer: its readers finished before the barrier that ended stage st - 1
        __syncthreads();
    }
    // ---- the reference's tree over the 16 chains (decode.rs:1419-1427), then + bias (decode.rs:3292)
    if (SPLIT) {
        // local chains l = 2 q + i <-> chain 4 q + 2 ch + i: (l0 + l4) + (l2 + l6) is (a0 + a8) + (a4 + a12) for half 0 and (a2 + a10) + (a6 + a14) for half 1
        float* xch = reinterpret_cast<float*>(smem) + ((wave & 1) * 64 + lane) * 16;      // the stage buffers are free after the last barrier
        float half[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float ca = (acc[0][i] + acc[4][i]) + (acc[2][i] + acc[6][i]);
            const float cb = (acc[1][i] + acc[5][i]) + (acc[3][i] + acc[7][i]);
            half[i] = ca + cb;
        }
        if (ch == 1) {
#pragma unroll
            for (int q = 0; q < 4; q++) *reinterpret_cast<rm_f4*>(xch + 4 * q) = rm_f4{half[4 * q], half[4 * q + 1], half[4 * q + 2], half[4 * q + 3]};
        }
        __syncthreads();
        if (ch == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const rm_f4 o = *reinterpret_cast<const rm_f4*>(xch + 4 * q);
                const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const int i = 4 * q + p;
                    float v = half[i] + ov[p];
                    if (bias) v += bv;
                    const int t = t0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (t < T && e < E) logits[(size_t)t * ldo + e] = v;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float c0 = (acc[0][i] + acc[SPLIT ? 0 : 8][i]) + (acc[4][i] + acc[SPLIT ? 0 : 12][i]);
        const float c1 = (acc[1][i] + acc[SPLIT ? 0 : 9][i]) + (acc[5][i] + acc[SPLIT ? 0 : 13][i]);
        const float c2 = (acc[2][i] + acc[SPLIT ? 0 : 10][i]) + (acc[6][i] + acc[SPLIT ? 0 : 14][i]);
        const float c3 = (acc[3][i] + acc[SPLIT ? 0 : 11][i]) + (acc[7][i] + acc[SPLIT ? 0 : 15][i]);
        float v = (c0 + c1) + (c2 + c3);
        if (bias) v += bv;
        const int t = t0 + tb + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (t < T && e < E) logits[(size_t)t * ldo + e] = v;
    }
}

// non-zero = geometry not covered (caller keeps the GEMV)
static int rm_launch(const void* gate_row, int gate_bf16, const float* x, const float* bias, float* out, int T, int E, int H, int ldx, int ldo, int batch, size_t x_bs,
                     size_t g_bs, size_t o_bs, hipStream_t st) {
    if (H % RM_KS || T < 1 || E < 1 || batch < 1) return 1;
    // too few 64 x 64 tiles for the chip: 32-token tiles with the chains split over wave pairs (bit-identical, see the kernel)
    const bool split = (long)((E + 63) / 64) * ((T + 63) / 64) * batch < 256;
    const int tt = split ? 32 : 64;
    const dim3 grid((E + 63) / 64, (T + tt - 1) / tt, batch);
    const size_t lds = 2 * ((size_t)tt * RM_LDA * 4 + (gate_bf16 ? (size_t)64 * RM_LDB16 * 2 : (size_t)64 * RM_LDB32 * 4));
#define KR_RM(B_, S_) do { if (kr_lds_optin(reinterpret_cast<const void*>(kr_route_logits_mfma_kernel<B_, S_>), lds)) return 1; \
        hipLaunchKernelGGL((kr_route_logits_mfma_kernel<B_, S_>), grid, dim3(256), lds, st, gate_row, x, bias, out, T, E, H, ldx, ldo, x_bs, g_bs, o_bs); } while (0)
    if (gate_bf16) { if (split) KR_RM(true, true); else KR_RM(true, false); }
    else { if (split) KR_RM(false, true); else KR_RM(false, false); }
#undef KR_RM
    return 0;
}
int kr_launch_route_logits_mfma(const void* gate_row, int gate_bf16, const float* x, const float* bias, float* logits, int T, int E, int H, hipStream_t st) {
    return rm_launch(gate_row, gate_bf16, x, bias, logits, T, E, H, H, E, 1, 0, 0, 0, st);
}
// MLA prompt pass: v_proj[t][h][o] = w_vc[h][o][:] . attn_lat[t][h][:] for all tokens and heads (bit-identical to kr_mla_wvc_kernel)
int kr_launch_mla_wvc_mfma(const float* w_vc, const float* attn_lat, float* v_proj, int T, int nh, int vhd, int klr, hipStream_t st) {
    return rm_launch(w_vc, 0, attn_lat, nullptr, v_proj, T, vhd, klr, nh * klr, nh * vhd, nh, (size_t)klr, (size_t)vhd * klr, (size_t)vhd, st);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// TOLERANCE form of the router logits (KR_GEMM_FAST prompt pass; round 4).  The exact kernel above reproduces the reference's 16 fma chains on the
// f32 MFMA -- 1/16 of the bf16 matrix rate, 512 accumulator registers, one wave per SIMD: 215 us per 2731-token chunk and layer (6.6 % of the kernel
// time of the tolerance prompt pass, profiles/r03_prefill_8192_attn_fast_gemm_fast_kernel_stats.txt), 17 % of the f32 MFMA peak.  In the tolerance pass
// the router's INPUT already differs from the exact pass in its last bits (f16 GEMM operands upstream), so what the mode keeps is "router ids exact for
// the logits it computes, logits to a stated bound" -- as KR_DECODE_FAST does with its tree sums.  Here: gate values are bf16 (checkpoints are; the
// launcher refuses an f32 gate), x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (x to 2^-17 relative), logit = sum over k of (hi + lo) * g on
// v_mfma_f32_32x32x16_bf16 with f32 accumulation: two MFMAs per 16 k at 16 x the f32 MFMA rate.  No LDS: a lane's A fragment is 32 contiguous bytes
// of its token row, its B fragment 16 contiguous bytes of its expert's gate row; both come from L2 (the x tile is read by the E / 128 column tiles,
// the 2 MB gate by every row tile).  Workgroup: 4 waves = 64 tokens x 128 experts, each wave 32 x 64 (two accumulators); the k loop keeps the
// fragments of the next two 16-k steps in flight.
// STATED TOLERANCE (tests/test_router_gpu.py): |fast - exact| <= 2e-5 * sum_k |x_k g_k| per logit (measured ~3e-6); top-k ids equal to the exact
// kernel's on every token whose k-th and (k+1)-th scores are further apart than that.
// ------------------------------------------------------------------------------------------------------------------------------------
typedef __bf16 rm_b8 __attribute__((ext_vector_type(8)));
// k mapping: the MFMA sums its 16 k in any order as long as A and B agree, so an iteration covers 32 k and lane half h takes the CONTIGUOUS 16 of them
// [32 j + 16 h, + 16) -- two MFMA steps (its first / second 8) -- instead of two separate 8-k chunks: a lane reads 64 contiguous bytes of its token row and
// 32 of its gate row, and the two lanes of a row cover one whole 128-byte line (the first form asked for half lines and ran at 120 us per 2752-token chunk).
// KS = 4 (round 6): the k range split over the FOUR waves of a workgroup -- one 32 x 64 output tile per workgroup, partial accumulators added in wave order
// through LDS.  The two-wave form above put 688 waves on the chip's 1024 SIMDs for a 2752-token chunk and each of them walked all of K behind its own
// hi / lo conversions (8 vector instructions per MFMA): 101 us per chunk and layer, 114 TFLOP/s.  Same products; the f32 sum of a logit is now four partial
// sums in k order added in wave order -- inside the bound stated above.
template <int KS>
__global__ void __launch_bounds__(64 * (KS == 1 ? 2 : KS), 2) kr_route_logits_fast_kernel(const uint16_t* __restrict__ gate_row, const float* __restrict__ x, const float* __restrict__ bias,
                                                                   float* __restrict__ logits, int T, int E, int H) {
    __shared__ float s_part[KS == 1 ? 1 : (KS - 1) * 2 * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    // KS == 1: two waves per workgroup = 32 tokens x 128 experts, each wave all of K; KS > 1: KS waves = 32 tokens x 64 experts, each wave H / KS of K
    const int t0 = blockIdx.y * 32, e0 = KS == 1 ? blockIdx.x * 128 + wave * 64 : blockIdx.x * 64;
    const int kq = KS == 1 ? 0 : wave * (H / KS);
    const int tr = min(t0 + r, T - 1);
    const float* xp = x + (size_t)tr * H + kq + 16 * h;
    const uint16_t* gp[2];
#pragma unroll
    for (int c = 0; c < 2; c++) gp[c] = gate_row + (size_t)min(e0 + 32 * c + r, E - 1) * H + kq + 16 * h;
    rm_v16f acc[2];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[c][i] = 0.0f;
    constexpr int PD = 2;      // 32-k iterations in flight
    rm_f4 xa[PD][4]; rm_u4 gb[PD][2][2];
    auto fetch = [&](int j, int buf) {
#pragma unroll
        for (int q = 0; q < 4; q++) xa[buf][q] = *reinterpret_cast<const rm_f4*>(xp + 32 * j + 4 * q);
#pragma unroll
        for (int c = 0; c < 2; c++) { gb[buf][c][0] = *reinterpret_cast<const rm_u4*>(gp[c] + 32 * j); gb[buf][c][1] = *reinterpret_cast<const rm_u4*>(gp[c] + 32 * j + 8); }
    };
    const int nit = H / 32 / KS;
#pragma unroll
    for (int p = 0; p < PD; p++) if (p < nit) fetch(p, p);
    for (int j0 = 0; j0 < nit; j0 += PD) {
#pragma unroll
        for (int p = 0; p < PD; p++) {
            if (j0 + p >= nit) break;
            rm_b8 ahi[2], alo[2], b[2][2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const float v[8] = {xa[p][2 * u].x, xa[p][2 * u].y, xa[p][2 * u].z, xa[p][2 * u].w, xa[p][2 * u + 1].x, xa[p][2 * u + 1].y, xa[p][2 * u + 1].z, xa[p][2 * u + 1].w};
#pragma unroll
                for (int i = 0; i < 8; i++) { const __bf16 hi = (__bf16)v[i]; ahi[u][i] = hi; alo[u][i] = (__bf16)(v[i] - (float)hi); }
#pragma unroll
                for (int c = 0; c < 2; c++) b[c][u] = __builtin_bit_cast(rm_b8, gb[p][c][u]);
            }
            if (j0 + p + PD < nit) fetch(j0 + p + PD, p);
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[u], b[c][u], acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[u], b[c][u], acc[c], 0, 0, 0);
                }
        }
    }
    if constexpr (KS > 1) {      // partial sums of waves 1 .. KS - 1 -> LDS ([wave - 1][c][i][lane]: conflict-free), wave 0 adds them in wave order
        if (wave > 0) {
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int i = 0; i < 16; i++) s_part[(((wave - 1) * 2 + c) * 16 + i) * 64 + lane] = acc[c][i];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 1; w < KS; w++)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[c][i] += s_part[(((w - 1) * 2 + c) * 16 + i) * 64 + lane];
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int e = e0 + 32 * c + r;
        const float bv = (bias && e < E) ? bias[e] : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int t = t0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (t < T && e < E) logits[(size_t)t * E + e] = acc[c][i] + bv;
        }
    }
}
// Round 6, second form: the same products as a workgroup-tiled GEMM.  The fragment-from-global kernels above re-read the x tile once per 64-expert column tile and the
// gate once per 32-token row tile -- 1 GB of L2 -> CU traffic per 8192-token launch (134 us: the L2 path, not the matrix pipe) -- and every wave converts the x values it
// reads.  Here a workgroup of four waves owns TM tokens x 128 experts: per 32-k step the x tile is read ONCE in whole 128-byte lines, split into its hi / lo bf16 planes
// by the thread that loaded it, and parked in LDS beside the gate tile (both double-buffered: the next step's global loads are in flight during the MFMAs, one barrier
// per step); a wave's fragments are 16-byte LDS reads (rows 80 bytes apart: conflict-free).  Per logit: MFMA(hi) then MFMA(lo) per 16 k, k ascending -- the order
// of the KS = 1 kernel.
#define RLT_LD 80            // bytes per LDS row: 32 bf16 + 16 pad
template <int TM>            // 64 or 128 tokens per workgroup; waves 2 x 2, each TM / 2 x 64
__global__ void __launch_bounds__(256) kr_route_logits_tiled_kernel(const uint16_t* __restrict__ gate_row, const float* __restrict__ x, const float* __restrict__ bias,
                                                                   float* __restrict__ logits, int T, int E, int H) {
    constexpr int NRB = TM / 64;                 // 32-row blocks per wave
    constexpr int XPT = TM / 32;                 // float4 of x per thread and step
    __shared__ __attribute__((aligned(16))) char lds[2][(2 * TM + 128) * RLT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int t0 = blockIdx.y * TM, e0 = blockIdx.x * 128;
    // staging maps: x -- row (tid >> 3) + 32 j, float4 tid & 7 of the step's 32 floats (8 lanes = one 128-byte line); gate -- row (tid >> 2) + 64 j, 16-byte chunk tid & 3
    const float* xp[XPT];
#pragma unroll
    for (int j = 0; j < XPT; j++) xp[j] = x + (size_t)min(t0 + (tid >> 3) + 32 * j, T - 1) * H + (tid & 7) * 4;
    const uint16_t* gp[2];
#pragma unroll
    for (int j = 0; j < 2; j++) gp[j] = gate_row + (size_t)min(e0 + (tid >> 2) + 64 * j, E - 1) * H + (tid & 3) * 8;
    rm_f4 px[XPT]; rm_u4 pg[2];
    auto fetch = [&](int st) {
#pragma unroll
        for (int j = 0; j < XPT; j++) px[j] = *reinterpret_cast<const rm_f4*>(xp[j] + 32 * st);
#pragma unroll
        for (int j = 0; j < 2; j++) pg[j] = *reinterpret_cast<const rm_u4*>(gp[j] + 32 * st);
    };
    auto commit = [&](int buf) {
        char* Xh = lds[buf]; char* Xl = Xh + TM * RLT_LD; char* G = Xl + TM * RLT_LD;
#pragma unroll
        for (int j = 0; j < XPT; j++) {
            const float v[4] = {px[j].x, px[j].y, px[j].z, px[j].w};
            uint16_t hi[4], lo[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { const __bf16 hb = (__bf16)v[i]; const __bf16 lb = (__bf16)(v[i] - (float)hb); hi[i] = __builtin_bit_cast(uint16_t, hb); lo[i] = __builtin_bit_cast(uint16_t, lb); }
            const int off = ((tid >> 3) + 32 * j) * RLT_LD + (tid & 7) * 8;
            *reinterpret_cast<uint2*>(Xh + off) = uint2{(uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16)};
            *reinterpret_cast<uint2*>(Xl + off) = uint2{(uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16)};
        }
#pragma unroll
        for (int j = 0; j < 2; j++) *reinterpret_cast<rm_u4*>(G + ((tid >> 2) + 64 * j) * RLT_LD + (tid & 3) * 16) = pg[j];
    };
    rm_v16f acc[NRB][2];
#pragma unroll
    for (int a_ = 0; a_ < NRB; a_++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[a_][c][i] = 0.0f;
    const int nst = H / 32;
    fetch(0); commit(0);
    __syncthreads();
    for (int st = 0; st < nst; st++) {
        const int buf = st & 1;
        if (st + 1 < nst) fetch(st + 1);
        const char* Xh = lds[buf]; const char* Xl = Xh + TM * RLT_LD; const char* G = Xl + TM * RLT_LD;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            rm_b8 ah[NRB], al[NRB], b[2];
#pragma unroll
            for (int a_ = 0; a_ < NRB; a_++) {
                const int row = wr * (TM / 2) + a_ * 32 + r;
                ah[a_] = *reinterpret_cast<const rm_b8*>(Xh + row * RLT_LD + ks * 32 + h * 16);
                al[a_] = *reinterpret_cast<const rm_b8*>(Xl + row * RLT_LD + ks * 32 + h * 16);
            }
#pragma unroll
            for (int c = 0; c < 2; c++) b[c] = *reinterpret_cast<const rm_b8*>(G + (wc * 64 + c * 32 + r) * RLT_LD + ks * 32 + h * 16);
            // every accumulator's hi product, then every accumulator's lo product: consecutive MFMAs never share an accumulator (per logit still hi, then lo)
#pragma unroll
            for (int a_ = 0; a_ < NRB; a_++)
#pragma unroll
                for (int c = 0; c < 2; c++) acc[a_][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a_], b[c], acc[a_][c], 0, 0, 0);
#pragma unroll
            for (int a_ = 0; a_ < NRB; a_++)
#pragma unroll
                for (int c = 0; c < 2; c++) acc[a_][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a_], b[c], acc[a_][c], 0, 0, 0);
        }
        if (st + 1 < nst) commit(buf ^ 1);      // the other buffer: its last readers passed the barrier of the previous step
        __syncthreads();
    }
#pragma unroll
    for (int a_ = 0; a_ < NRB; a_++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int e = e0 + wc * 64 + c * 32 + r;
            const float bv = (bias && e < E) ? bias[e] : 0.0f;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int t = t0 + wr * (TM / 2) + a_ * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (t < T && e < E) logits[(size_t)t * E + e] = acc[a_][c][i] + bv;
            }
        }
}
// non-zero = not covered (f32 gate, H not a multiple of 32): the caller keeps the exact kernel
int kr_launch_route_logits_fast(const void* gate_row, int gate_bf16, const float* x, const float* bias, float* logits, int T, int E, int H, hipStream_t st) {
    if (!gate_bf16 || H % 32 || T < 1 || E < 1) return 1;
    {   // workgroup-tiled form where its tiles still give (nearly) every CU a workgroup: 128-token tiles from ~7 k tokens x 512 experts on (8192 tokens: 74 us against 134),
        // 64-token tiles for shorter chunks only when the k-split form below does not apply (2752 tokens: both 58 us)
        const int n128 = (E + 127) / 128;
        const bool ksplit_ok = H % 128 == 0 && H >= 512;
        if ((long)((T + 127) / 128) * n128 >= 224) {
            hipLaunchKernelGGL(kr_route_logits_tiled_kernel<128>, dim3(n128, (T + 127) / 128), dim3(256), 0, st, reinterpret_cast<const uint16_t*>(gate_row), x, bias, logits, T, E, H);
            return 0;
        }
        if (T >= 256 && !ksplit_ok) {
            hipLaunchKernelGGL(kr_route_logits_tiled_kernel<64>, dim3(n128, (T + 63) / 64), dim3(256), 0, st, reinterpret_cast<const uint16_t*>(gate_row), x, bias, logits, T, E, H);
            return 0;
        }
    }
    if (H % 128 == 0 && H >= 512) {      // four k quarters per output tile: 4 x the waves (2752 tokens: 57.7 us against 70.1 stand-alone, profiles/r06_route_fast_kernel_stats.txt)
        const dim3 grid((E + 63) / 64, (T + 31) / 32);
        hipLaunchKernelGGL(kr_route_logits_fast_kernel<4>, grid, dim3(256), 0, st, reinterpret_cast<const uint16_t*>(gate_row), x, bias, logits, T, E, H);
        return 0;
    }
    const dim3 grid((E + 127) / 128, (T + 31) / 32);
    hipLaunchKernelGGL(kr_route_logits_fast_kernel<1>, grid, dim3(128), 0, st, reinterpret_cast<const uint16_t*>(gate_row), x, bias, logits, T, E, H);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// MLA prompt pass: the w_kc absorption of a chunk (mla_absorb_wkc_avx2, decode.rs:4508): q_abs[t][h][j] = chain over i ascending of
// fma(q[t][h][i], w_kc[h][i][j], acc) -- ONE chain per output, so one accumulator per 32 x 32 block and the k pairs (2m, 2m+1) in order.
// Workgroup: 64 tokens x 64 latent columns of one head, K = nd in one stage.  The q tile is stored with the k of every 8-group de-interleaved
// (even k first), so the lane half that supplies k = 2m + h reads four consecutive floats.  Bit-identical to kr_mla_prep_kernel's loop.
// ------------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) kr_mla_absorb_mfma_kernel(const float* __restrict__ q_full, int ld_q, int hd, int nd, const float* __restrict__ w_kc, int klr,
                                                                 float* __restrict__ q_abs, int T, int nh) {
    extern __shared__ __attribute__((aligned(16))) float ab_smem[];
    const int lda = nd + 4, ldb = 68;
    float* As = ab_smem; float* Bs = As + 64 * lda;
    const int j0 = blockIdx.x * 64, t0 = blockIdx.y * 64, h = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    for (int i = tid; i < 64 * (nd / 4); i += 256) {
        const int row = i / (nd / 4), c4 = (i % (nd / 4)) * 4, t = min(t0 + row, T - 1);
        const rm_f4 v = *reinterpret_cast<const rm_f4*>(q_full + (size_t)t * ld_q + (size_t)h * hd + c4);
        float* d = As + row * lda + (c4 & ~7);                   // k = c4 .. c4 + 3 of the 8-group: even k -> slots 0..3, odd k -> slots 4..7
        const int e0 = (c4 & 7) >> 1;
        d[e0] = v.x; d[4 + e0] = v.y; d[e0 + 1] = v.z; d[4 + e0 + 1] = v.w;
    }
    for (int i = tid; i < nd * 16; i += 256) {
        const int k = i >> 4, c4 = (i & 15) * 4;
        *reinterpret_cast<rm_f4*>(Bs + k * ldb + c4) = *reinterpret_cast<const rm_f4*>(w_kc + ((size_t)h * nd + k) * klr + j0 + c4);
    }
    __syncthreads();
    const int tb = (wave >> 1) * 32, jb = (wave & 1) * 32;
    rm_v16f acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    for (int q8 = 0; q8 < nd; q8 += 8) {
        const rm_f4 a4 = *reinterpret_cast<const rm_f4*>(As + (tb + r) * lda + q8 + 4 * hh);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
        float bv[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) bv[s2] = Bs[(q8 + 2 * s2 + hh) * ldb + jb + r];
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], bv[s2], acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int t = t0 + tb + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (t < T) q_abs[((size_t)t * nh + h) * klr + j0 + jb + r] = acc[i];
    }
}
int kr_launch_mla_absorb_mfma(const float* q_full, int ld_q, int hd, int nd, const float* w_kc, int klr, float* q_abs, int T, int nh, hipStream_t st) {
    if (nd % 8 || klr % 64 || T < 1) return 1;
    const size_t lds = (size_t)(64 * (nd + 4) + nd * 68) * 4;
    if (lds > 160 * 1024 || kr_lds_optin(reinterpret_cast<const void*>(kr_mla_absorb_mfma_kernel), lds)) return 1;
    hipLaunchKernelGGL(kr_mla_absorb_mfma_kernel, dim3(klr / 64, (T + 63) / 64, nh), dim3(256), lds, st, q_full, ld_q, hd, nd, w_kc, klr, q_abs, T, nh);
    return 0;
}
