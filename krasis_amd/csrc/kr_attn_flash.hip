// kr_attn_flash.hip -- prompt-pass GQA attention on the matrix cores (FAST / tolerance mode, kr_decode_set_attention_mode).
//
// The exact prompt pass (kr_prefill_ops.hip: scores -> softmax -> P.V passes over an HBM score scratch) keeps the reference CPU decode's
// operation order per query and runs on the vector ALUs: its cost grows with the square of the prompt (12 GQA layers of a 49 863-token
// prompt: ~2.5e14 flop) and dominates long prompts.  The reference's own GPU prefill uses flashinfer's bf16 flash attention
// (python/krasis/attention.py:612-640) -- i.e. tolerance-level numerics are what the reference itself ships for this step.  This kernel is
// the gfx950 counterpart: causal flash attention over the FP16 / FP8-E4M3 cache with v_mfma_f32_32x32x16_f16, f32 accumulation and f32
// online softmax; q and the probabilities are rounded to f16 (2^-11, eight times finer than the reference's bf16), K / V are exact in f16.
//
// Layout of the computation (one workgroup = 128 query rows = the G query heads of one KV head x 128 / G consecutive tokens, 4 waves x 32 rows,
// so a staged K / V tile serves every head of the group):
//   S^T = K Q^T   : A = K tile rows from LDS ([position][dim], 16-byte reads), B = Q^T kept in registers for the whole KV loop
//                   -> accumulator column = query row = lane % 32: the softmax of a row is a LANE-LOCAL reduction over 32 values plus one
//                   exchange with lane ^ 32 (no cross-lane butterflies), and the rescale factor of O is a per-lane scalar.
//   O^T = V^T P^T : A = V^T tile from LDS ([dim][position], written transposed -- two positions per dword -- while the tile is staged),
//                   B = P^T straight from the S^T accumulators (f16-packed in registers): the k order of an MFMA is free as long as A and
//                   B agree, so B uses the accumulator's own row order {0-3, 8-11} / {4-7, 12-15} and A reads V^T in two 8-byte pieces.
// The V tile of the current 64 positions is fetched while S^T and the softmax run, the next K tile while O^T accumulates (two barriers per tile).
#include "kr_device.h"
#include "kr_libm.h"
#include "kr_lds_optin.h"
#include "kr_prefill_ops.h"

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define FA_TK 64
#define FA_ROWS 128

__device__ __forceinline__ uint32_t fa_fp8x2_to_h2(uint32_t w, bool hi) {      // two E4M3 bytes -> packed f16 pair (exact)
    const v2f f = hi ? __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true) : __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false);
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(f.x, f.y));
}

template <int HD, bool FP8>
__global__ void __launch_bounds__(256) kr_pfm_gqa_flash_kernel(const KrPfmGqaArgs a, int C) {
    constexpr int KSTEPS = HD / 16, DB = HD / 32, LDK = HD * 2 + 16, LDV = FA_TK * 2 + 16;
    constexpr int CPR = FP8 ? HD / 16 : HD / 8;                 // 16-byte global chunks per cache row
    constexpr int KCH = FA_TK * CPR / 256;                      // K chunks per thread per tile  (>= 1 for HD >= 64)
    constexpr int VUN = (FA_TK / 2) * CPR;                      // V units (position pair x chunk) per tile
    constexpr int VPT = (VUN + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char fa_smem[];
    char* Ks = fa_smem;                                         // [64 positions][LDK]   f16
    char* Vt = fa_smem + FA_TK * LDK;                           // [HD dims][LDV]        f16, positions contiguous
    const int G = a.nh / a.nkv, TQ = FA_ROWS / G;
    const int kvh = blockIdx.y, t0 = blockIdx.x * TQ;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n31 = lane & 31, khalf = lane >> 5;
    const int r = wave * 32 + n31, hl = r / TQ, ti = r % TQ, tok = t0 + ti;
    const bool row_ok = tok < C;
    const int h = kvh * G + hl, p_q = a.pos0 + tok;
    const int kvs = a.nkv * HD, esz = FP8 ? 1 : 2;
    const int kv_end = a.pos0 + (t0 + TQ < C ? t0 + TQ : C);     // positions [0, kv_end) are visible to some row of the tile
    const int n_tiles = (kv_end + FA_TK - 1) / FA_TK;
    const int full_vis = a.pos0 + t0;                            // positions <= full_vis are visible to EVERY row

    // ---- Q^T fragments (B operand): row = this lane's query, k = dims 16 ks + 8 khalf + i; scale and log2(e) folded in, rounded to f16
    v8h qf[KSTEPS];
    {
        const float* q = a.q_out + ((size_t)(row_ok ? tok : 0) * a.nh + h) * HD + 8 * khalf;
        const float sc = a.sm_scale * 1.4426950408889634f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
            const float4 x0 = *reinterpret_cast<const float4*>(q + 16 * ks), x1 = *reinterpret_cast<const float4*>(q + 16 * ks + 4);
            const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int i = 0; i < 8; i++) qf[ks][i] = (_Float16)(row_ok ? xv[i] * sc : 0.0f);
        }
    }
    v16f oacc[DB];
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
        for (int i = 0; i < 16; i++) oacc[db][i] = 0.0f;
    float m_run = -__builtin_inff(), l_run = 0.0f;

    // ---- tile staging: global -> registers (issued one tile ahead) -> LDS
    const unsigned char* kc = reinterpret_cast<const unsigned char*>(a.k_cache) + (size_t)kvh * HD * esz;
    const unsigned char* vc = reinterpret_cast<const unsigned char*>(a.v_cache) + (size_t)kvh * HD * esz;
    u32x4 pk[KCH], pva[VPT], pvb[VPT];
    auto load_k = [&](int p0) {
#pragma unroll
        for (int j = 0; j < KCH; j++) {
            const int c = tid + j * 256, row = c / CPR, dc = c % CPR, p = p0 + row;
            pk[j] = p < kv_end ? *reinterpret_cast<const u32x4*>(kc + (size_t)p * kvs * esz + dc * 16) : u32x4{0, 0, 0, 0};
        }
    };
    auto load_v = [&](int p0) {
#pragma unroll
        for (int j = 0; j < VPT; j++) {
            const int u = tid + j * 256, pp = u & 31, dc = u >> 5, p = p0 + 2 * pp;      // lanes walk the position pairs: conflict-free transposed LDS writes
            pva[j] = u32x4{0, 0, 0, 0}; pvb[j] = u32x4{0, 0, 0, 0};
            if (u < VUN) {
                if (p < kv_end) pva[j] = *reinterpret_cast<const u32x4*>(vc + (size_t)p * kvs * esz + dc * 16);
                if (p + 1 < kv_end) pvb[j] = *reinterpret_cast<const u32x4*>(vc + (size_t)(p + 1) * kvs * esz + dc * 16);
            }
        }
    };
    auto commit_k = [&]() {
#pragma unroll
        for (int j = 0; j < KCH; j++) {
            const int c = tid + j * 256, row = c / CPR, dc = c % CPR;
            if (FP8) {
                const u32x4 w = pk[j];
                u32x4 lo = {fa_fp8x2_to_h2(w.x, false), fa_fp8x2_to_h2(w.x, true), fa_fp8x2_to_h2(w.y, false), fa_fp8x2_to_h2(w.y, true)};
                u32x4 hi = {fa_fp8x2_to_h2(w.z, false), fa_fp8x2_to_h2(w.z, true), fa_fp8x2_to_h2(w.w, false), fa_fp8x2_to_h2(w.w, true)};
                *reinterpret_cast<u32x4*>(Ks + row * LDK + dc * 32) = lo; *reinterpret_cast<u32x4*>(Ks + row * LDK + dc * 32 + 16) = hi;
            } else *reinterpret_cast<u32x4*>(Ks + row * LDK + dc * 16) = pk[j];
        }
    };
    auto commit_v = [&]() {
#pragma unroll
        for (int j = 0; j < VPT; j++) {
            const int u = tid + j * 256, pp = u & 31, dc = u >> 5;
            if (u < VUN) {
                uint32_t ha[8], hb[8];                    // f16 pairs of the two rows (FP16: 4 dwords each, FP8: 8 dwords each)
                constexpr int NW = FP8 ? 8 : 4;
                if (FP8) {
                    const uint32_t wa[4] = {pva[j].x, pva[j].y, pva[j].z, pva[j].w}, wb[4] = {pvb[j].x, pvb[j].y, pvb[j].z, pvb[j].w};
#pragma unroll
                    for (int m = 0; m < 4; m++) { ha[2 * m] = fa_fp8x2_to_h2(wa[m], false); ha[2 * m + 1] = fa_fp8x2_to_h2(wa[m], true);
                                                  hb[2 * m] = fa_fp8x2_to_h2(wb[m], false); hb[2 * m + 1] = fa_fp8x2_to_h2(wb[m], true); }
                } else {
                    ha[0] = pva[j].x; ha[1] = pva[j].y; ha[2] = pva[j].z; ha[3] = pva[j].w; hb[0] = pvb[j].x; hb[1] = pvb[j].y; hb[2] = pvb[j].z; hb[3] = pvb[j].w;
                }
                char* base = Vt + (size_t)(dc * (FP8 ? 16 : 8)) * LDV + pp * 4;
#pragma unroll
                for (int m = 0; m < NW; m++) {            // dims 2m, 2m+1 of the chunk: {row a, row b} -> one dword each
                    *reinterpret_cast<uint32_t*>(base + (2 * m) * LDV) = __builtin_amdgcn_perm(hb[m], ha[m], 0x05040100u);
                    *reinterpret_cast<uint32_t*>(base + (2 * m + 1) * LDV) = __builtin_amdgcn_perm(hb[m], ha[m], 0x07060302u);
                }
            }
        }
    };

    // per tile: [K -> LDS] barrier [V loads in flight | S^T, softmax] [V -> LDS, next K loads in flight] barrier [O^T += V^T P^T]
    // (only one of the K / V register sets is live at a time: 512 registers hold Q^T, O^T, S^T and one prefetch set without spilling)
    load_k(0);
    for (int tile = 0; tile < n_tiles; tile++) {
        const int p0 = tile * FA_TK;
        commit_k();
        __syncthreads();
        load_v(p0);
        // ---- S^T = K Q^T  (two 32-position blocks)
        v16f sacc[2];
#pragma unroll
        for (int pb = 0; pb < 2; pb++) {
#pragma unroll
            for (int i = 0; i < 16; i++) sacc[pb][i] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks++) {
                const v8h kf = *reinterpret_cast<const v8h*>(Ks + (32 * pb + n31) * LDK + (16 * ks + 8 * khalf) * 2);
                sacc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], sacc[pb], 0, 0, 0);
                if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // keep the LDS fragment loads a few MFMAs ahead, not a whole tile (registers)
            }
        }
        // ---- online softmax of this lane's row (32 of the tile's 64 positions live here, the rest in lane ^ 32)
        const bool need_mask = p0 + FA_TK - 1 > full_vis;
        float mloc = -__builtin_inff();
#pragma unroll
        for (int pb = 0; pb < 2; pb++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (need_mask) { const int p = p0 + 32 * pb + (i & 3) + 8 * (i >> 2) + 4 * khalf; if (p > p_q || !row_ok) sacc[pb][i] = -__builtin_inff(); }
                mloc = fmaxf(mloc, sacc[pb][i]);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);
        const float m_use = m_new == -__builtin_inff() ? 0.0f : m_new;                   // nothing visible yet: exp2(-inf - 0) = 0 everywhere
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);                        // m_run = -inf -> 0
        float lsum = 0.0f;
        v8h pf[2][2];
#pragma unroll
        for (int pb = 0; pb < 2; pb++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float pv = __builtin_amdgcn_exp2f(sacc[pb][i] - m_use);
                lsum += pv;
                pf[pb][i >> 3][i & 7] = (_Float16)pv;
            }
        lsum += __shfl_xor(lsum, 32);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int db = 0; db < DB; db++)
#pragma unroll
                for (int i = 0; i < 16; i++) oacc[db][i] *= alpha;
        }
        commit_v();
        if (tile + 1 < n_tiles) load_k(p0 + FA_TK);
        __syncthreads();
        // ---- O^T += V^T P^T
#pragma unroll
        for (int db = 0; db < DB; db++)
#pragma unroll
            for (int kt = 0; kt < 4; kt++) {
                const char* vr = Vt + (size_t)(32 * db + n31) * LDV + (16 * kt + 4 * khalf) * 2;
                const u32x2 v0 = *reinterpret_cast<const u32x2*>(vr), v1 = *reinterpret_cast<const u32x2*>(vr + 16);
                const u32x4 vv = {v0.x, v0.y, v1.x, v1.y};
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, vv), pf[kt >> 1][kt & 1], oacc[db], 0, 0, 0);
                if (kt == 3) __builtin_amdgcn_sched_barrier(0);
            }
    }
    // ---- normalise, gate (attention.py:664-666 / decode.rs:4272-4280), store: accumulator rows 4g .. 4g+3 are dims 32 db + 8 g + 4 khalf + 0..3
    if (row_ok) {
        const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
        const size_t ob = ((size_t)tok * a.nh + h) * HD;
#pragma unroll
        for (int db = 0; db < DB; db++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const int d = 32 * db + 8 * g4 + 4 * khalf;
                float4 o = make_float4(oacc[db][4 * g4] * inv, oacc[db][4 * g4 + 1] * inv, oacc[db][4 * g4 + 2] * inv, oacc[db][4 * g4 + 3] * inv);
                if (a.gated) {
                    const float4 gt = *reinterpret_cast<const float4*>(a.gate + ob + d);
                    o.x *= 1.0f / (1.0f + kr_expf(-gt.x)); o.y *= 1.0f / (1.0f + kr_expf(-gt.y)); o.z *= 1.0f / (1.0f + kr_expf(-gt.z)); o.w *= 1.0f / (1.0f + kr_expf(-gt.w));
                }
                *reinterpret_cast<float4*>(a.attn_out + ob + d) = o;
            }
    }
}

// non-zero = geometry not covered (the caller falls back to the exact passes)
int kr_launch_pfm_gqa_flash(const KrPfmGqaArgs& a, int C, hipStream_t st) {
    const int G = a.nkv > 0 ? a.nh / a.nkv : 0;
    if (a.nh % a.nkv || G < 1 || G > FA_ROWS || (FA_ROWS % G) || (a.hd != 64 && a.hd != 128 && a.hd != 256)) return 1;
    const int TQ = FA_ROWS / G;
    const size_t lds = (size_t)FA_TK * (a.hd * 2 + 16) + (size_t)a.hd * (FA_TK * 2 + 16);
    {
        const void* fn = a.hd == 256 ? (a.kv_fp8 ? (const void*)kr_pfm_gqa_flash_kernel<256, true> : (const void*)kr_pfm_gqa_flash_kernel<256, false>)
                       : a.hd == 128 ? (a.kv_fp8 ? (const void*)kr_pfm_gqa_flash_kernel<128, true> : (const void*)kr_pfm_gqa_flash_kernel<128, false>)
                                     : (a.kv_fp8 ? (const void*)kr_pfm_gqa_flash_kernel<64, true> : (const void*)kr_pfm_gqa_flash_kernel<64, false>);
        if (kr_lds_optin(fn, 96 * 1024)) return 1;
    }
    dim3 grid((C + TQ - 1) / TQ, a.nkv);
#define KR_FA(H_, F_) hipLaunchKernelGGL((kr_pfm_gqa_flash_kernel<H_, F_>), grid, dim3(256), lds, st, a, C)
    if (a.hd == 256) { if (a.kv_fp8) KR_FA(256, true); else KR_FA(256, false); }
    else if (a.hd == 128) { if (a.kv_fp8) KR_FA(128, true); else KR_FA(128, false); }
    else { if (a.kv_fp8) KR_FA(64, true); else KR_FA(64, false); }
#undef KR_FA
    return 0;
}
