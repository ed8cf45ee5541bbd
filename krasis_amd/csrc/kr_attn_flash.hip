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
#include "kr_decode_ops.h"

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define FA_TK 64
#define FA_ROWS 128
#ifdef KR_TIMING   // tools/probes/flash_timing.hip: wall-clock stamps (10 ns units) of wave 0 of workgroup (0, 0) at one tile; no-op in the product build
__device__ unsigned long long kr_fstamps[32];
#define FA_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1 && blockIdx.y == 0 && tile == 40) kr_fstamps[i] = wall_clock64(); } while (0)
#else
#define FA_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ uint32_t fa_fp8x2_to_h2(uint32_t w, bool hi) {      // two E4M3 bytes -> packed f16 pair (exact): ONE v_cvt_scalef32_pk_f16_fp8 (scale 1)
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t h = hi ? __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, true) : __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, false);
    return __builtin_bit_cast(uint32_t, h);
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
    // E4M3 caches (16 + 16 prefetch registers) keep BOTH sets in flight: K and V of tile t + 1 are requested as soon as tile t went to LDS, a
    // whole tile ahead of their use.  FP16 caches (32 + 32) would spill: there only one set is live (V of the current tile, then K of the next).
    load_k(0);
    if (FP8) load_v(0);
    for (int tile = 0; tile < n_tiles; tile++) {
        const int p0 = tile * FA_TK;
        FA_STAMP(0);
        commit_k();
        if (FP8 && tile + 1 < n_tiles) load_k(p0 + FA_TK);
        __syncthreads();
        if (!FP8) load_v(p0);
        FA_STAMP(1);
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
        FA_STAMP(2);
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
        FA_STAMP(3);
        commit_v();
        if (tile + 1 < n_tiles) { if (FP8) load_v(p0 + FA_TK); else load_k(p0 + FA_TK); }
        FA_STAMP(4);
        __syncthreads();
        FA_STAMP(5);
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
        FA_STAMP(6);
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

// ------------------------------------------------------------------------------------------------------------------------------------
// Eight-wave form (head_dim 128 / 256): the phase probe of the four-wave kernel (tools/probes/flash_timing.hip) shows S^T 0.9 / softmax 1.1 /
// staging 1.2 / PV 1.1 us per 64-position tile, strictly one after the other -- with 471 registers there is ONE wave per SIMD and nothing
// overlaps the matrix core with the vector ALU.  Here the query tile lives in LDS instead of registers and the work of a tile is split over
// wave PAIRS by position: wave (rg, ph) takes query rows [32 rg, +32) and the positions [32 ph, +32) of every tile, with its OWN online-softmax
// state (max, sum, O) -- two independent flash streams over disjoint position subsets, merged once at the end (the split-KV merge, inside the
// workgroup).  Half the MFMAs, half the softmax and half the registers per wave: two waves per SIMD, the one's softmax under the other's MFMAs.
// ------------------------------------------------------------------------------------------------------------------------------------
template <int HD, bool FP8>
__global__ void __launch_bounds__(512, 2) kr_pfm_gqa_flash8_kernel(const KrPfmGqaArgs a, int C) {
    constexpr int KSTEPS = HD / 16, DB = HD / 32, LDK = HD * 2 + 16, LDV = FA_TK * 2 + 16;
    constexpr int CPR = FP8 ? HD / 16 : HD / 8;                 // 16-byte global chunks per cache row
    constexpr int KCH = FA_TK * CPR / 512;                      // K chunks per thread per tile
    constexpr int VUN = (FA_TK / 2) * CPR, VPT = (VUN + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) char fa_smem[];
    char* Qs = fa_smem;                                         // [128 rows][LDK]       f16, scale and log2 e folded in
    char* Ks = Qs + FA_ROWS * LDK;                              // [64 positions][LDK]   f16
    char* Vt = Ks + FA_TK * LDK;                                // [HD dims][LDV]        f16, positions contiguous
    const int G = a.nh / a.nkv, TQ = FA_ROWS / G;
    const int kvh = blockIdx.y, t0 = blockIdx.x * TQ;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n31 = lane & 31, khalf = lane >> 5, rg = wave & 3, ph = wave >> 2;
    const int r = rg * 32 + n31, hl = r / TQ, ti = r % TQ, tok = t0 + ti;
    const bool row_ok = tok < C;
    const int h = kvh * G + hl, p_q = a.pos0 + tok;
    const int kvs = a.nkv * HD, esz = FP8 ? 1 : 2;
    const int kv_end = a.pos0 + (t0 + TQ < C ? t0 + TQ : C);
    const int n_tiles = (kv_end + FA_TK - 1) / FA_TK;
    const int full_vis = a.pos0 + t0;
    // ---- query tile -> LDS
    {
        const float sc = a.sm_scale * 1.4426950408889634f;
        for (int i = tid; i < FA_ROWS * (HD / 8); i += 512) {
            const int rr = i / (HD / 8), c8 = i % (HD / 8), hh = rr / TQ, tt = t0 + rr % TQ;
            u32x4 o = {0, 0, 0, 0};
            if (tt < C) {
                const float* src = a.q_out + ((size_t)tt * a.nh + kvh * G + hh) * HD + c8 * 8;
                const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
                o = u32x4{__builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(x0.x * sc, x0.y * sc)), __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(x0.z * sc, x0.w * sc)),
                          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(x1.x * sc, x1.y * sc)), __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(x1.z * sc, x1.w * sc))};
            }
            *reinterpret_cast<u32x4*>(Qs + rr * LDK + c8 * 16) = o;
        }
    }
    v16f oacc[DB];
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
        for (int i = 0; i < 16; i++) oacc[db][i] = 0.0f;
    float m_run = -__builtin_inff(), l_run = 0.0f;
    const unsigned char* kc = reinterpret_cast<const unsigned char*>(a.k_cache) + (size_t)kvh * HD * esz;
    const unsigned char* vc = reinterpret_cast<const unsigned char*>(a.v_cache) + (size_t)kvh * HD * esz;
    u32x4 pk[KCH], pva[VPT], pvb[VPT];
    auto load_k = [&](int p0) {
#pragma unroll
        for (int j = 0; j < KCH; j++) {
            const int c = tid + j * 512, row = c / CPR, dc = c % CPR, p = min(p0 + row, kv_end - 1);      // rows past the end are masked below
            pk[j] = *reinterpret_cast<const u32x4*>(kc + (size_t)p * kvs * esz + dc * 16);
        }
    };
    auto load_v = [&](int p0) {
#pragma unroll
        for (int j = 0; j < VPT; j++) {
            const int u = tid + j * 512, pp = u & 31, dc = u >> 5, p = p0 + 2 * pp;
            pva[j] = u32x4{0, 0, 0, 0}; pvb[j] = u32x4{0, 0, 0, 0};
            if (u < VUN) {
                const u32x4 va = *reinterpret_cast<const u32x4*>(vc + (size_t)min(p, kv_end - 1) * kvs * esz + dc * 16);
                const u32x4 vb = *reinterpret_cast<const u32x4*>(vc + (size_t)min(p + 1, kv_end - 1) * kvs * esz + dc * 16);
                if (p < kv_end) pva[j] = va;
                if (p + 1 < kv_end) pvb[j] = vb;
            }
        }
    };
    auto commit_k = [&]() {
#pragma unroll
        for (int j = 0; j < KCH; j++) {
            const int c = tid + j * 512, row = c / CPR, dc = c % CPR;
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
            const int u = tid + j * 512, pp = u & 31, dc = u >> 5;
            if (u < VUN) {
                uint32_t ha[8], hb[8];
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
                for (int m = 0; m < NW; m++) {
                    *reinterpret_cast<uint32_t*>(base + (2 * m) * LDV) = __builtin_amdgcn_perm(hb[m], ha[m], 0x05040100u);
                    *reinterpret_cast<uint32_t*>(base + (2 * m + 1) * LDV) = __builtin_amdgcn_perm(hb[m], ha[m], 0x07060302u);
                }
            }
        }
    };
    load_k(0); load_v(0);
    for (int tile = 0; tile < n_tiles; tile++) {
        const int p0 = tile * FA_TK;
        commit_k();
        if (tile + 1 < n_tiles) load_k(p0 + FA_TK);
        __syncthreads();                                  // K tile (and, in the first round, the query tile) is complete
        // ---- S^T of this wave's 32 positions
        v16f sacc;
#pragma unroll
        for (int i = 0; i < 16; i++) sacc[i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
            const v8h kf = *reinterpret_cast<const v8h*>(Ks + (32 * ph + n31) * LDK + (16 * ks + 8 * khalf) * 2);
            const v8h qf = *reinterpret_cast<const v8h*>(Qs + r * LDK + (16 * ks + 8 * khalf) * 2);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf, sacc, 0, 0, 0);
        }
        const bool need_mask = p0 + FA_TK - 1 > full_vis || p0 + FA_TK > kv_end;
        float mloc = -__builtin_inff();
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (need_mask) { const int p = p0 + 32 * ph + (i & 3) + 8 * (i >> 2) + 4 * khalf; if (p > p_q || !row_ok) sacc[i] = -__builtin_inff(); }
            mloc = fmaxf(mloc, sacc[i]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);
        const float m_use = m_new == -__builtin_inff() ? 0.0f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        float lsum = 0.0f;
        v8h pf[2];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float pv = __builtin_amdgcn_exp2f(sacc[i] - m_use);
            lsum += pv;
            pf[i >> 3][i & 7] = (_Float16)pv;
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
        if (tile + 1 < n_tiles) load_v(p0 + FA_TK);
        __syncthreads();
        // ---- O^T += V^T P^T over this wave's 32 positions
#pragma unroll
        for (int db = 0; db < DB; db++)
#pragma unroll
            for (int kt = 0; kt < 2; kt++) {
                const char* vr = Vt + (size_t)(32 * db + n31) * LDV + (32 * ph + 16 * kt + 4 * khalf) * 2;
                const u32x2 v0 = *reinterpret_cast<const u32x2*>(vr), v1 = *reinterpret_cast<const u32x2*>(vr + 16);
                const u32x4 vv = {v0.x, v0.y, v1.x, v1.y};
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, vv), pf[kt], oacc[db], 0, 0, 0);
            }
    }
    // ---- merge the two position streams of a row group: the ph = 1 waves hand (m, l, O) over through LDS (everything staged there is dead now)
    __syncthreads();
    float* Ox = reinterpret_cast<float*>(fa_smem);              // [4 row groups][HD dims][32 rows] f32 + [4][2][32] (m, l)   (HD = 256: 128 KiB + 1 KiB)
    float* MLx = Ox + 4 * HD * 32;
    if (ph == 1) {
#pragma unroll
        for (int db = 0; db < DB; db++)
#pragma unroll
            for (int i = 0; i < 16; i++) Ox[((size_t)rg * HD + 32 * db + (i & 3) + 8 * (i >> 2) + 4 * khalf) * 32 + n31] = oacc[db][i];
        if (khalf == 0) { MLx[(rg * 2 + 0) * 32 + n31] = m_run; MLx[(rg * 2 + 1) * 32 + n31] = l_run; }
    }
    __syncthreads();
    if (ph == 0 && row_ok) {
        const float m1 = MLx[(rg * 2 + 0) * 32 + n31], l1 = MLx[(rg * 2 + 1) * 32 + n31];
        const float M = fmaxf(m_run, m1), Mu = M == -__builtin_inff() ? 0.0f : M;
        const float w0 = __builtin_amdgcn_exp2f(m_run - Mu), w1 = __builtin_amdgcn_exp2f(m1 - Mu);
        const float lt = l_run * w0 + l1 * w1;
        const float inv = lt > 0.0f ? 1.0f / lt : 0.0f;
        const size_t ob = ((size_t)tok * a.nh + h) * HD;
#pragma unroll
        for (int db = 0; db < DB; db++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const int d = 32 * db + 8 * g4 + 4 * khalf;
                float o4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) o4[u] = (oacc[db][4 * g4 + u] * w0 + Ox[((size_t)rg * HD + d + u) * 32 + n31] * w1) * inv;
                float4 o = make_float4(o4[0], o4[1], o4[2], o4[3]);
                if (a.gated) {
                    const float4 gt = *reinterpret_cast<const float4*>(a.gate + ob + d);
                    o.x *= 1.0f / (1.0f + kr_expf(-gt.x)); o.y *= 1.0f / (1.0f + kr_expf(-gt.y)); o.z *= 1.0f / (1.0f + kr_expf(-gt.z)); o.w *= 1.0f / (1.0f + kr_expf(-gt.w));
                }
                *reinterpret_cast<float4*>(a.attn_out + ob + d) = o;
            }
    }
}

// non-zero = geometry not covered (the caller falls back to the exact passes)
bool kr_pfm_gqa_flash_ok(int nh, int nkv, int hd) {
    const int G = nkv > 0 ? nh / nkv : 0;
    return nkv > 0 && nh % nkv == 0 && G >= 1 && G <= FA_ROWS && (FA_ROWS % G) == 0 && (hd == 64 || hd == 128 || hd == 256);
}

int kr_launch_pfm_gqa_flash(const KrPfmGqaArgs& a, int C, hipStream_t st) {
    const int G = a.nkv > 0 ? a.nh / a.nkv : 0;
    if (!kr_pfm_gqa_flash_ok(a.nh, a.nkv, a.hd)) return 1;
    const int TQ = FA_ROWS / G;
    dim3 grid((C + TQ - 1) / TQ, a.nkv);
    if (a.hd >= 128) {      // eight waves, query tile in LDS, wave pairs split the positions of a tile (the four-wave form at these head sizes needed 512 registers + scratch: removed)
        const size_t l8 = (size_t)(FA_ROWS + FA_TK) * (a.hd * 2 + 16) + (size_t)a.hd * (FA_TK * 2 + 16);
        const size_t lm = (size_t)(4 * a.hd * 32 + 4 * 2 * 32) * 4;
        const size_t lds8 = l8 > lm ? l8 : lm;
        const void* fn8 = a.hd == 256 ? (a.kv_fp8 ? (const void*)kr_pfm_gqa_flash8_kernel<256, true> : (const void*)kr_pfm_gqa_flash8_kernel<256, false>)
                                      : (a.kv_fp8 ? (const void*)kr_pfm_gqa_flash8_kernel<128, true> : (const void*)kr_pfm_gqa_flash8_kernel<128, false>);
        if (lds8 <= 160 * 1024 && kr_lds_optin(fn8, lds8) == 0) {
#define KR_FA8(H_, F_) hipLaunchKernelGGL((kr_pfm_gqa_flash8_kernel<H_, F_>), grid, dim3(512), lds8, st, a, C)
            if (a.hd == 256) { if (a.kv_fp8) KR_FA8(256, true); else KR_FA8(256, false); }
            else { if (a.kv_fp8) KR_FA8(128, true); else KR_FA8(128, false); }
#undef KR_FA8
            return 0;
        }
        return 1;           // LDS window refused: the caller takes the exact passes
    }
    const size_t lds = (size_t)FA_TK * (a.hd * 2 + 16) + (size_t)a.hd * (FA_TK * 2 + 16);
    {
        const void* fn = a.kv_fp8 ? (const void*)kr_pfm_gqa_flash_kernel<64, true> : (const void*)kr_pfm_gqa_flash_kernel<64, false>;
        if (kr_lds_optin(fn, 96 * 1024)) return 1;
    }
#define KR_FA(H_, F_) hipLaunchKernelGGL((kr_pfm_gqa_flash_kernel<H_, F_>), grid, dim3(256), lds, st, a, C)
    if (a.kv_fp8) KR_FA(64, true); else KR_FA(64, false);      // head_dim 64: the four-wave form
#undef KR_FA
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// FAST decode attention over a long cache (one token): split-KV flash-decode on the same MFMA forms.
//   kr_fd_flash_kernel  grid (chunks, KV heads): the workgroup stages its chunk's K and V^T tiles (64 positions at a time) ONCE for all G query
//                       heads of the group.  The G heads are accumulator columns 0..G-1 (the other columns of the 32-wide block carry q = 0 and
//                       are never stored: the launch is bound by the K / V stream, not by the MFMAs).  Every wave forms the same S^T and softmax
//                       statistics (32 MFMAs per tile -- cheaper than an exchange) and owns HD / 128 of the O^T dimension blocks.  Output: the
//                       chunk's un-normalised O [G][HD] and (max, sum) per head, max in log2 units.
//   kr_fd_merge2_kernel one workgroup per head: log-sum-exp merge of the chunk partials with independent partial sums (the first version's
//                       single dependent fma per chunk made this launch as long as the partial launch), gate, o-projection image.
// This replaces BOTH the exact scores launch and the first split-KV form (kr_attn_fd.h: exact scores + VALU p.v): at position 32 766 of an
// FP8 cache those were 34 + 36 + 38 us per GQA layer (profiles/r02_decode_32k_fast_fp8_kernel_stats.txt).
// ------------------------------------------------------------------------------------------------------------------------------------
template <int HD, bool FP8>
__global__ void __launch_bounds__(256) kr_fd_flash_kernel(const KrFdFlashArgs a, int n_chunks, int chunk) {
    constexpr int KSTEPS = HD / 16, DB = HD / 32, DBW = DB >= 4 ? DB / 4 : 1, LDK = HD * 2 + 16, LDV = FA_TK * 2 + 16;
    constexpr int CPR = FP8 ? HD / 16 : HD / 8, KCH = FA_TK * CPR / 256, VUN = (FA_TK / 2) * CPR, VPT = (VUN + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char fa_smem[];
    char* Ks = fa_smem; char* Vt = fa_smem + FA_TK * LDK;
    const int c = blockIdx.x, kvh = blockIdx.y, G = a.nh / a.nkv;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n31 = lane & 31, khalf = lane >> 5;
    const int seq = a.step->pos + 1, pbeg = c * chunk;
    if (pbeg >= seq) return;
    const int pend = min(seq, pbeg + chunk), n_tiles = (pend - pbeg + FA_TK - 1) / FA_TK;
    const bool row_ok = n31 < G;
    const int h = kvh * G + (row_ok ? n31 : 0);
    const int kvs = a.nkv * HD, esz = FP8 ? 1 : 2;
    v8h qf[KSTEPS];
    {
        const float* q = a.q + (size_t)h * HD + 8 * khalf;
        const float sc = a.sm_scale * 1.4426950408889634f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
            const float4 x0 = *reinterpret_cast<const float4*>(q + 16 * ks), x1 = *reinterpret_cast<const float4*>(q + 16 * ks + 4);
            const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int i = 0; i < 8; i++) qf[ks][i] = (_Float16)(row_ok ? xv[i] * sc : 0.0f);
        }
    }
    v16f oacc[DBW];
#pragma unroll
    for (int db = 0; db < DBW; db++)
#pragma unroll
        for (int i = 0; i < 16; i++) oacc[db][i] = 0.0f;
    float m_run = -__builtin_inff(), l_run = 0.0f;
    const unsigned char* kc = reinterpret_cast<const unsigned char*>(a.k_cache) + (size_t)kvh * HD * esz;
    const unsigned char* vc = reinterpret_cast<const unsigned char*>(a.v_cache) + (size_t)kvh * HD * esz;
    u32x4 pk[KCH], pva[VPT], pvb[VPT];
    auto load_k = [&](int p0) {
#pragma unroll
        for (int j = 0; j < KCH; j++) {
            const int cc = tid + j * 256, row = cc / CPR, dc = cc % CPR, p = min(p0 + row, pend - 1);     // clamped: rows past the end are masked below
            pk[j] = *reinterpret_cast<const u32x4*>(kc + (size_t)p * kvs * esz + dc * 16);
        }
    };
    auto load_v = [&](int p0) {
#pragma unroll
        for (int j = 0; j < VPT; j++) {
            const int u = tid + j * 256, pp = u & 31, dc = u >> 5, p = p0 + 2 * pp;
            pva[j] = u32x4{0, 0, 0, 0}; pvb[j] = u32x4{0, 0, 0, 0};
            if (u < VUN) {
                const u32x4 va = *reinterpret_cast<const u32x4*>(vc + (size_t)min(p, pend - 1) * kvs * esz + dc * 16);
                const u32x4 vb = *reinterpret_cast<const u32x4*>(vc + (size_t)min(p + 1, pend - 1) * kvs * esz + dc * 16);
                if (p < pend) pva[j] = va;                 // a probability of exactly 0 meets a finite value: rows past the end must not be NaN patterns
                if (p + 1 < pend) pvb[j] = vb;
            }
        }
    };
    auto commit_k = [&]() {
#pragma unroll
        for (int j = 0; j < KCH; j++) {
            const int cc = tid + j * 256, row = cc / CPR, dc = cc % CPR;
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
                uint32_t ha[8], hb[8];
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
                for (int m = 0; m < NW; m++) {
                    *reinterpret_cast<uint32_t*>(base + (2 * m) * LDV) = __builtin_amdgcn_perm(hb[m], ha[m], 0x05040100u);
                    *reinterpret_cast<uint32_t*>(base + (2 * m + 1) * LDV) = __builtin_amdgcn_perm(hb[m], ha[m], 0x07060302u);
                }
            }
        }
    };
    // K and V of tile t + 1 are requested while tile t is consumed (each register set is free as soon as its tile went to LDS; a second
    // register set -- requests two tiles ahead -- was measured: 279 registers, one workgroup per CU, 18.4 -> 26.5 us per launch)
    load_k(pbeg); load_v(pbeg);
    for (int tile = 0; tile < n_tiles; tile++) {
        const int p0 = pbeg + tile * FA_TK;
        commit_k();
        if (tile + 1 < n_tiles) load_k(p0 + FA_TK);
        __syncthreads();
        v16f sacc[2];
#pragma unroll
        for (int pb = 0; pb < 2; pb++) {
#pragma unroll
            for (int i = 0; i < 16; i++) sacc[pb][i] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks++) {
                const v8h kf = *reinterpret_cast<const v8h*>(Ks + (32 * pb + n31) * LDK + (16 * ks + 8 * khalf) * 2);
                sacc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], sacc[pb], 0, 0, 0);
                if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        const bool need_mask = p0 + FA_TK > pend;
        float mloc = -__builtin_inff();
#pragma unroll
        for (int pb = 0; pb < 2; pb++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (need_mask) { const int p = p0 + 32 * pb + (i & 3) + 8 * (i >> 2) + 4 * khalf; if (p >= pend) sacc[pb][i] = -__builtin_inff(); }
                mloc = fmaxf(mloc, sacc[pb][i]);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);                                           // finite: every tile holds at least one visible position
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float lsum = 0.0f;
        v8h pf[2][2];
#pragma unroll
        for (int pb = 0; pb < 2; pb++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float pv = __builtin_amdgcn_exp2f(sacc[pb][i] - m_new);
                lsum += pv;
                pf[pb][i >> 3][i & 7] = (_Float16)pv;
            }
        lsum += __shfl_xor(lsum, 32);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < DBW; db++)
#pragma unroll
            for (int i = 0; i < 16; i++) oacc[db][i] *= alpha;
        commit_v();
        if (tile + 1 < n_tiles) load_v(p0 + FA_TK);
        __syncthreads();
        if (wave * DBW < DB) {
#pragma unroll
            for (int db = 0; db < DBW; db++)
#pragma unroll
                for (int kt = 0; kt < 4; kt++) {
                    const char* vr = Vt + (size_t)(32 * (wave * DBW + db) + n31) * LDV + (16 * kt + 4 * khalf) * 2;
                    const u32x2 v0 = *reinterpret_cast<const u32x2*>(vr), v1 = *reinterpret_cast<const u32x2*>(vr + 16);
                    const u32x4 vv = {v0.x, v0.y, v1.x, v1.y};
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, vv), pf[kt >> 1][kt & 1], oacc[db], 0, 0, 0);
                }
        }
    }
    // ---- partials: accumulator rows 4 g4 .. 4 g4 + 3 of block db are dims 32 (wave DBW + db) + 8 g4 + 4 khalf + 0..3, column = head n31
    if (row_ok) {
        float* po = a.fd_o + (((size_t)kvh * n_chunks + c) * G + n31) * HD;
        if (wave * DBW < DB) {
#pragma unroll
            for (int db = 0; db < DBW; db++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++)
                    *reinterpret_cast<float4*>(po + 32 * (wave * DBW + db) + 8 * g4 + 4 * khalf) =
                        make_float4(oacc[db][4 * g4], oacc[db][4 * g4 + 1], oacc[db][4 * g4 + 2], oacc[db][4 * g4 + 3]);
        }
        if (wave == 0 && khalf == 0) { float* ml = a.fd_ml + ((size_t)(kvh * G + n31) * n_chunks + c) * 2; ml[0] = m_run; ml[1] = l_run; }
    }
}

template <int HD>
__global__ void __launch_bounds__(1024) kr_fd_merge2_kernel(const KrFdFlashArgs a, int n_chunks, int chunk) {
    __shared__ float wc[1024]; __shared__ float qs[1024]; __shared__ float red[32];
    const int h = blockIdx.x, t = threadIdx.x, G = a.nh / a.nkv, kvh = h / G, g = h % G;
    const int seq = a.step->pos + 1, nc = min((seq + chunk - 1) / chunk, 1024);
    const float* ml = a.fd_ml + (size_t)h * n_chunks * 2;
    float mx = -__builtin_inff();
    const float mv = t < nc ? ml[t * 2] : -__builtin_inff(), lv = t < nc ? ml[t * 2 + 1] : 0.0f;      // nc <= 1024 = one chunk per thread
    mx = mv;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int i = 1; i < 16; i++) mx = fmaxf(mx, red[i]);
    const float wv = t < nc ? __builtin_amdgcn_exp2f(mv - mx) : 0.0f;
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
    // thread = (dim d, chunk phase): 1024 / HD phases walk interleaved chunks with 4 independent sums each
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
        qs[t] = o;                                   // HD * NPH == 1024
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
    if (a.img_out) {
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

int kr_fd_flash_chunk(int max_seq) { return max_seq > 131072 ? 256 : 128; }      // at most 1024 chunks (the merge gives each a thread)
size_t kr_fd_flash_chunks(int max_seq) { const int ch = kr_fd_flash_chunk(max_seq); return ((size_t)max_seq + ch - 1) / ch; }
// outside graph capture (the windows are per (kernel, device))
int kr_fd_flash_prepare(int hd, int fp8) {
    const size_t lds = (size_t)FA_TK * (hd * 2 + 16) + (size_t)hd * (FA_TK * 2 + 16);
    const void* fn = hd == 256 ? (fp8 ? (const void*)kr_fd_flash_kernel<256, true> : (const void*)kr_fd_flash_kernel<256, false>)
                   : hd == 128 ? (fp8 ? (const void*)kr_fd_flash_kernel<128, true> : (const void*)kr_fd_flash_kernel<128, false>)
                               : (fp8 ? (const void*)kr_fd_flash_kernel<64, true> : (const void*)kr_fd_flash_kernel<64, false>);
    return lds > 64 * 1024 ? kr_lds_optin(fn, lds) : 0;
}
// non-zero = geometry not covered
int kr_launch_fd_flash(const KrFdFlashArgs& a, int hd, int fp8, int max_seq, hipStream_t st) {
    const int G = a.nkv > 0 ? a.nh / a.nkv : 0;
    if (a.nh % a.nkv || G < 1 || G > 32 || (hd != 64 && hd != 128 && hd != 256)) return 1;
    const int chunk = kr_fd_flash_chunk(max_seq), nch = (int)kr_fd_flash_chunks(max_seq);
    if (nch > 1024) return 1;
    const size_t lds = (size_t)FA_TK * (hd * 2 + 16) + (size_t)hd * (FA_TK * 2 + 16);
    dim3 grid(nch, a.nkv);
#define KR_FF(H_, F_) hipLaunchKernelGGL((kr_fd_flash_kernel<H_, F_>), grid, dim3(256), lds, st, a, nch, chunk)
    if (hd == 256) { if (fp8) KR_FF(256, true); else KR_FF(256, false); hipLaunchKernelGGL(kr_fd_merge2_kernel<256>, dim3(a.nh), dim3(1024), 0, st, a, nch, chunk); }
    else if (hd == 128) { if (fp8) KR_FF(128, true); else KR_FF(128, false); hipLaunchKernelGGL(kr_fd_merge2_kernel<128>, dim3(a.nh), dim3(1024), 0, st, a, nch, chunk); }
    else { if (fp8) KR_FF(64, true); else KR_FF(64, false); hipLaunchKernelGGL(kr_fd_merge2_kernel<64>, dim3(a.nh), dim3(1024), 0, st, a, nch, chunk); }
#undef KR_FF
    return 0;
}

void kr_launch_fd_merge2(const KrFdFlashArgs& a, int hd, int nch, int chunk, hipStream_t st) {     // also used by the MLA flash-decode (hd = kv_lora_rank)
    if (hd == 512) hipLaunchKernelGGL(kr_fd_merge2_kernel<512>, dim3(a.nh), dim3(1024), 0, st, a, nch, chunk);
    else if (hd == 256) hipLaunchKernelGGL(kr_fd_merge2_kernel<256>, dim3(a.nh), dim3(1024), 0, st, a, nch, chunk);
    else if (hd == 128) hipLaunchKernelGGL(kr_fd_merge2_kernel<128>, dim3(a.nh), dim3(1024), 0, st, a, nch, chunk);
    else hipLaunchKernelGGL(kr_fd_merge2_kernel<64>, dim3(a.nh), dim3(1024), 0, st, a, nch, chunk);
}
