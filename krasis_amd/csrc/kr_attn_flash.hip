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
#include <type_traits>
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
#define FA_STAMP2(i) do { if (threadIdx.x == 256 && blockIdx.x == gridDim.x - 1 && blockIdx.y == 0 && tile == 40) kr_fstamps[i] = wall_clock64(); } while (0)
#else
#define FA_STAMP(i) do { } while (0)
#define FA_STAMP2(i) do { } while (0)
#endif

__device__ __forceinline__ uint32_t fa_fp8x2_to_h2(uint32_t w, bool hi) {      // two E4M3 bytes -> packed f16 pair (exact): ONE v_cvt_scalef32_pk_f16_fp8 (scale 1)
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t h = hi ? __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, true) : __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, false);
    return __builtin_bit_cast(uint32_t, h);
}

// V^T rows in LDS hold the 64 positions of a tile as 32 dwords (position pairs).  Inside every group of 16 positions the pairs are stored in the order
// (0-3, 8-11 | 4-7, 12-15) -- pair index bits 1 and 2 swapped -- so that the 8 positions lane half `khalf` multiplies with its P^T fragment (the S^T
// accumulator's own row order) are ONE 16-byte read; in natural order they were two 8-byte reads whose 144-byte row stride put lanes i and i + 16 on
// the same banks (the P.V phase ran at half the LDS rate).
__device__ __forceinline__ int fa_vslot(int pp) { return (pp & ~6) | ((pp & 2) << 1) | ((pp & 4) >> 1); }

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
                char* base = Vt + (size_t)(dc * (FP8 ? 16 : 8)) * LDV + fa_vslot(pp) * 4;
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
                const v8h vv = *reinterpret_cast<const v8h*>(Vt + (size_t)(32 * db + n31) * LDV + 32 * kt + 16 * khalf);
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vv, pf[kt >> 1][kt & 1], oacc[db], 0, 0, 0);
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
// Wave-specialised form (head_dim 128 / 256).  Round 3 ran these head sizes as eight waves that all did the same thing in lock step (query tile in LDS, wave
// pairs splitting the positions of a tile, a split-KV merge at the end): every S^T MFMA fetched a K AND a Q fragment from LDS (2 KB per 32 x 32 x 16 MFMA),
// the phases of a tile stood one after the other on both waves of a SIMD (staging 0.9, S^T 0.56, softmax 0.44, P.V 0.56 of 2.64 us) and V reached LDS through
// a transposing register pass (16 conversions + 16 byte permutes + 16 four-byte LDS writes per thread and tile).  Here the two waves of a SIMD do DIFFERENT things:
//   waves 0-3 ("S waves")   own 32 query rows each, Q^T in REGISTERS (no Q in LDS at all): S^T over the 64-position tile, the online softmax (the row
//                           state m, l lives here only), P as f16 rows + the row's rescale factor into LDS (double-buffered); they stage K
//   waves 4-7 ("PV waves")  own HD / 4 output dims each for ALL 128 rows (a V^T fragment serves four row blocks): O^T += V^T P^T one tile BEHIND the S
//                           waves, so the softmax arithmetic of tile t runs under the P.V MFMAs of tile t - 1 on the same SIMD; they stage V, ROW-major like
//                           K -- the transposed fragments come out of gfx950's LDS transpose-read (ds_read_b64_tr_b16)
// No split-KV merge at the end: one softmax stream per row.  V is double-buffered: the PV waves stage V(t) FIRST (their matrix work would only queue behind the S
// waves' S^T on the same SIMD) and run P.V(t - 1) while the S waves are in their softmax.  The S^T accumulators start at -m (the row's running maximum), so once
// the maximum has settled no element needs a subtraction before its exponential and the rescale factor is 1; fragments of both sides are requested several
// MFMAs ahead by hand (one wave of a kind per SIMD: nothing else covers an LDS round trip).
// tools/probes/flash_timing.hip (1024 queries late in a 32 k cache, E4M3, 128 of 256 CUs busy): 355 -> 476 TFLOP/s; per 64-position tile S^T 0.72 | softmax
// 0.56 | K staging 0.40 on the S side, V staging 0.88 (with the fragment requests) | P.V 0.92 on the PV side, 2.28 us in all against 1.02 us of pure MFMA time
// at the 2.0 GHz the chip holds under matrix load (tools/probes/clock_probe.hip).  Tried on the way and not kept: a 32-position tile with K, V and P all
// double-buffered and one barrier per tile (same speed: the waves' own instruction streams, not the barriers, are the limit), S^T(t + 1) issued between the
// pieces of the softmax of tile t in the same wave (slower: the K fragment waits came to stand in front of every piece), K double-buffered instead of V
// (433: P.V and S^T collide on the matrix pipe again).
// ------------------------------------------------------------------------------------------------------------------------------------
// one v_max3_f32 (the compiler's fmaxf chain canonicalises MFMA results first: two instructions per value)
__device__ __forceinline__ float fa_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// the value lane ^ 32 holds: one v_permlane32_swap (the ds_bpermute of __shfl_xor is an LDS round trip on the softmax's serial path)
__device__ __forceinline__ float fa_other_half(float x, int khalf) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, khalf ? sw[0] : sw[1]);
}
template <int HD, bool FP8>
__global__ void __launch_bounds__(512, 2) kr_pfm_gqa_flashw_kernel(const KrPfmGqaArgs a, int C) {
    constexpr int KSTEPS = HD / 16, DBW = HD / 128, LDK = HD * 2 + 16, LDP = FA_TK * 2 + 16;
    constexpr int LDV = HD * 2 + 64;                            // V rows [position][dim]: 16 banks between rows, so the four rows x 64 bytes a 32-lane group of a transpose-read touches are disjoint
    constexpr int CPR = HD / 8;                                 // staging chunks per cache row: 8 values each (8 bytes of an E4M3 row, 16 of an FP16 row) -> 16 bytes of f16 in LDS,
    constexpr int NCH = FA_TK * CPR / 256;                      // consecutive lanes on consecutive 16-byte slots (conflict-free writes); chunks per thread and tile
    extern __shared__ __attribute__((aligned(16))) char fa_smem[];
    char* Ks = fa_smem;                                         // [64 positions][LDK]   f16
    char* Vs = Ks + FA_TK * LDK;                                // [2][64 positions][LDV] f16, ROW-major like K: the P.V fragments come out of ds_read_b64_tr_b16
    char* Ps = Vs + 2 * FA_TK * LDV;                                // [2][128 rows][LDP]    f16 probabilities of a tile, positions contiguous
    float* Al = reinterpret_cast<float*>(Ps + 2 * FA_ROWS * LDP);      // [2][128] rescale factor of the row for the tile
    float* Il = Al + 2 * FA_ROWS;                               // [128] 1 / l at the end
    const int G = a.nh / a.nkv, TQ = FA_ROWS / G;
    const int kvh = blockIdx.y, t0 = (gridDim.x - 1 - blockIdx.x) * TQ;      // longest rows first: the tail of the launch is made of short workgroups
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n31 = lane & 31, khalf = lane >> 5;
    const int kvs = a.nkv * HD, esz = FP8 ? 1 : 2;
    const int kv_end = a.pos0 + (t0 + TQ < C ? t0 + TQ : C);
    const int n_tiles = (kv_end + FA_TK - 1) / FA_TK;
    const int full_vis = a.pos0 + t0;
    // ---- staging of a K (S waves) or V (PV waves) tile by the 256 threads of a side: global -> registers (a tile ahead) -> f16 rows in LDS
    typedef typename std::conditional<FP8, u32x2, u32x4>::type chunk_t;
    const int stid = tid & 255, srow = stid / CPR, sdc = stid % CPR;
    constexpr int RSTEP = 256 / CPR;
    chunk_t stg[NCH];
    const unsigned char* cache = reinterpret_cast<const unsigned char*>(wave < 4 ? a.k_cache : a.v_cache) + (size_t)kvh * HD * esz + sdc * (FP8 ? 8 : 16);
    const size_t rowb = (size_t)kvs * esz;                       // bytes between cache rows (wave-uniform)
    auto stage_load = [&](int p0, bool zero_past_end) {
        if (p0 + FA_TK <= kv_end) {                             // whole tile inside the cache rows this workgroup may see: one 64-bit multiply per tile, then row steps
            const unsigned char* rp = cache + (size_t)(p0 + srow) * rowb;
#pragma unroll
            for (int j = 0; j < NCH; j++) stg[j] = *reinterpret_cast<const chunk_t*>(rp + (size_t)(RSTEP * j) * rowb);
        } else {
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                const int p = p0 + srow + RSTEP * j;
                const chunk_t v = *reinterpret_cast<const chunk_t*>(cache + (size_t)min(p, kv_end - 1) * rowb);
                stg[j] = (zero_past_end && p >= kv_end) ? chunk_t{} : v;      // K rows past the end are masked by the causal test; V rows meet P = 0 and must be finite
            }
        }
    };
    auto stage_commit = [&](char* dst, int ld) {
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            u32x4 o;
            if constexpr (FP8) o = u32x4{fa_fp8x2_to_h2(stg[j].x, false), fa_fp8x2_to_h2(stg[j].x, true), fa_fp8x2_to_h2(stg[j].y, false), fa_fp8x2_to_h2(stg[j].y, true)};
            else o = stg[j];
            *reinterpret_cast<u32x4*>(dst + (srow + RSTEP * j) * ld + sdc * 16) = o;
        }
    };
    if (wave < 4) {
        // ================================================================ S waves
        const int r = wave * 32 + n31, hl = r / TQ, ti = r % TQ, tok = t0 + ti;
        const bool row_ok = tok < C;
        const int h = kvh * G + hl, p_q = a.pos0 + tok;
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
        float m_run = 0.0f, l_run = 0.0f;                      // m_run: the origin the row's scores are measured from (its running maximum from the first tile on)
        const char* krow = Ks + n31 * LDK + 16 * khalf;
        stage_load(0, false);
        stage_commit(Ks, LDK);
        if (n_tiles > 1) stage_load(FA_TK, false);
        for (int tile = 0; tile <= n_tiles; tile++) {
            FA_STAMP(0);
            __syncthreads();                                  // X: K(tile) is in LDS (and, for the other side, V(tile - 1) and P(tile - 1))
            FA_STAMP(1);
            if (tile < n_tiles) {
                const int p0 = tile * FA_TK;
                v16f sacc[2];
#pragma unroll
                for (int pb = 0; pb < 2; pb++)
#pragma unroll
                    for (int i = 0; i < 16; i++) sacc[pb][i] = -m_run;
                // K fragments through a ring of FW_RING registers sets: with one wave of this kind per SIMD an LDS round trip (a few hundred cycles while the
                // other side writes its tile) has to be covered by this wave's own MFMAs -- the compiler's schedule kept ~2 in flight and S^T ran at 0.58 of the pipe
                constexpr int NF = 2 * KSTEPS, RING = 8;
                v8h kf[RING];
                auto kaddr = [&](int f) { return krow + 32 * (f & 1) * LDK + 32 * (f >> 1); };      // fragment f: position block f & 1, k-step f >> 1 (the two accumulators alternate)
#pragma unroll
                for (int f = 0; f < RING; f++) kf[f] = *reinterpret_cast<const v8h*>(kaddr(f));
#pragma unroll
                for (int f = 0; f < NF; f++) {
                    sacc[f & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[f % RING], qf[f >> 1], sacc[f & 1], 0, 0, 0);
                    if (f + RING < NF) kf[f % RING] = *reinterpret_cast<const v8h*>(kaddr(f + RING));
                    if ((f & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
                FA_STAMP(2);
                // The accumulators were started at -m (the row's running maximum, 0 before the first tile), so sacc = s - m already.  Once the maximum has settled
                // -- almost every tile of a long prompt -- no element needs another subtraction and the rescale factor is 1.
                const bool need_mask = p0 + FA_TK - 1 > full_vis || p0 + FA_TK > kv_end;
                if (need_mask) {
#pragma unroll
                    for (int pb = 0; pb < 2; pb++)
#pragma unroll
                        for (int i = 0; i < 16; i++) { const int p = p0 + 32 * pb + (i & 3) + 8 * (i >> 2) + 4 * khalf; if (p > p_q || !row_ok) sacc[pb][i] = -__builtin_inff(); }
                }
                float mloc = fa_max3(sacc[0][0], sacc[1][0], sacc[0][1]);
#pragma unroll
                for (int i = 1; i < 16; i++) mloc = fa_max3(mloc, sacc[0][i], sacc[1][i]);      // (element [0][1] twice: harmless)
                mloc = fa_max3(mloc, fa_other_half(mloc, khalf), tile == 0 ? -__builtin_inff() : 0.0f);      // the first tile sets the maximum, later ones only raise it
                const float shift = mloc == -__builtin_inff() ? 0.0f : mloc;       // >= 0 after the first tile; a row that has seen nothing keeps its origin
                const float alpha = __builtin_amdgcn_exp2f(-shift);
                float lsum = 0.0f;
                char* prow = Ps + (size_t)(tile & 1) * FA_ROWS * LDP + (size_t)r * LDP + 8 * khalf;
                if (__any(shift != 0.0f)) {
#pragma unroll
                    for (int pb = 0; pb < 2; pb++)
#pragma unroll
                        for (int i = 0; i < 16; i++) sacc[pb][i] -= shift;
                }
#pragma unroll
                for (int pb = 0; pb < 2; pb++)
#pragma unroll
                    for (int g4 = 0; g4 < 4; g4++) {      // accumulator rows 4 g4 .. + 3 = positions 32 pb + 8 g4 + 4 khalf .. + 3: one 8-byte store
                        float pv[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) { pv[u] = __builtin_amdgcn_exp2f(sacc[pb][4 * g4 + u]); lsum += pv[u]; }
                        const u32x2 w = {__builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(pv[0], pv[1])), __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(pv[2], pv[3]))};
                        *reinterpret_cast<u32x2*>(prow + (32 * pb + 8 * g4) * 2) = w;
                    }
                lsum += fa_other_half(lsum, khalf);
                l_run = l_run * alpha + lsum;
                m_run += shift;
                if (khalf == 0) Al[(tile & 1) * FA_ROWS + r] = alpha;
                FA_STAMP(3);
            }
            __syncthreads();                                  // Y: every S wave is past its K reads
            FA_STAMP(4);
            if (tile + 1 < n_tiles) { stage_commit(Ks, LDK); if (tile + 2 < n_tiles) stage_load((tile + 2) * FA_TK, false); }
            FA_STAMP(5);
        }
        if (khalf == 0) Il[r] = l_run > 0.0f ? 1.0f / l_run : 0.0f;
        __syncthreads();
    } else {
        // ================================================================ PV waves
        const int pw = wave - 4;
        v16f oacc[4][DBW];
#pragma unroll
        for (int rb = 0; rb < 4; rb++)
#pragma unroll
            for (int db = 0; db < DBW; db++)
#pragma unroll
                for (int i = 0; i < 16; i++) oacc[rb][db][i] = 0.0f;
        // V is staged exactly like K (f16 rows in LDS).  The A operand of O^T += V^T P^T wants, per lane, 8 POSITIONS of one dim -- a column of this image;
        // gfx950's LDS transpose-read delivers it: within a 16-lane group, lane i receives element (i & 3) of the 8 bytes addressed by lanes (i >> 2) + 4 j,
        // j = 0..3.  So lane s = 4 j + c of a group points at row (position) j, dims 4 c .. 4 c + 3 of the group's 16 dims, and lane i ends up with dim i
        // of positions 0..3: two reads (positions +0..3, +4..7) make one MFMA fragment.  (The transposing register pass of the eight-wave kernel -- 16
        // conversions + 16 byte permutes + 16 four-byte LDS writes per thread and tile -- was the longest phase of a tile.)
        typedef __fp16 fa_h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
        const int tr_off = (8 * khalf + ((lane & 15) >> 2)) * LDV + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;      // this lane's source row / dim quad inside a (16-position, 32-dim) block
        // V is double-buffered: the PV waves bring V(tile) in FIRST (their matrix work would only queue up behind the S waves' S^T MFMAs on the same SIMD for the
        // first half microsecond of an iteration), then run P.V(tile - 1) on the other buffer while the S waves are in their softmax
        stage_load(0, true);
        for (int tile = 0; tile <= n_tiles; tile++) {
            FA_STAMP2(8);
            __syncthreads();                                  // X: P(tile - 1) is in LDS; V(tile - 1) was committed before the previous Y
            FA_STAMP2(9);
            const char* pb_ = Ps + (size_t)((tile - 1) & 1) * FA_ROWS * LDP;
            const char* vb_ = Vs + (size_t)((tile - 1) & 1) * FA_TK * LDV;
            v8h vf[2][DBW], pf[2][4];
            auto fetch = [&](int kt, int bsel) {
#pragma unroll
                for (int db = 0; db < DBW; db++) {
                    const char* va = vb_ + (size_t)(16 * kt) * LDV + 32 * (pw * DBW + db) * 2 + tr_off;
                    const fa_h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fa_h4*)(va));
                    const fa_h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fa_h4*)(va + 4 * LDV));
                    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                    vf[bsel][db] = __builtin_bit_cast(v8h, u32x4{l2.x, l2.y, h2.x, h2.y});
                }
#pragma unroll
                for (int rb = 0; rb < 4; rb++) pf[bsel][rb] = *reinterpret_cast<const v8h*>(pb_ + (size_t)(32 * rb + n31) * LDP + (16 * kt + 8 * khalf) * 2);
            };
            float av[4] = {1.0f, 1.0f, 1.0f, 1.0f};
            if (tile >= 1) {                                   // the first two k-steps' fragments (and the rows' factors) travel while this wave stages V
                const float* al = Al + ((tile - 1) & 1) * FA_ROWS;
#pragma unroll
                for (int rb = 0; rb < 4; rb++) av[rb] = al[32 * rb + n31];
                fetch(0, 0); fetch(1, 1);
            }
            if (tile < n_tiles) { stage_commit(Vs + (size_t)(tile & 1) * FA_TK * LDV, LDV); if (tile + 1 < n_tiles) stage_load((tile + 1) * FA_TK, true); }
            FA_STAMP2(10);
            if (tile >= 1) {
                if (__any(av[0] != 1.0f || av[1] != 1.0f || av[2] != 1.0f || av[3] != 1.0f)) {
#pragma unroll
                    for (int rb = 0; rb < 4; rb++)
#pragma unroll
                        for (int db = 0; db < DBW; db++)
#pragma unroll
                            for (int i = 0; i < 16; i++) oacc[rb][db][i] *= av[rb];
                }
#pragma unroll
                for (int kt = 0; kt < 4; kt++) {
#pragma unroll
                    for (int rb = 0; rb < 4; rb++)
#pragma unroll
                        for (int db = 0; db < DBW; db++) oacc[rb][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[kt & 1][db], pf[kt & 1][rb], oacc[rb][db], 0, 0, 0);
                    if (kt + 2 < 4) fetch(kt + 2, kt & 1);      // two steps ahead, into the set this step has just consumed
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            FA_STAMP2(11);
            __syncthreads();                                  // Y (the S waves' K buffer turns over)
            FA_STAMP2(12);
        }
        __syncthreads();                                      // 1 / l of every row is in LDS
        // ---- normalise, gate (attention.py:664-666 / decode.rs:4272-4280), store: accumulator rows 4 g .. 4 g + 3 are dims 32 (pw DBW + db) + 8 g + 4 khalf + 0..3
#pragma unroll
        for (int rb = 0; rb < 4; rb++) {
            const int r = 32 * rb + n31, hl = r / TQ, ti = r % TQ, tok = t0 + ti;
            if (tok >= C) continue;
            const float inv = Il[r];
            const size_t ob = ((size_t)tok * a.nh + kvh * G + hl) * HD;
#pragma unroll
            for (int db = 0; db < DBW; db++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const int d = 32 * (pw * DBW + db) + 8 * g4 + 4 * khalf;
                    float4 o = make_float4(oacc[rb][db][4 * g4] * inv, oacc[rb][db][4 * g4 + 1] * inv, oacc[rb][db][4 * g4 + 2] * inv, oacc[rb][db][4 * g4 + 3] * inv);
                    if (a.gated) {
                        const float4 gt = *reinterpret_cast<const float4*>(a.gate + ob + d);
                        o.x *= 1.0f / (1.0f + kr_expf(-gt.x)); o.y *= 1.0f / (1.0f + kr_expf(-gt.y)); o.z *= 1.0f / (1.0f + kr_expf(-gt.z)); o.w *= 1.0f / (1.0f + kr_expf(-gt.w));
                    }
                    *reinterpret_cast<float4*>(a.attn_out + ob + d) = o;
                }
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
    if (a.hd >= 128) {      // wave-specialised form (S waves / PV waves)
        const size_t ldw = (size_t)FA_TK * (a.hd * 2 + 16) + 2 * (size_t)FA_TK * (a.hd * 2 + 64) + 2 * (size_t)FA_ROWS * (FA_TK * 2 + 16) + 3 * FA_ROWS * 4;
        const void* fnw = a.hd == 256 ? (a.kv_fp8 ? (const void*)kr_pfm_gqa_flashw_kernel<256, true> : (const void*)kr_pfm_gqa_flashw_kernel<256, false>)
                                      : (a.kv_fp8 ? (const void*)kr_pfm_gqa_flashw_kernel<128, true> : (const void*)kr_pfm_gqa_flashw_kernel<128, false>);
        if (kr_lds_optin(fnw, ldw) == 0) {
#define KR_FAW(H_, F_) hipLaunchKernelGGL((kr_pfm_gqa_flashw_kernel<H_, F_>), grid, dim3(512), ldw, st, a, C)
            if (a.hd == 256) { if (a.kv_fp8) KR_FAW(256, true); else KR_FAW(256, false); }
            else { if (a.kv_fp8) KR_FAW(128, true); else KR_FAW(128, false); }
#undef KR_FAW
            return 0;
        }
        return 1;           // LDS window refused: the caller reports it
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
                char* base = Vt + (size_t)(dc * (FP8 ? 16 : 8)) * LDV + fa_vslot(pp) * 4;
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
                    const v8h vv = *reinterpret_cast<const v8h*>(Vt + (size_t)(32 * (wave * DBW + db) + n31) * LDV + 32 * kt + 16 * khalf);
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vv, pf[kt >> 1][kt & 1], oacc[db], 0, 0, 0);
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
