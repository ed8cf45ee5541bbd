// kr_prefill_h.hip -- FAST (tolerance) form of the prompt-pass GEMMs: f16 activations x INT4 / INT8 weights de-quantized in registers on
// v_mfma_f32_32x32x16_f16, f32 accumulation over the WHOLE k range.
//
// Why a second GEMM.  The exact kernel (kr_prefill_gemm2.inc) reproduces the CPU engine's arithmetic: INT16 activation digits on the int8
// matrix cores, i32 sums reset every 128-wide quantization group and one f32 fma per (output, group).  That per-group epilogue is 4 VALU
// instructions per output per group -- 626 VALU per 32 MFMA in the main loop (ISA count), so the matrix pipe idles >= 60 % of the time whatever the
// tiling.  The reference's own GPU prompt pass (gpu_prefill.py:64-239, moe_wna16_marlin_gemm) does not have that epilogue: it de-quantizes the
// weight to the activation type, folds the group scale into the fragment and accumulates in f32.  This kernel is that dataflow, written for
// gfx950: tolerance mode (KR_GEMM_FAST of kr_decode_set_attention_mode / kr_moe_set_gemm_mode), checked against the exact kernel.
//
// Numerics.  A row is scaled by a power of two so that its largest |value| is in [1, 2) and rounded to f16 (RNE, 11 significant bits, vs 15
// bits per 128-group of the INT16 digits); the row multiplier comes back in the final store (exact).  An INT4 weight becomes the f16 value
// (nibble - 8) * s * 16 EXACTLY (3 x 8 significant bits; s = the bf16 group scale): the nibble is placed at mantissa bits 6..9 of 1024.0
// (0x6400 | n << 6 = 1024 + 64 n) and one packed fma (1024 + 64 n) * (s / 4) - 1536 * (s / 4) lands on (n - 8) * 16 s with a single rounding of
// an 11-bit exact value; the constant 1536 * s / 4 is exact in f16 because 1536 has two significant bits (with the nibble at bits 0..3 the
// constant 1032 * s would need 16 bits and its rounding would bias every weight of the group by up to a quarter of a quantization step).
// An INT8 weight: (0x6400 | b ^ 0x80) - 1152 = b exactly, times s * 16 (one rounding to 11 bits).  Products are accumulated by the MFMA in f32.
// The factor 16 keeps small scales out of the f16 subnormals; scales above 132 (INT4) would overflow -- two orders of magnitude beyond any weight
// a checkpoint holds.
//
// Tile: 64 rows x (128 | 256) columns per workgroup of 4 waves, each wave 64 rows x (32 | 64) columns = 2 x (1 | 2) accumulators of 32 x 32;
// one group PAIR (256 k) per LDS stage; every global load of stage st + 1 is issued before the MFMA loop of stage st (registers), as in the
// exact kernel.  Per 16-k step a wave reads 2 x 2 A fragments (b128) and NC packed words of B for 4 * NC MFMA: 80 B per lane per 8 MFMA at NC = 2,
// 62 % of the LDS bandwidth at full matrix rate (the exact kernel's 64 x 32 wave tile needs 112 %).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <type_traits>

#include "kr_lds_optin.h"
#include "kr_device.h"
#include "kr_libm.h"
#include "kr_kernels.h"
#include "kr_prefill.h"

#include "kr_pfh_dev.h"


#define PFH_BM 64
#ifdef KR_TIMING   // tools/probes/gemm_h_timing.hip: shader-clock stamps of wave 0 of one mid-grid workgroup at stage 3; no-op in the product build
__device__ unsigned long long kr_hstamps[16];
#define PFH_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 && st == 3) kr_hstamps[i] = clock64(); } while (0)
#define PFH_STAMPW(i) do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) kr_hstamps[i] = clock64(); } while (0)
#else
#define PFH_STAMP(i) do { } while (0)
#define PFH_STAMPW(i) do { } while (0)
#endif
#define PFH_KS 256
#define PFH_LDA (PFH_KS * 2 + 16)       // bytes per A row of a stage (f16)
#define PFH_LDB4 136                    // bytes per B column, INT4: 8 lane records x 16 B + 8 pad
#define PFH_LDB8 (PFH_KS + 16)          // bytes per B column, INT8

// ------------------------------------------------------------------------------------------
// A operand: rows -> f16 with a power-of-two row multiplier
// ------------------------------------------------------------------------------------------
// SRC 0: f32 rows (ld floats apart), SRC 1: bf16 rows (ld elements apart).  grid (rows), 256 threads, K % 8 == 0
// f16 sum of the 8 ROUNDED values of a chunk, completed over the 4 consecutive lanes of a 32-wide block (Q4_K copy: the operand of the offset columns)
__device__ __forceinline__ void pfh_store_sum32(const float (&v)[8], float scl, int c, uint16_t* sums_row) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += (float)(_Float16)(v[i] * scl);
    s += __int_as_float(KR_DPP(__float_as_int(s), KR_DPP_XOR1));
    s += __int_as_float(KR_DPP(__float_as_int(s), KR_DPP_XOR2));
    if ((c & 3) == 0) { const _Float16 h = (_Float16)s; sums_row[c >> 2] = __builtin_bit_cast(uint16_t, h); }
}

template <int SRC>
__global__ void __launch_bounds__(256) kr_pfh_rows_kernel(const void* __restrict__ x, int ld, int K, uint16_t* __restrict__ out, float* __restrict__ mul, uint16_t* __restrict__ sums) {
    __shared__ uint32_t smax;
    const int t = blockIdx.x;
    if (threadIdx.x == 0) smax = 0;
    __syncthreads();
    auto load8 = [&](int c, float (&v)[8]) {
        if (SRC == 0) {
            const float* p = reinterpret_cast<const float*>(x) + (size_t)t * ld + (size_t)c * 8;
            const float4 p0 = *reinterpret_cast<const float4*>(p), p1 = *reinterpret_cast<const float4*>(p + 4);
            v[0] = p0.x; v[1] = p0.y; v[2] = p0.z; v[3] = p0.w; v[4] = p1.x; v[5] = p1.y; v[6] = p1.z; v[7] = p1.w;
        } else {
            const u32x4 r = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(x) + (size_t)t * ld + (size_t)c * 8);
            v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xFFFF0000u); v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xFFFF0000u);
            v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xFFFF0000u); v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xFFFF0000u);
        }
    };
    float mx = 0.0f;
    for (int c = threadIdx.x; c < K / 8; c += 256) {
        float v[8]; load8(c, v);
#pragma unroll
        for (int i = 0; i < 8; i++) mx = fmaxf(mx, fabsf(v[i]));
    }
    mx = pfh_block_max(mx, &smax);
    float scl, inv; pfh_row_scale(mx, scl, inv);
    for (int c = threadIdx.x; c < K / 8; c += 256) {
        float v[8]; load8(c, v);
        u32x4 o;
        o.x = pfh_pack_h2(v[0] * scl, v[4] * scl); o.y = pfh_pack_h2(v[1] * scl, v[5] * scl);      // image order (0,4,1,5,2,6,3,7): kr_pfh_dev.h
        o.z = pfh_pack_h2(v[2] * scl, v[6] * scl); o.w = pfh_pack_h2(v[3] * scl, v[7] * scl);
        *reinterpret_cast<u32x4*>(out + (size_t)t * K + (size_t)c * 8) = o;
        if (sums) pfh_store_sum32(v, scl, c, sums + (size_t)t * (K / 32));
    }
    if (threadIdx.x == 0) mul[t] = inv * 0.0625f;       // 2^e / 16: undoes the row scaling and the weight factor 16
}

// hidden rows from gate | up rows: h = silu(g) * u with the exact kernel's sigmoid (avx2.rs:2310) or the GPT-OSS activation (moe.rs:268-287),
// then the same f16 row form.  grid (rows), 256 threads; the row is computed twice (max, then store) -- the inputs come from L2.
template <int ACT>
__global__ void __launch_bounds__(256) kr_pfh_act_kernel(const float* __restrict__ gu, int n, int gu_ld, float swiglu_limit, float alpha, uint16_t* __restrict__ out,
                                                        float* __restrict__ mul) {
    __shared__ uint32_t smax;
    const int row = blockIdx.x;
    const float* g = gu + (size_t)row * gu_ld;
    if (threadIdx.x == 0) smax = 0;
    __syncthreads();
    auto act8 = [&](int c, float (&h)[8]) {
        const float4 g0 = *reinterpret_cast<const float4*>(g + c * 8), g1 = *reinterpret_cast<const float4*>(g + c * 8 + 4);
        const float4 u0 = *reinterpret_cast<const float4*>(g + n + c * 8), u1 = *reinterpret_cast<const float4*>(g + n + c * 8 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, uu[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (ACT == KR_ACT_GPTOSS) {
                float gate = gg[i], up = uu[i];
                if (gate > swiglu_limit) gate = swiglu_limit;
                if (up > swiglu_limit) up = swiglu_limit;
                if (up < -swiglu_limit) up = -swiglu_limit;
                h[i] = (up + 1.0f) * (gate * kr_sigmoid_poly5_scalar(gate * alpha));
            } else h[i] = (gg[i] * kr_sigmoid_poly5(gg[i])) * uu[i];
        }
    };
    float mx = 0.0f;
    for (int c = threadIdx.x; c < n / 8; c += 256) {
        float h[8]; act8(c, h);
#pragma unroll
        for (int i = 0; i < 8; i++) mx = fmaxf(mx, fabsf(h[i]));
    }
    mx = pfh_block_max(mx, &smax);
    float scl, inv; pfh_row_scale(mx, scl, inv);
    for (int c = threadIdx.x; c < n / 8; c += 256) {
        float h[8]; act8(c, h);
        u32x4 o;
        o.x = pfh_pack_h2(h[0] * scl, h[4] * scl); o.y = pfh_pack_h2(h[1] * scl, h[5] * scl);
        o.z = pfh_pack_h2(h[2] * scl, h[6] * scl); o.w = pfh_pack_h2(h[3] * scl, h[7] * scl);
        *reinterpret_cast<u32x4*>(out + (size_t)row * n + (size_t)c * 8) = o;
    }
    if (threadIdx.x == 0) mul[row] = inv * 0.0625f;
}

// the same for rows of up to 2048 values (expert intermediates): one WAVE per row, the row's values stay in registers between the max and the store
// (no second evaluation, no LDS, no barrier); 4 rows per workgroup.  CPL = 8-value chunks per lane.
#define KR_ACT_SILU_LIBM 3   // expert_forward_gguf (gguf_kernels.rs:733-737): silu = g / (1 + exp(-g)) with libm exp, times up
#define KR_ACT_NONE 4        // the rows ARE the hidden values (formed in the epilogue of the gate | up GEMM): only the f16 row form is made here
template <int ACT, int CPL>
__global__ void __launch_bounds__(256) kr_pfh_act_wave_kernel(const float* __restrict__ gu, int rows, int n, int gu_ld, float swiglu_limit, float alpha,
                                                             uint16_t* __restrict__ out, float* __restrict__ mul, uint16_t* __restrict__ sums) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* g = gu + (size_t)row * gu_ld;
    float h[CPL][8];
    float mx = 0.0f;
#pragma unroll
    for (int q = 0; q < CPL; q++) {
        const int c = lane + 64 * q;
        if (c * 8 < n) {
            const float4 g0 = *reinterpret_cast<const float4*>(g + c * 8), g1 = *reinterpret_cast<const float4*>(g + c * 8 + 4);
            float4 u0 = g0, u1 = g1;
            if (ACT != KR_ACT_NONE) { u0 = *reinterpret_cast<const float4*>(g + n + c * 8); u1 = *reinterpret_cast<const float4*>(g + n + c * 8 + 4); }
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, uu[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (ACT == KR_ACT_NONE) h[q][i] = gg[i];
                else if (ACT == KR_ACT_GPTOSS) {
                    float gate = gg[i], up = uu[i];
                    if (gate > swiglu_limit) gate = swiglu_limit;
                    if (up > swiglu_limit) up = swiglu_limit;
                    if (up < -swiglu_limit) up = -swiglu_limit;
                    h[q][i] = (up + 1.0f) * (gate * kr_sigmoid_poly5_scalar(gate * alpha));
                } else if (ACT == KR_ACT_SILU_LIBM) h[q][i] = (gg[i] / (1.0f + kr_expf(-gg[i]))) * uu[i];
                else h[q][i] = (gg[i] * kr_sigmoid_poly5(gg[i])) * uu[i];
                mx = fmaxf(mx, fabsf(h[q][i]));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) h[q][i] = 0.0f;
        }
    }
    mx = kr_red16_max_f32(mx);
    mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
    float scl, inv; pfh_row_scale(mx, scl, inv);
#pragma unroll
    for (int q = 0; q < CPL; q++) {
        const int c = lane + 64 * q;
        if (c * 8 < n) {
            u32x4 o;
            o.x = pfh_pack_h2(h[q][0] * scl, h[q][4] * scl); o.y = pfh_pack_h2(h[q][1] * scl, h[q][5] * scl);
            o.z = pfh_pack_h2(h[q][2] * scl, h[q][6] * scl); o.w = pfh_pack_h2(h[q][3] * scl, h[q][7] * scl);
            *reinterpret_cast<u32x4*>(out + (size_t)row * n + (size_t)c * 8) = o;
            if (sums) pfh_store_sum32(h[q], scl, c, sums + (size_t)row * (n / 32));
        }
    }
    if (lane == 0) mul[row] = inv * 0.0625f;
}

// ------------------------------------------------------------------------------------------
// the GEMM
// ------------------------------------------------------------------------------------------




// SB = 1: the de-quantized B fragments of a k-step are SINGLE-buffered -- fragment (c, h) of step t + 1 is formed right after the MFMAs of step t that
// read fragment (c, h) have been issued, into the same registers (16 registers less than two fragment sets, 8 less for the raw INT4 words): the
// 64 x 256 tile then fits the 256 registers of two waves per SIMD without scratch (the double-buffered form spilled 29 VGPRs, 11 of the reloads inside
// the MFMA block).
// G = 1: the Q4_K copy (KrMatDev::qs / qo): one scale per 32-wide sub-block -- a k-step of this kernel IS one sub-block (both lane halves together
// cover k = 32 t .. 32 t + 32 of the stage), so the step picks its scale from the 8 the stage loaded; the per-sub-block offsets
// (8 d sc_j - dmin mn_j) enter after the k loop as K / 32 extra k-columns: A' = the rows' per-32 sums, B' = the offset table (K / 512 more MFMA steps).
template <int NC, int BITS, int SB = 0, int G = 0, int OCC = 2>
__global__ void __launch_bounds__(256, OCC) kr_pfh_gemm_kernel(const KrPfGemmHArgs a) {
    static_assert(G == 0 || (SB == 1 && BITS == 4) || (SB == 0 && BITS == 8), "the Q4_K copy runs the single-buffered INT4 form, the Q8_0 copy the INT8 form");
    constexpr int BN = 128 * NC, LDA = PFH_LDA, LDB = BITS == 8 ? PFH_LDB8 : PFH_LDB4, NS = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                                               // [64][LDA]  f16; BITS == 4: k in image order (0,4,1,5,2,6,3,7) inside every 8, BITS == 8: natural
    char* Bs = As + PFH_BM * LDA;                                  // [BN][LDB]  packed lane records, copied verbatim
    float* rmul = reinterpret_cast<float*>(Bs + BN * LDB);         // [64]
    int* row_src = reinterpret_cast<int*>(rmul + PFH_BM);          // [64]
    int* row_dst = row_src + PFH_BM;                               // [64]
    int* row_idx = row_dst + PFH_BM;                               // [64] source row (G = 1: addresses the per-32 sums)

    PFH_STAMPW(6);
    const bool actf = NC == 2 && a.act_fused != 0;       // uniform; N = 2 I with I % 128 == 0 (the launcher checks): a tile = 128 gate + the 128 matching up columns
    const int ncb0 = (a.m.N + BN - 1) / BN, ncb1 = a.n_extra > 0 ? (a.mx[0].N + BN - 1) / BN : 0, ncb2 = a.n_extra > 1 ? (a.mx[1].N + BN - 1) / BN : 0;
    const int ncb = ncb0 + ncb1 + ncb2, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int mt, cb;
    if (a.single_expert) {
        // dense: an XCD works on one SUPER-TILE of sr x sc (row tile, column block) pairs at a time -- its 32 CUs hold ~64 of them: sr A tiles and sc B
        // slices (256 KiB each at K = 2048) serve sr * sc workgroups from one L2.  Row-tile-major order streams a fresh B slice from the Infinity
        // Cache / HBM for every workgroup (half of all operand bytes; a CU pulls only ~10-13 B/clk from beyond its L2).
        const int nrt = (a.total_rows + PFH_BM - 1) / PFH_BM, nsc = (ncb + a.sc - 1) / a.sc, ssz = a.sr * a.sc;
        const int sup = (slot / ssz) * 8 + xcd, w = slot % ssz;
        mt = (sup / nsc) * a.sr + w % a.sr; cb = (sup % nsc) * a.sc + w / a.sr;
        if (mt >= nrt || cb >= ncb) return;
    } else {
        const int per = ncb * a.run, grp = slot / per, local = slot - grp * per;
        mt = (grp * 8 + xcd) * a.run + local / ncb; cb = local % ncb;
    }
    int expert, row0, rows;
    if (a.single_expert) { expert = 0; row0 = mt * PFH_BM; rows = a.total_rows - row0 < PFH_BM ? a.total_rows - row0 : PFH_BM; if (rows <= 0) return; }
    else { if (mt >= a.n_tiles[0]) return; expert = a.tile_expert[mt]; row0 = a.tile_row0[mt]; rows = a.tile_rows[mt]; }
    KrMatDev m = a.m; float* out_p = a.out; int out_ld = a.out_ld;
    if (cb >= ncb0 + ncb1) { cb -= ncb0 + ncb1; m = a.mx[1]; out_p = a.outx[1]; out_ld = a.out_ldx[1]; }
    else if (cb >= ncb0) { cb -= ncb0; m = a.mx[0]; out_p = a.outx[0]; out_ld = a.out_ldx[0]; }
    const int n0 = cb * BN;
    const bool two = rows > 32;                                     // tiles of a short chunk: skip the empty second 32-row block (uniform)
    const int K = m.ng * 128;
    const char* wq = reinterpret_cast<const char*>(m.q) + (size_t)expert * m.q_stride;
    const uint32_t* wsc = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(m.s) + (size_t)expert * m.s_stride);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    v16f acc[NS][NC];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[s][c][r] = 0.0f;
    const int n31 = lane & 31, khalf = lane >> 5;
    uint32_t M0 = 0x000F000Fu, M1 = 0x00F000F0u, MH = 0x03C003C0u, Kc = 0x64006400u;      // de-quantization masks, kept in registers (see pfh_dq4)
    asm volatile("" : "+v"(M0), "+v"(M1), "+v"(MH), "+v"(Kc));
    // column blocks of a wave inside the tile's 128 NC local columns: consecutive (wave * 32 NC + 32 c), or -- fused activation -- block c of the
    // gate half / up half (local 128 c + 32 wave), so that accumulator element (c = 0, r) and (c = 1, r) of a lane are one (gate, up) pair
    const int half_n = m.N >> 1, n0h = cb * 128;
    int col[NC], ctile[NC], cin[NC], lcolb[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int lc = actf ? c * 128 + wave * 32 + n31 : wave * (32 * NC) + c * 32 + n31;
        lcolb[c] = lc * LDB;
        col[c] = actf ? c * half_n + n0h + wave * 32 + n31 : n0 + lc;
        const int cc = col[c] < m.N ? col[c] : m.N - 1; ctile[c] = cc >> 3; cin[c] = cc & 7;
    }

    constexpr int APT = 8;                         // 16-byte A chunks per thread per stage (4 threads per row, 128 B each)
    constexpr int RPT = BN * 8 / 256;              // B lane records per thread (per group for INT8)
    constexpr int NBW = BITS == 8 ? 2 * RPT : RPT;
    // A requests cover WHOLE 128-byte lines: a wave-instruction takes 8 lines (8 lanes x 16 B each) = 2 rows x 4 lines of the 512-byte row segment of
    // the stage.  (4 threads per row x 128 contiguous bytes per thread, the mapping of the exact kernel, puts the 64 lanes of one instruction on 64
    // different lines, each line re-requested by 8 instructions: the wave sat ~150 cycles in the issue of every load.)
    // load j of thread tid: row 8 j + (tid >> 5), line (tid >> 3) & 3 of the segment, chunk tid & 7 of the line
    const int arow = tid >> 5, aseg = (tid >> 3) & 3, achk = tid & 7;
    const uint32_t* rofs = reinterpret_cast<const uint32_t*>(row_src);      // byte offset of every tile row in the A matrix
    const int nst = m.ngp;
    u32x4 pa[APT], pbw[NBW];
    uint32_t pspv[NC];
    u32x4 pq[G ? NC : 1], sqw[G ? NC : 1];            // G = 1: the 8 f16 sub-block scales of this lane's column(s) for the stage
    const char* qs_b = G ? reinterpret_cast<const char*>(m.qs) + (size_t)expert * m.qs_stride : nullptr;
    // Loads are never masked: a tile row past `rows` reads row 0 / token 0, a column tile past the last one re-reads the last tile, a line past K
    // re-reads a valid line of the row -- all finite or irrelevant: rows and columns are independent in a GEMM, the stores are guarded and a group
    // past ng gets scale 0.  A: whole 128-byte lines per request, row offsets from the LDS table (above); B: a wave-uniform record base per load
    // (the 64 lanes of a wave copy one 1-KiB tile record row: scalar base + lane * 16).
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int last_tile = (m.N - 1) >> 3;
    // (buffer-descriptor loads for the two operand streams were measured here: 1.42 ms per expert layer against 1.22 ms with flat addresses)
    uint32_t brec[RPT];                                              // byte offset of this wave's tile records inside the expert's block (scalars)
#pragma unroll
    for (int j = 0; j < RPT; j++) {
        int tile = (n0 >> 3) + wv + 4 * j;
        if (actf) tile = (n0h >> 3) + wv + 4 * (j % (RPT / 2)) + (j >= RPT / 2 ? half_n >> 3 : 0);
        tile = tile < last_tile ? tile : last_tile;
        brec[j] = (uint32_t)__builtin_amdgcn_readfirstlane(tile * (BITS == 8 ? m.ng : m.ngp) * 1024);
    }
    auto load_B = [&](int st) {      // scales + weight records of a stage: they do not depend on the row table
        if constexpr (G) {
#pragma unroll
            for (int c = 0; c < NC; c++) pq[c] = *reinterpret_cast<const u32x4*>(qs_b + (((size_t)ctile[c] * m.ngp + st) * 8 + cin[c]) * 16);
        } else {
#pragma unroll
            for (int c = 0; c < NC; c++) pspv[c] = wsc[((size_t)ctile[c] * m.ngp + st) * 8 + cin[c]];
        }
        if (BITS == 8) {
#pragma unroll
            for (int gg = 0; gg < 2; gg++) {
                int g = 2 * st + gg; g = g < m.ng ? g : m.ng - 1;
#pragma unroll
                for (int j = 0; j < RPT; j++) pbw[(BITS == 8 ? gg * RPT : 0) + j] = kr_ldg_nt(reinterpret_cast<const u32x4*>(wq + brec[j] + (size_t)g * 1024) + lane);
            }
        } else {
#pragma unroll
            for (int j = 0; j < RPT; j++) pbw[j] = kr_ldg_nt(reinterpret_cast<const u32x4*>(wq + brec[j] + (size_t)st * 1024) + lane);
        }
    };
    auto load_A = [&](int st) {
        {
            const int kvalid = K - st * PFH_KS;                  // 256, or 128 in the last stage of an odd group count: lines 2, 3 re-read lines 0, 1
            const uint32_t segoff = (uint32_t)((aseg * 64 < kvalid ? aseg : aseg - 2) * 128 + achk * 16);
            const char* ab = reinterpret_cast<const char*>(a.a) + (size_t)st * (PFH_KS * 2) + segoff;
#pragma unroll
            for (int j = 0; j < APT; j++) pa[j] = *reinterpret_cast<const u32x4*>(ab + rofs[arow + 8 * j]);
        }
    };
    auto load_stage = [&](int st) { load_A(st); load_B(st); };
    // Prologue order: the first stage's weights are requested BEFORE the row table is built (tile info -> row_pair -> table -> barrier -> A rows is a
    // chain of dependent round trips; the weight stream needs none of it), and the rows' multipliers -- only the store phase reads them -- are fetched
    // here but parked in LDS after the k loop.  A w2 tile (K = 512: two stages) spent as long getting started as in its k loop.
    // (wave 0 builds the table: the in-order load counter would make its table reads wait for the weight records, so it requests them after the table)
    if (wave != 0) load_B(0);
    float mulv = 0.0f;
    if (tid < PFH_BM) {
        int src = -1;
        if (tid < rows) {
            if (a.single_expert) src = row0 + tid;
            else { const int pair = a.row_pair[row0 + tid]; src = a.gather_tokens ? pair / a.topk : row0 + tid; }
        }
        row_src[tid] = (int)((uint32_t)(src < 0 ? 0 : src) * (uint32_t)(K * 2));      // byte offset of the row in the A matrix (rows past `rows`: row 0)
        row_dst[tid] = (a.scatter_rows && !a.single_expert && tid < rows) ? a.row_pair[row0 + tid] : row0 + tid;
        row_idx[tid] = src < 0 ? 0 : src;
        if (src >= 0) mulv = a.a_mul[src];
    }
    __syncthreads();

    if (wave == 0) load_B(0);
    load_A(0);
    uint32_t spv[NC];
    auto commit_stage = [&]() {
        if constexpr (G) {
#pragma unroll
            for (int c = 0; c < NC; c++) sqw[c] = pq[c];
        } else {
#pragma unroll
            for (int c = 0; c < NC; c++) spv[c] = pspv[c];
        }
#pragma unroll
        for (int j = 0; j < APT; j++) {
            u32x4 v = pa[j];
            if (BITS == 8)      // the row image holds (a0,a4 | a1,a5 | a2,a6 | a3,a7) (kr_pfh_dev.h); pfh_dq8 emits natural k order: (a0,a1 | a2,a3 | a4,a5 | a6,a7)
                v = u32x4{__builtin_amdgcn_perm(v.y, v.x, 0x05040100u), __builtin_amdgcn_perm(v.w, v.z, 0x05040100u),
                          __builtin_amdgcn_perm(v.y, v.x, 0x07060302u), __builtin_amdgcn_perm(v.w, v.z, 0x07060302u)};
            *reinterpret_cast<u32x4*>(As + (arow + 8 * j) * LDA + aseg * 128 + achk * 16) = v;
        }
        if (BITS == 8) {
#pragma unroll
            for (int gg = 0; gg < 2; gg++)
#pragma unroll
                for (int j = 0; j < RPT; j++) {
                    const int rec = tid + j * 256, t8 = rec >> 6, ln = rec & 63, c = ln >> 3, l8 = ln & 7;
                    *reinterpret_cast<u32x4*>(Bs + (t8 * 8 + c) * LDB + gg * 128 + l8 * 16) = pbw[(BITS == 8 ? gg * RPT : 0) + j];
                }
        } else {
#pragma unroll
            for (int j = 0; j < RPT; j++) {
                const int rec = tid + j * 256, t8 = rec >> 6, ln = rec & 63, c = ln >> 3, l8 = ln & 7;
                const u32x4 w = pbw[j];
                const u32x2 h0 = {w.x, w.y}, h1 = {w.z, w.w};     // 8-byte stores: the padded column stride is 8-byte aligned only
                *reinterpret_cast<u32x2*>(Bs + (t8 * 8 + c) * LDB + l8 * 16) = h0;
                *reinterpret_cast<u32x2*>(Bs + (t8 * 8 + c) * LDB + l8 * 16 + 8) = h1;
            }
        }
    };
    // The MFMA loop of one stage for NSA active 32-row blocks (compile-time: the whole stage is ONE basic block), software-pipelined over its
    // 8 k-steps of 16: while the 4 * NSA... MFMAs of step t run, the wave de-quantizes the B words of step t + 1 and the LDS reads of step t + 2 are
    // issued (a wave issues in order: an LDS wait or a dependent MFMA in front of the VALU work would idle the matrix pipe -- SQ_WAIT_INST_ANY was
    // 45 % and SQ_WAIT_ANY 23 % of the wave cycles of the unpipelined form).  Consecutive MFMAs go to different accumulators.
    auto stage_mfma = [&](int st, auto nsa) {
        constexpr int NSA = decltype(nsa)::value;
        v2h sq[2][NC], cq[2][NC];
        if constexpr (!G)
#pragma unroll
        for (int hh = 0; hh < 2; hh++)
#pragma unroll
            for (int c = 0; c < NC; c++) {
                float sc = __uint_as_float((hh ? (spv[c] >> 16) : (spv[c] & 0xFFFFu)) << 16);
                if (2 * st + hh >= m.ng) sc = 0.0f;                      // the missing second group of an odd group count contributes 0
                const _Float16 s1 = (_Float16)(BITS == 8 ? sc * 16.0f : sc * 0.25f);
                sq[hh][c] = v2h{s1, s1};
                const _Float16 c1 = (_Float16)(-1536.0f * (float)s1);
                cq[hh][c] = v2h{c1, c1};
            }
        v8h af[2][NSA][2];
        if constexpr (SB == 1) {
            static_assert(BITS == 4, "single-buffered fragments: INT4 form");
            v8h bf1[NC][2];
            u32x2 br2[2][NC];
            auto rd = [&](int t, int buf) {
                const int hh = t >> 2, lp = 2 * (t & 3) + khalf;
#pragma unroll
                for (int s2 = 0; s2 < NSA; s2++) {
                    af[buf][s2][0] = *reinterpret_cast<const v8h*>(As + (s2 * 32 + n31) * LDA + hh * 256 + lp * 32);
                    af[buf][s2][1] = *reinterpret_cast<const v8h*>(As + (s2 * 32 + n31) * LDA + hh * 256 + lp * 32 + 16);
                }
#pragma unroll
                for (int c = 0; c < NC; c++) br2[buf][c] = *reinterpret_cast<const u32x2*>(Bs + lcolb[c] + lp * 16 + hh * 8);
            };
            v2h sqs[NC], cqs[NC];            // G = 1: scale / constant of the step being de-quantized
            auto step_scale = [&](int t) {
                if constexpr (G) {
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        const uint32_t w4[4] = {sqw[c].x, sqw[c].y, sqw[c].z, sqw[c].w};
                        const uint32_t wd = w4[t >> 1], s16 = (t & 1) ? (wd >> 16) : (wd & 0xFFFFu);
                        sqs[c] = __builtin_bit_cast(v2h, s16 | (s16 << 16));
                        cqs[c] = sqs[c] * v2h{(_Float16)-1536.0f, (_Float16)-1536.0f};
                    }
                }
            };
            auto dq1 = [&](int t, int buf, int c, int h) {
                const int hh = t >> 2;
                if constexpr (G) bf1[c][h] = pfh_dq4(h ? br2[buf][c].y : br2[buf][c].x, sqs[c], cqs[c], M0, M1, MH, Kc);
                else bf1[c][h] = pfh_dq4(h ? br2[buf][c].y : br2[buf][c].x, sq[hh][c], cq[hh][c], M0, M1, MH, Kc);
            };
            rd(0, 0); rd(1, 1);
            step_scale(0);
#pragma unroll
            for (int c = 0; c < NC; c++) { dq1(0, 0, c, 0); dq1(0, 0, c, 1); }
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const int cur = t & 1, nxt = cur ^ 1;
                if (t + 1 < 8) step_scale(t + 1);
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int c = 0; c < NC; c++) {
#pragma unroll
                        for (int s2 = 0; s2 < NSA; s2++) acc[s2][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][s2][h], bf1[c][h], acc[s2][c], 0, 0, 0);
                        if (t + 1 < 8) dq1(t + 1, nxt, c, h);
                    }
                if (t + 2 < 8) rd(t + 2, cur);
                // issue order: the NSA MFMAs of a fragment, then the 12 VALU that rebuild it for the next step, ...; the LDS reads of step t + 2 last
#pragma unroll
                for (int i = 0; i < 2 * NC; i++) {
                    // round 6 (the ring kernel's finding, A/B on one box: experts only 0.98 -> 0.96 ms per layer, Q4_K copy 1.09 -> 1.07): a pair issued back to back parks the
                    // wave on the busy matrix pipe while its vector work waits -- one MFMA, the 8 mask / shift operations of the fragment being rebuilt, the second MFMA,
                    // the 4 packed fma (+ address work); the fma write the registers the pair's second MFMA still reads, so they cannot move up
                    if (NSA == 2) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                        continue;
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, NSA, 0);     // NSA MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 13, 0);      // the fragment's de-quantization
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * NSA + NC, 0);   // DS reads
            }
            return;
        }
        v8h bf[2][NC][2];
        u32x4 br[2][NC];          // raw B words of a step (INT4 uses .x .y)
        auto rd = [&](int t, int buf) {
            const int hh = t >> 2, lp = 2 * (t & 3) + khalf;             // lane record of this lane half: k = 16 lp .. 16 lp + 16 of group hh
#pragma unroll
            for (int s2 = 0; s2 < NSA; s2++) {
                af[buf][s2][0] = *reinterpret_cast<const v8h*>(As + (s2 * 32 + n31) * LDA + hh * 256 + lp * 32);
                af[buf][s2][1] = *reinterpret_cast<const v8h*>(As + (s2 * 32 + n31) * LDA + hh * 256 + lp * 32 + 16);
            }
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (BITS == 8) br[buf][c] = *reinterpret_cast<const u32x4*>(Bs + lcolb[c] + hh * 128 + lp * 16);
                else { const u32x2 pk = *reinterpret_cast<const u32x2*>(Bs + lcolb[c] + lp * 16 + hh * 8); br[buf][c].x = pk.x; br[buf][c].y = pk.y; }
            }
        };
        auto dq = [&](int t, int buf) {
            const int hh = t >> 2;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if constexpr (G && BITS == 8) {      // Q8_0 copy: k-step t of the stage lies in 32-wide block t, whose f16 scale (16 d) is half t of the 8 the stage loaded
                    const uint32_t w4[4] = {sqw[c].x, sqw[c].y, sqw[c].z, sqw[c].w};
                    const uint32_t wd = w4[t >> 1], s16 = (t & 1) ? (wd >> 16) : (wd & 0xFFFFu);
                    const v2h sc2 = __builtin_bit_cast(v2h, s16 | (s16 << 16));
                    bf[buf][c][0] = pfh_dq8(br[buf][c].x, br[buf][c].y, sc2); bf[buf][c][1] = pfh_dq8(br[buf][c].z, br[buf][c].w, sc2);
                } else
                if (BITS == 8) { bf[buf][c][0] = pfh_dq8(br[buf][c].x, br[buf][c].y, sq[hh][c]); bf[buf][c][1] = pfh_dq8(br[buf][c].z, br[buf][c].w, sq[hh][c]); }
                else { bf[buf][c][0] = pfh_dq4(br[buf][c].x, sq[hh][c], cq[hh][c], M0, M1, MH, Kc); bf[buf][c][1] = pfh_dq4(br[buf][c].y, sq[hh][c], cq[hh][c], M0, M1, MH, Kc); }
            }
        };
        rd(0, 0); rd(1, 1); dq(0, 0);
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int cur = t & 1, nxt = cur ^ 1;
            if (t + 1 < 8) dq(t + 1, nxt);
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int c = 0; c < NC; c++)
#pragma unroll
                    for (int s2 = 0; s2 < NSA; s2++) acc[s2][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][s2][h], bf[cur][c][h], acc[s2][c], 0, 0, 0);
            if (t + 2 < 8) rd(t + 2, cur);
            // issue order of the step: one MFMA, then a share of the de-quantization VALU, ...; the LDS reads of step t + 2 last
            constexpr int NM = 2 * NC * NSA, VPM = (BITS == 8 ? 30 : 24) * NC / NM + 1;
#pragma unroll
            for (int i = 0; i < NM; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);     // VPM VALU
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * NSA + NC, 0);   // DS reads
        }
    };
    PFH_STAMPW(7);
    auto main_loop = [&](auto nsa) {          // one copy of the loop per number of active row blocks (chosen once per workgroup)
        for (int st = 0; st < nst; st++) {
            PFH_STAMP(0);
            commit_stage();
            PFH_STAMP(1);
            if (st + 1 < nst) load_stage(st + 1);
            PFH_STAMP(2);
            __syncthreads();
            PFH_STAMP(3);
            stage_mfma(st, nsa);
            PFH_STAMP(4);
            __syncthreads();
            PFH_STAMP(5);
        }
    };
    // SB form (big problems: nearly every tile is full): one copy of the loop -- the second copy's hoisted values were what pushed the kernel into scratch
    if (SB || two) main_loop(std::integral_constant<int, 2>{}); else main_loop(std::integral_constant<int, 1>{});
    if constexpr (G && BITS == 4) {
        // the offset columns: 16 sub-blocks per MFMA step, operands straight from global memory (8 f16 per lane and operand)
        const int nsub = K / 32;
        const char* qo_b = reinterpret_cast<const char*>(m.qo) + (size_t)expert * m.qs_stride;
        for (int q0 = 0; q0 < nsub; q0 += 16) {
            const int sb0 = q0 + 8 * khalf;                     // first sub-block of this lane half
            const bool live = sb0 < nsub;
            v8h av[NS], bv[NC];
#pragma unroll
            for (int s2 = 0; s2 < NS; s2++) {
                av[s2] = v8h{0, 0, 0, 0, 0, 0, 0, 0};
                if (live) av[s2] = *reinterpret_cast<const v8h*>(a.a_sum + (size_t)row_idx[s2 * 32 + n31] * nsub + sb0);
            }
#pragma unroll
            for (int c = 0; c < NC; c++) {
                bv[c] = v8h{0, 0, 0, 0, 0, 0, 0, 0};
                if (live) bv[c] = *reinterpret_cast<const v8h*>(qo_b + (((size_t)ctile[c] * m.ngp + (sb0 >> 3)) * 8 + cin[c]) * 16);
            }
#pragma unroll
            for (int s2 = 0; s2 < NS; s2++)
#pragma unroll
                for (int c = 0; c < NC; c++) acc[s2][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s2], bv[c], acc[s2][c], 0, 0, 0);
        }
    }
    PFH_STAMPW(8);
    if (tid < PFH_BM) rmul[tid] = a.out_bf16 == 2 ? 1.0f : mulv;      // f16 rows carry the raw accumulators: the combine pass applies the row multiplier
    __syncthreads();
    {
        const bool full = rows == (two ? 64 : 32) && (actf || n0 + BN <= m.N) && !(a.scatter_rows && !a.single_expert);     // uniform
        const int nsb = two ? 2 : 1;
#define PFH_ST(F_, OT_, A_) pfh_store_tile<NS, NC, F_, OT_, A_>(acc, nsb, rows, row0, rmul, row_dst, out_p, out_ld, col, m.N, lane, a.act_fused, a.act_limit, a.act_alpha)
        if (NC == 2 && a.act_fused) { if (full) PFH_ST(true, 0, NC == 2); else PFH_ST(false, 0, NC == 2); }
        else if (a.out_bf16 == 1) { if (full) PFH_ST(true, 1, false); else PFH_ST(false, 1, false); }
        else if (a.out_bf16 == 2) { if (full) PFH_ST(true, 2, false); else PFH_ST(false, 2, false); }
        else { if (full) PFH_ST(true, 0, false); else PFH_ST(false, 0, false); }
#undef PFH_ST
    }
    PFH_STAMPW(9);
}


template <int NC, int BITS, int SB = 0, int G = 0, int OCC = 2>
static void pfh_launch(const KrPfGemmHArgs& a, int mt, hipStream_t st) {
    constexpr int BN = 128 * NC, LDB = BITS == 8 ? PFH_LDB8 : PFH_LDB4;
    const size_t lds = (size_t)PFH_BM * PFH_LDA + (size_t)BN * LDB + 4 * PFH_BM * 4;
    (void)kr_lds_optin((const void*)kr_pfh_gemm_kernel<NC, BITS, SB, G, OCC>, 80 * 1024);
    int ncb = (a.m.N + BN - 1) / BN;
    for (int i = 0; i < a.n_extra; i++) ncb += (a.mx[i].N + BN - 1) / BN;
    KrPfGemmHArgs b = a;
    dim3 grid;
    if (a.single_expert) { int n_super; kr_pf_super_tile(mt, ncb, &b.sr, &b.sc, &n_super); grid = dim3(((n_super + 7) / 8) * 8 * b.sr * b.sc); }
    else { const int span = 8 * a.run; grid = dim3(((mt + span - 1) / span) * span * ncb); }
    hipLaunchKernelGGL((kr_pfh_gemm_kernel<NC, BITS, SB, G, OCC>), grid, dim3(256), lds, st, b);
}
static void pfh_dispatch(const KrPfGemmHArgs& a, int mt, hipStream_t st) {
    if (kr_pfr_try_launch(a, mt, st) == 0) return;       // big INT4 problems: the LDS-ring form (bit-identical results)
    if (a.m.bits == 8) { if (a.m.qs) pfh_launch<1, 8, 0, 1>(a, mt, st); else pfh_launch<1, 8>(a, mt, st); return; }
    // 256-column tiles when they still fill the chip (>= 2 workgroups per CU), else 128-column tiles
    long n128 = (a.m.N + 127) / 128;
    for (int i = 0; i < a.n_extra; i++) n128 += (a.mx[i].N + 127) / 128;
    // the 64 x 256 form: ONE copy of the stage loop, single-buffered B fragments -- 220 VGPRs, no scratch.  Measured on MI355X (experts only, QCN shape,
    // 8192 tokens, tools/probes/experts_gemm_probe.py): 1.65 ms per layer for the round-2 form (two loop copies, 29 VGPRs in scratch), 1.49 ms for
    // this one, 1.48 ms with double-buffered fragments and one loop copy (223 VGPRs; not kept: same speed, fewer registers to spare)
    if (a.m.qs) {      // Q4_K copy
        if ((long)mt * n128 >= 2048) pfh_launch<2, 4, 1, 1>(a, mt, st); else pfh_launch<1, 4, 1, 1>(a, mt, st);
        return;
    }
#ifdef KR_PFH_OCC3      // A/B build: 64 x 128 tiles, one loop copy, three workgroups per CU (<= 168 registers, 52 KB of LDS each)
    pfh_launch<1, 4, 1, 0, 3>(a, mt, st);
#else
    if ((long)mt * n128 >= 2048) pfh_launch<2, 4, 1>(a, mt, st);
    else pfh_launch<1, 4>(a, mt, st);
#endif
}

void kr_launch_pfh_rows_f32(const float* x, int rows, int ld, int K, uint16_t* out, float* mul, hipStream_t st) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(kr_pfh_rows_kernel<0>, dim3(rows), dim3(256), 0, st, (const void*)x, ld, K, out, mul, (uint16_t*)nullptr);
}
void kr_launch_pfh_rows_bf16(const uint16_t* x, int rows, int ld, int K, uint16_t* out, float* mul, hipStream_t st, uint16_t* sums32) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(kr_pfh_rows_kernel<1>, dim3(rows), dim3(256), 0, st, (const void*)x, ld, K, out, mul, sums32);
}
void kr_launch_pfh_act(const float* gu, int rows, int n, int gu_ld, int act_mode, float swiglu_limit, float alpha, uint16_t* out, float* mul, hipStream_t st, uint16_t* sums32) {
    if (rows <= 0) return;
    const bool oss = act_mode == KR_ACT_GPTOSS, libm = act_mode == KR_ACT_SILU_LIBM;
#define KR_ACTW(C_) do { if (oss) hipLaunchKernelGGL((kr_pfh_act_wave_kernel<KR_ACT_GPTOSS, C_>), dim3((rows + 3) / 4), dim3(256), 0, st, gu, rows, n, gu_ld, swiglu_limit, alpha, out, mul, sums32); \
                         else if (libm) hipLaunchKernelGGL((kr_pfh_act_wave_kernel<KR_ACT_SILU_LIBM, C_>), dim3((rows + 3) / 4), dim3(256), 0, st, gu, rows, n, gu_ld, swiglu_limit, alpha, out, mul, sums32); \
                         else hipLaunchKernelGGL((kr_pfh_act_wave_kernel<KR_ACT_SILU_MUL, C_>), dim3((rows + 3) / 4), dim3(256), 0, st, gu, rows, n, gu_ld, swiglu_limit, alpha, out, mul, sums32); } while (0)
    if (n <= 512) { KR_ACTW(1); return; }
    if (n <= 1024) { KR_ACTW(2); return; }
    if (n <= 2048 || libm || sums32) { KR_ACTW(4); if (n <= 2048) return; }
#undef KR_ACTW
    if (oss) hipLaunchKernelGGL(kr_pfh_act_kernel<KR_ACT_GPTOSS>, dim3(rows), dim3(256), 0, st, gu, n, gu_ld, swiglu_limit, alpha, out, mul);
    else hipLaunchKernelGGL(kr_pfh_act_kernel<KR_ACT_SILU_MUL>, dim3(rows), dim3(256), 0, st, gu, n, gu_ld, swiglu_limit, alpha, out, mul);
}
void kr_launch_pfh_gemm(const KrMatDev& m, const uint16_t* a_h, const float* a_mul, const KrPfSort* sort, int topk, int gather_tokens, int max_tiles,
                        int single_expert_rows, float* out, int out_ld, hipStream_t st, int scatter_rows, int out_bf16, int run, const uint16_t* a_sum32) {
    KrPfGemmHArgs a{};
    a.m = m; a.a = a_h; a.a_mul = a_mul; a.a_sum = a_sum32; a.topk = topk; a.gather_tokens = gather_tokens; a.scatter_rows = scatter_rows; a.out_bf16 = out_bf16;
    if (sort) { a.row_pair = sort->row_pair; a.tile_expert = sort->tile_expert; a.tile_row0 = sort->tile_row0; a.tile_rows = sort->tile_rows; a.n_tiles = sort->n_tiles; }
    a.out = out; a.out_ld = out_ld; a.single_expert = single_expert_rows > 0; a.total_rows = single_expert_rows;
    a.run = (single_expert_rows > 0 || run < 1) ? 1 : run;
    const int mt = single_expert_rows > 0 ? (single_expert_rows + PFH_BM - 1) / PFH_BM : max_tiles;
    pfh_dispatch(a, mt, st);
}
// gate | up GEMM + activation -> the f16 hidden rows (+ multipliers, + per-32 sums for a Q4_K down copy).  Big problems (the 64 x 256 tile form) with
// I % 128 == 0 form h in the GEMM's epilogue: `gu` then holds [rows][I] f32 hidden values instead of [rows][2 I] gate | up values, and the row pass
// only makes the f16 form.  Otherwise: the GEMM, then the activation pass over gate | up rows.  Same arithmetic either way.
void kr_launch_pfh_w13_act(const KrMatDev& m, const uint16_t* a_h, const float* a_mul, const KrPfSort* sort, int topk, int gather_tokens, int max_tiles,
                           int single_expert_rows, float* gu, int rows, int act_mode, float swiglu_limit, float alpha, uint16_t* h_out, float* h_mul,
                           hipStream_t st, int run, const uint16_t* a_sum32, uint16_t* h_sums32) {
    const int I = m.N / 2;
    const int mt = single_expert_rows > 0 ? (single_expert_rows + PFH_BM - 1) / PFH_BM : max_tiles;
    const bool fuse = m.bits == 4 && I % 128 == 0 && I <= 2048 && (long)mt * (m.N / 128) >= 2048;
    if (!fuse) {
        kr_launch_pfh_gemm(m, a_h, a_mul, sort, topk, gather_tokens, max_tiles, single_expert_rows, gu, 2 * I, st, 0, 0, run, a_sum32);
        kr_launch_pfh_act(gu, rows, I, 2 * I, act_mode, swiglu_limit, alpha, h_out, h_mul, st, h_sums32);
        return;
    }
    KrPfGemmHArgs a{};
    a.m = m; a.a = a_h; a.a_mul = a_mul; a.a_sum = a_sum32; a.topk = topk; a.gather_tokens = gather_tokens;
    if (sort) { a.row_pair = sort->row_pair; a.tile_expert = sort->tile_expert; a.tile_row0 = sort->tile_row0; a.tile_rows = sort->tile_rows; a.n_tiles = sort->n_tiles; }
    a.out = gu; a.out_ld = I; a.single_expert = single_expert_rows > 0; a.total_rows = single_expert_rows;
    a.run = (single_expert_rows > 0 || run < 1) ? 1 : run;
    a.act_fused = act_mode == KR_ACT_GPTOSS ? 2 : (act_mode == KR_ACT_SILU_LIBM ? 3 : 1); a.act_limit = swiglu_limit; a.act_alpha = alpha;
    if (m.qs) pfh_launch<2, 4, 1, 1>(a, mt, st); else if (kr_pfr_try_launch(a, mt, st) != 0) pfh_launch<2, 4, 1>(a, mt, st);
    const int cpl = I <= 512 ? 1 : (I <= 1024 ? 2 : 4);
#define KR_ROWW(C_) hipLaunchKernelGGL((kr_pfh_act_wave_kernel<KR_ACT_NONE, C_>), dim3((rows + 3) / 4), dim3(256), 0, st, (const float*)gu, rows, I, I, 0.0f, 0.0f, h_out, h_mul, h_sums32)
    if (rows > 0) { if (cpl == 1) KR_ROWW(1); else if (cpl == 2) KR_ROWW(2); else KR_ROWW(4); }
#undef KR_ROWW
}
void kr_launch_pfh_gemm_multi(const KrMatDev* mats, float* const* outs, const int* out_lds, int n, const uint16_t* a_h, const float* a_mul, int M, hipStream_t st) {
    KrPfGemmHArgs a{};
    a.m = mats[0]; a.out = outs[0]; a.out_ld = out_lds[0]; a.a = a_h; a.a_mul = a_mul; a.topk = 1; a.single_expert = 1; a.total_rows = M; a.n_extra = n - 1; a.run = 1;
    for (int i = 1; i < n; i++) { a.mx[i - 1] = mats[i]; a.outx[i - 1] = outs[i]; a.out_ldx[i - 1] = out_lds[i]; }
    pfh_dispatch(a, (M + PFH_BM - 1) / PFH_BM, st);
}
