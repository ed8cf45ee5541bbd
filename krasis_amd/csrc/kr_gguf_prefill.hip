// kr_gguf_prefill.hip -- native GGUF blocks (Q4_K, Q8_0) on the int8-MFMA grouped GEMM: the prompt-pass form of moe_forward_gguf
// (src/moe.rs:990-1110 -> expert_forward_gguf, src/gguf_kernels.rs:690-756 -> matvec_q4_k_avx2 :271-370, matvec_q8_0_avx2 :379-432).
//
// north_star: "Q4_K/Q8_0 super-block dequant staged in LDS feeding int8 MFMA for the prefill expert grouped-GEMM".  A k stage is one
// Q4_K super-block (256 weights: the 128 quant bytes of every column are staged RAW in LDS, the 16-byte header is decoded once per stage
// into d*sc_j / dmin*mn_j tables) or eight Q8_0 blocks; one 32-wide sub-block = one v_mfma_i32_32x32x32_i8 per activation digit plane:
//      a (i16) = AH*256 + (AL' + 128)      sum_k q*a = 256 * mfma(AH, q) + mfma(AL', q) + 128 * sum_k q         (exact i32)
// and the per-sub-block epilogue is the reference's:  out = fma(f32(isum), (d * sc_j) * a_scale_j, out);  corr += ((dmin * mn_j) * a_scale_j) * f32(asum_j);
// y = out - corr.
//
// NUMERICS (stated deviation): the AVX2 kernel keeps EIGHT f32 accumulators per output row (one per SIMD lane, each fed by the
// 4 elements {2l, 2l+1, 16+2l, 17+2l} of a sub-block) and adds them at the end (hsum); reproducing that on a matrix core would need 8
// masked MFMAs per sub-block (a K = 4 contraction inside a K = 32 instruction: 8x the work).  Here the integer sum of the WHOLE sub-block
// is formed exactly and ONE f32 chain per output runs over the sub-blocks in order: same integers, same scale products, a different
// (shorter) f32 summation order.  Result: within ~1e-6 relative of kr_moe_forward on the same layer (tests/test_gguf_gpu.py states the
// tolerance); the bit-exact form of these formats remains the streaming kernels of kr_gguf.hip.
//
// k order inside a sub-block: the HBM lane records (kr_gguf.hip header) hold bytes {2l, 2l+1, 16+2l, 17+2l} for l = 0..7, i.e. position
// p = 4l + i <-> element e(p); the activation digit planes are written in the same order by the quantizers below, so records are copied
// to LDS without a byte shuffle (the MFMA sums over k in any order as long as A and B agree).
#include "kr_lds_optin.h"
#include "kr_device.h"
#include "kr_libm.h"
#include "kr_gguf.h"
#include "kr_prefill.h"
#include <hip/hip_fp16.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define GPF_BM 64
#define GPF_BN 128
#define GPF_KS 256
#define GPF_LDA (GPF_KS + 16)

__device__ __forceinline__ float gpf_f16(uint32_t bits16) { return __half2float(__ushort_as_half((uint16_t)bits16)); }

// ------------------------------------------------------------------------------------------
// per (column, sub-block) sums of the quants (the 128 * sum_k q term): built once per weight set
//   Q4_K: ws[tile][block][row r] = 8 x u16 (nibble sums of sub-blocks 0..7)      Q8_0: ws[tile][block group][row r] = 4 x i16
// grid (tiles, units, experts), 64 threads = 8 rows x 8 lanes
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) kr_gpf_wsum_kernel(const GgMat m, char* ws, size_t ws_stride) {
    const int tile = blockIdx.x, unit = blockIdx.y, e = blockIdx.z, lane = threadIdx.x, r = lane >> 3, l = lane & 7;
    const int units = m.type == GG_Q4_K ? m.K / 256 : (m.K / 32 + 3) / 4;
    const u32x4 w = *(reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(m.q) + (size_t)e * m.q_stride) + ((size_t)tile * units + unit) * 64 + lane);
    const uint32_t wj[4] = {w.x, w.y, w.z, w.w};
    char* dst = ws + (size_t)e * ws_stride;
    if (m.type == GG_Q4_K) {
        int s[8];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            s[2 * j] = kr_red8_add_i32((int)__builtin_amdgcn_udot4(wj[j] & 0x0F0F0F0Fu, 0x01010101u, 0u, false));
            s[2 * j + 1] = kr_red8_add_i32((int)__builtin_amdgcn_udot4((wj[j] >> 4) & 0x0F0F0F0Fu, 0x01010101u, 0u, false));
        }
        if (l == 0) {
            u32x4 o = {(uint32_t)s[0] | ((uint32_t)s[1] << 16), (uint32_t)s[2] | ((uint32_t)s[3] << 16), (uint32_t)s[4] | ((uint32_t)s[5] << 16), (uint32_t)s[6] | ((uint32_t)s[7] << 16)};
            *reinterpret_cast<u32x4*>(dst + (((size_t)tile * units + unit) * 8 + r) * 16) = o;
        }
    } else {
        int s[4];
#pragma unroll
        for (int u = 0; u < 4; u++) s[u] = kr_red8_add_i32(__builtin_amdgcn_sdot4((int)wj[u], 0x01010101, 0, false));
        if (l == 0) {
            u32x2 o = {((uint32_t)s[0] & 0xFFFFu) | ((uint32_t)s[1] << 16), ((uint32_t)s[2] & 0xFFFFu) | ((uint32_t)s[3] << 16)};
            *reinterpret_cast<u32x2*>(dst + (((size_t)tile * units + unit) * 8 + r) * 8) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------
// activation quantizers: quantize_bf16_to_int16 / quantize_f32_to_int16 (gguf_kernels.rs:110-172) per 32 elements, stored as two int8 digit
// planes in the lane-record k order + f32 scale + f32(sum) per sub-block.  grid (rows), one thread per 8 elements (4 threads = a sub-block)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void gpf_quant_store(const float (&v)[8], int c, int8_t* hi, int8_t* lo, float* sc, float* sm) {
    float mx = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) mx = fmaxf(mx, fabsf(v[i]));
    mx = kr_red4_max_f32(mx);
    const float scale = mx > 0.0f ? mx / 32767.0f : 1.0f, inv = mx > 0.0f ? 32767.0f / mx : 0.0f;
    int q[8]; int s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int t = (int)roundf(v[i] * inv); t = t > 32767 ? 32767 : (t < -32768 ? -32768 : t);
        q[i] = t; s += t;
    }
    s += KR_DPP(s, KR_DPP_XOR1); s += KR_DPP(s, KR_DPP_XOR2);
    const int sb = c >> 2, part = c & 3, half = part >> 1;
    uint16_t* h16 = reinterpret_cast<uint16_t*>(hi + (size_t)sb * 32); uint16_t* l16 = reinterpret_cast<uint16_t*>(lo + (size_t)sb * 32);
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int l = (part & 1) * 4 + p;           // elements (2l, 2l+1) of the first or (16+2l, 17+2l) of the second half -> positions 4l + 2*half, +1
        const int a0 = q[2 * p], a1 = q[2 * p + 1];
        h16[l * 2 + half] = (uint16_t)(((a0 >> 8) & 0xFF) | (((a1 >> 8) & 0xFF) << 8));
        l16[l * 2 + half] = (uint16_t)((((a0 & 0xFF) - 128) & 0xFF) | ((((a1 & 0xFF) - 128) & 0xFF) << 8));
    }
    if (part == 0) { sc[sb] = scale; sm[sb] = (float)s; }
}

__global__ void kr_gpf_quant_x_kernel(const uint16_t* __restrict__ x, int K, int8_t* __restrict__ xh, int8_t* __restrict__ xl, float* __restrict__ xs,
                                      float* __restrict__ xm) {
    const size_t t = blockIdx.x;
    for (int c = threadIdx.x; c < K / 8; c += blockDim.x) {
        const u32x4 r = *reinterpret_cast<const u32x4*>(x + t * K + (size_t)c * 8);
        float v[8];
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xFFFF0000u); v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xFFFF0000u);
        v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xFFFF0000u); v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xFFFF0000u);
        gpf_quant_store(v, c, xh + t * K, xl + t * K, xs + t * (K / 32), xm + t * (K / 32));
    }
}

// hidden = silu(gate) * up with libm exp (gguf_kernels.rs:733-737), then the per-32 quantization.  gu row = [gate(n) | up(n)], ld = gu_ld
__global__ void kr_gpf_act_kernel(const float* __restrict__ gu, int n, int gu_ld, int8_t* __restrict__ hh, int8_t* __restrict__ hl, float* __restrict__ hs,
                                  float* __restrict__ hm) {
    const size_t row = blockIdx.x;
    const float* g = gu + row * gu_ld;
    for (int c = threadIdx.x; c < n / 8; c += blockDim.x) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { const float gg = g[c * 8 + i]; v[i] = (gg / (1.0f + kr_expf(-gg))) * g[n + c * 8 + i]; }
        gpf_quant_store(v, c, hh + row * n, hl + row * n, hs + row * (n / 32), hm + row * (n / 32));
    }
}

// ------------------------------------------------------------------------------------------
// grouped GEMM: 64 rows (tokens routed to one expert) x 128 columns (weight rows) per workgroup of 4 waves, one 256-k stage in LDS
// ------------------------------------------------------------------------------------------
struct GpfGemmArgs {
    GgMat m; const char* ws; size_t ws_stride;
    const int8_t* a_hi; const int8_t* a_lo; const float* a_scale; const float* a_sum;    // digits [rows][K], scale / f32(sum) [rows][K/32]
    const int* row_pair; int topk; int gather_tokens;
    const int* tile_expert; const int* tile_row0; const int* tile_rows; const int* n_tiles;
    float* out; int out_ld; int col_off;
    int single_expert; int total_rows;
};

template <int TYPE>
__global__ void __launch_bounds__(256, 2) kr_gpf_gemm_kernel(const GpfGemmArgs a) {
    constexpr bool Q4K = TYPE == GG_Q4_K;
    constexpr int LDB = (Q4K ? 128 : 256) + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int8_t* As_hi = reinterpret_cast<int8_t*>(smem);                        // [64][LDA]
    int8_t* As_lo = As_hi + GPF_BM * GPF_LDA;
    char* Bs = reinterpret_cast<char*>(As_lo + GPF_BM * GPF_LDA);            // [128 cols][LDB]: Q4_K [chunk jj][lane l][4 B]; Q8_0 [block][lane l][4 B]
    float* As_sc = reinterpret_cast<float*>(Bs + GPF_BN * LDB);              // [8][64]  a_scale of the stage's sub-blocks
    float* As_sm = As_sc + 8 * GPF_BM;                                       // [8][64]  f32(sum)
    float* Dsc = As_sm + 8 * GPF_BM;                                         // [8][128] d * sc_j      (Q8_0: d_j)
    float* Dmn = Dsc + 8 * GPF_BN;                                           // [8][128] dmin * mn_j   (Q4_K only)
    int* Wq = reinterpret_cast<int*>(Dmn + 8 * GPF_BN);                      // [8][128] 128 * sum_k q
    int* row_src = Wq + 8 * GPF_BN;                                          // [64]

    const GgMat& m = a.m;
    const int ncb = (m.N + GPF_BN - 1) / GPF_BN, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int mt = (slot / ncb) * 8 + xcd, cb = slot % ncb;                  // all column blocks of a row tile share one XCD's L2
    int expert, row0, rows;
    if (a.single_expert) { expert = 0; row0 = mt * GPF_BM; rows = a.total_rows - row0 < GPF_BM ? a.total_rows - row0 : GPF_BM; if (rows <= 0) return; }
    else { if (mt >= a.n_tiles[0]) return; expert = a.tile_expert[mt]; row0 = a.tile_row0[mt]; rows = a.tile_rows[mt]; }
    const int n0 = cb * GPF_BN, K = m.K;
    const int nsub = K / 32;                                                 // sub-blocks (Q4_K) / blocks (Q8_0) in a row
    const int units = Q4K ? K / 256 : (nsub + 3) / 4;                        // HBM record units per row tile: blocks / groups of 4 blocks
    const char* wq = reinterpret_cast<const char*>(m.q) + (size_t)expert * m.q_stride;
    const char* wh = reinterpret_cast<const char*>(m.h) + (size_t)expert * m.h_stride;
    const char* wws = a.ws + (size_t)expert * a.ws_stride;

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    if (tid < GPF_BM) {
        int src = -1;
        if (tid < rows) {
            if (a.single_expert) src = row0 + tid;
            else { const int pair = a.row_pair[row0 + tid]; src = a.gather_tokens ? pair / a.topk : row0 + tid; }
        }
        row_src[tid] = src;
    }
    __syncthreads();

    float outv[2][16], corr[2][16];
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int r = 0; r < 16; r++) { outv[s][r] = 0.0f; corr[s][r] = 0.0f; }
    const int n31 = lane & 31, khalf = lane >> 5, cw = wave * 32 + n31;     // this lane's column inside the tile
    const int ar = tid >> 2, aq = tid & 3;                                  // activation scales / sums: 4 threads per row
    const int arow = tid >> 4, aseg = (tid >> 3) & 1, achk = tid & 7;       // activation digits: whole lines (below)
    const int nst = (K + GPF_KS - 1) / GPF_KS;

    u32x4 pvh[4], pvl[4], pbw[Q4K ? 4 : 8], phd, pws;
    float2 pasc, pasm;
    auto load_stage = [&](int st) {
        {   // A digits: whole 128-byte lines per request (8 lanes x 16 B; load j of thread tid = row (tid >> 4) + 16 j, line (tid >> 3) & 1, chunk tid & 7 --
            // the grouped INT4 GEMM's finding, kr_prefill_gemm2.inc), never masked: a tile row past `rows` reads row 0 and is not stored, a line past K
            // re-reads line 0 of the segment and meets activation scale 0 / weight blocks that are zero (the scale / sum loads below keep their guards)
            const int kvalid = K - st * GPF_KS;
            const int so = aseg * 128 + achk * 16 < kvalid ? aseg * 128 + achk * 16 : 0;     // (per chunk: K % 128 may be 64, nothing is read past a row's end)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int srcj = row_src[arow + 16 * j];
                const size_t go = (size_t)(srcj < 0 ? 0 : srcj) * K + (size_t)st * GPF_KS + so;
                pvh[j] = *reinterpret_cast<const u32x4*>(a.a_hi + go);
                pvl[j] = *reinterpret_cast<const u32x4*>(a.a_lo + go);
            }
            const int src = row_src[ar];
            pasc = make_float2(0.0f, 0.0f); pasm = make_float2(0.0f, 0.0f);
            int aqv = aq;
            asm volatile("" : "+v"(aqv));      // keeps the widened per-lane offset out of the loop-invariant set (the Q8_0 form spilled it: 256 registers, two workgroups per CU)
            const int sb0 = st * 8 + aqv * 2;
            if (src >= 0 && sb0 < nsub) {      // nsub is even for every supported K (K % 64 == 0 is checked on the host)
                pasc = *reinterpret_cast<const float2*>(a.a_scale + (size_t)src * nsub + sb0);
                pasm = *reinterpret_cast<const float2*>(a.a_sum + (size_t)src * nsub + sb0);
            }
        }
        // B lane records: 128 columns x 8 lanes (x 2 block groups for Q8_0), 16 B each
#pragma unroll
        for (int j = 0; j < (Q4K ? 4 : 8); j++) {
            const int rec = tid + (j & 3) * 256, t8 = rec >> 6, ln = rec & 63, c = ln >> 3, gcol = n0 + t8 * 8 + c;
            const int unit = Q4K ? st : st * 2 + (j >> 2);
            u32x4 w = {0, 0, 0, 0};
            if (gcol < m.N && unit < units) w = kr_ldg_nt(reinterpret_cast<const u32x4*>(wq + (((size_t)(gcol >> 3) * units + unit) * 64 + ln) * 16));
            pbw[j] = w;
        }
        phd = u32x4{0, 0, 0, 0}; pws = u32x4{0, 0, 0, 0};
        if (tid < GPF_BN) {   // headers + quant sums of this thread's column
            const int gcol = n0 + tid;
            if (gcol < m.N) {
                if (Q4K) {
                    phd = *reinterpret_cast<const u32x4*>(wh + (((size_t)(gcol >> 3) * units + st) * 8 + (gcol & 7)) * 16);
                    pws = *reinterpret_cast<const u32x4*>(wws + (((size_t)(gcol >> 3) * units + st) * 8 + (gcol & 7)) * 16);
                } else {
#pragma unroll
                    for (int u2 = 0; u2 < 2; u2++) {
                        const int unit = st * 2 + u2;
                        if (unit < units) {
                            const u32x2 hd = *reinterpret_cast<const u32x2*>(wh + (((size_t)(gcol >> 3) * units + unit) * 8 + (gcol & 7)) * 8);
                            const u32x2 wsv = *reinterpret_cast<const u32x2*>(wws + (((size_t)(gcol >> 3) * units + unit) * 8 + (gcol & 7)) * 8);
                            if (u2 == 0) { phd.x = hd.x; phd.y = hd.y; pws.x = wsv.x; pws.y = wsv.y; } else { phd.z = hd.x; phd.w = hd.y; pws.z = wsv.x; pws.w = wsv.y; }
                        }
                    }
                }
            }
        }
    };
    auto commit_stage = [&]() {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            *reinterpret_cast<u32x4*>(As_hi + (arow + 16 * j) * GPF_LDA + aseg * 128 + achk * 16) = pvh[j];
            *reinterpret_cast<u32x4*>(As_lo + (arow + 16 * j) * GPF_LDA + aseg * 128 + achk * 16) = pvl[j];
        }
        As_sc[(aq * 2) * GPF_BM + ar] = pasc.x; As_sc[(aq * 2 + 1) * GPF_BM + ar] = pasc.y;
        As_sm[(aq * 2) * GPF_BM + ar] = pasm.x; As_sm[(aq * 2 + 1) * GPF_BM + ar] = pasm.y;
#pragma unroll
        for (int j = 0; j < (Q4K ? 4 : 8); j++) {
            const int rec = tid + (j & 3) * 256, t8 = rec >> 6, ln = rec & 63, c = ln >> 3, l8 = ln & 7;
            char* base = Bs + (t8 * 8 + c) * LDB + (Q4K ? 0 : (j >> 2) * 128) + l8 * 4;       // piece u of the record -> [chunk / block u][lane l]
            const u32x4 w = pbw[j];
            *reinterpret_cast<uint32_t*>(base) = w.x; *reinterpret_cast<uint32_t*>(base + 32) = w.y;
            *reinterpret_cast<uint32_t*>(base + 64) = w.z; *reinterpret_cast<uint32_t*>(base + 96) = w.w;
        }
        if (tid < GPF_BN) {
            if (Q4K) {
                const float d = gpf_f16(phd.x & 0xFFFFu), dmin = gpf_f16(phd.x >> 16);
                const uint32_t wsw[4] = {pws.x, pws.y, pws.z, pws.w};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t w3[3] = {phd.y, phd.z, phd.w};
                    auto B = [&](int i) -> uint32_t { return (w3[i >> 2] >> ((i & 3) * 8)) & 0xFFu; };
                    int sc, mn;                                                    // get_scale_min_k4 (gguf_kernels.rs:640-648)
                    if (j < 4) { sc = (int)(B(j) & 63u); mn = (int)(B(j + 4) & 63u); }
                    else { sc = (int)((B(j + 4) & 0xFu) | ((B(j - 4) >> 6) << 4)); mn = (int)((B(j + 4) >> 4) | ((B(j) >> 6) << 4)); }
                    Dsc[j * GPF_BN + tid] = d * (float)sc; Dmn[j * GPF_BN + tid] = dmin * (float)mn;
                    Wq[j * GPF_BN + tid] = (int)((wsw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu) << 7;
                }
            } else {
                const uint32_t hw[4] = {phd.x, phd.y, phd.z, phd.w}, wsw[4] = {pws.x, pws.y, pws.z, pws.w};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    Dsc[j * GPF_BN + tid] = gpf_f16((hw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu);
                    Wq[j * GPF_BN + tid] = ((int)(int16_t)((wsw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu)) << 7;
                }
            }
        }
    };

    load_stage(0);
    for (int st = 0; st < nst; st++) {
        commit_stage();
        if (st + 1 < nst) load_stage(st + 1);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (st * 8 + j < nsub) {
                v16i acc_hi[2], acc_lo[2];
                const int w128 = Wq[j * GPF_BN + cw];
#pragma unroll
                for (int s = 0; s < 2; s++)
#pragma unroll
                    for (int r = 0; r < 16; r++) { acc_hi[s][r] = 0; acc_lo[s][r] = w128; }
                v4i bf;
                if (Q4K) {
                    const u32x4 raw = *reinterpret_cast<const u32x4*>(Bs + cw * LDB + (j >> 1) * 32 + khalf * 16);
                    const int sh = (j & 1) * 4;
                    bf[0] = (int)((raw.x >> sh) & 0x0F0F0F0Fu); bf[1] = (int)((raw.y >> sh) & 0x0F0F0F0Fu);
                    bf[2] = (int)((raw.z >> sh) & 0x0F0F0F0Fu); bf[3] = (int)((raw.w >> sh) & 0x0F0F0F0Fu);
                } else bf = *reinterpret_cast<const v4i*>(Bs + cw * LDB + j * 32 + khalf * 16);
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    const v4i ah = *reinterpret_cast<const v4i*>(As_hi + (s * 32 + n31) * GPF_LDA + j * 32 + khalf * 16);
                    const v4i al = *reinterpret_cast<const v4i*>(As_lo + (s * 32 + n31) * GPF_LDA + j * 32 + khalf * 16);
                    acc_hi[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ah, bf, acc_hi[s], 0, 0, 0);
                    acc_lo[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(al, bf, acc_lo[s], 0, 0, 0);
                }
                const float dsc = Dsc[j * GPF_BN + cw], dmn = Q4K ? Dmn[j * GPF_BN + cw] : 0.0f;
#pragma unroll
                for (int s = 0; s < 2; s++)
#pragma unroll
                    for (int r4 = 0; r4 < 4; r4++) {
                        const int rowb = s * 32 + 8 * r4 + 4 * khalf;         // accumulators r4*4 .. +3 are rows rowb .. rowb+3
                        const float4 asc = *reinterpret_cast<const float4*>(As_sc + j * GPF_BM + rowb);
                        const float4 asm4 = *reinterpret_cast<const float4*>(As_sm + j * GPF_BM + rowb);
                        const float as4[4] = {asc.x, asc.y, asc.z, asc.w}, am4[4] = {asm4.x, asm4.y, asm4.z, asm4.w};
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int r = r4 * 4 + i;
                            const int isum = (acc_hi[s][r] << 8) + acc_lo[s][r];
                            outv[s][r] = __builtin_fmaf((float)isum, dsc * as4[i], outv[s][r]);          // (d * sc) * a_scale
                            if (Q4K) corr[s][r] += (dmn * as4[i]) * am4[i];                               // ((dmin * mn) * a_scale) * f32(sum)
                        }
                    }
            }
        }
        __syncthreads();
    }
    int lane_e;      // the lane id again (v_mbcnt needs no live register): column and lane half of the store phase are not carried across the k loop
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int col = n0 + wave * 32 + (lane_e & 31), khalf_e = lane_e >> 5;
    if (col < m.N) {
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = s * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf_e;
                if (row < rows) a.out[(size_t)(row0 + row) * a.out_ld + a.col_off + col] = Q4K ? outv[s][r] - corr[s][r] : outv[s][r];
            }
    }
}

// ------------------------------------------------------------------------------------------
// synthetic GGUF experts (SURVEY 8d defines the distribution the reference's generator lacks): quant bytes and the 12 packed scale bytes
// are raw hash words, d = f16((0.005 + u * 0.045) / 63), dmin = f16(8 d); Q8_0: d = f16((0.005 + u * 0.045) / 127)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t gpf_mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
__global__ void kr_gpf_fill_q_kernel(uint32_t* q, size_t n_words, uint64_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) q[i] = (uint32_t)gpf_mix(seed + i);
}
__global__ void kr_gpf_fill_h_kernel(char* h, size_t n_rec, int type, uint64_t seed) {   // one header record per thread
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t r0 = gpf_mix(seed ^ (i * 3 + 0)), r1 = gpf_mix(seed ^ (i * 3 + 1)), r2 = gpf_mix(seed ^ (i * 3 + 2));
        auto dval = [](uint32_t u, float div) { return (0.005f + ((float)u / 4294967295.0f) * 0.045f) / div; };
        if (type == GG_Q4_K) {
            const float d = __half2float(__float2half(dval((uint32_t)r0, 63.0f)));
            const uint32_t db = __half_as_ushort(__float2half(d)), mb = __half_as_ushort(__float2half(d * 8.0f));
            u32x4 o = {db | (mb << 16), (uint32_t)(r0 >> 32), (uint32_t)r1, (uint32_t)(r1 >> 32)};
            *reinterpret_cast<u32x4*>(h + i * 16) = o;
        } else {   // Q8_0: 4 x f16 d
            const uint32_t a = __half_as_ushort(__float2half(dval((uint32_t)r0, 127.0f))), b = __half_as_ushort(__float2half(dval((uint32_t)(r0 >> 32), 127.0f)));
            const uint32_t c = __half_as_ushort(__float2half(dval((uint32_t)r2, 127.0f))), d2 = __half_as_ushort(__float2half(dval((uint32_t)(r2 >> 32), 127.0f)));
            u32x2 o = {a | (b << 16), c | (d2 << 16)};
            *reinterpret_cast<u32x2*>(h + i * 8) = o;
        }
    }
}
void kr_launch_gpf_fill_synth(void* q, size_t q_bytes, void* h, size_t h_bytes, int type, uint64_t seed, hipStream_t st) {
    hipLaunchKernelGGL(kr_gpf_fill_q_kernel, dim3(2048), dim3(256), 0, st, (uint32_t*)q, q_bytes / 4, seed);
    hipLaunchKernelGGL(kr_gpf_fill_h_kernel, dim3(512), dim3(256), 0, st, (char*)h, h_bytes / (type == GG_Q4_K ? 16 : 8), type, seed ^ 0x5DEECE66Dull);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
bool kr_gpf_type_supported(int type, int K) { return (type == GG_Q4_K && K % 256 == 0) || (type == GG_Q8_0 && K % 64 == 0); }
size_t kr_gpf_ws_bytes(int type, int K, int N) {
    const size_t nt = (size_t)(N + 7) / 8;
    return type == GG_Q4_K ? nt * (K / 256) * 8 * 16 : nt * (((size_t)K / 32 + 3) / 4) * 8 * 8;
}
void kr_launch_gpf_wsum(const GgMat& m, int n_experts, void* ws, size_t ws_stride, hipStream_t st) {
    const int units = m.type == GG_Q4_K ? m.K / 256 : (m.K / 32 + 3) / 4;
    hipLaunchKernelGGL(kr_gpf_wsum_kernel, dim3((m.N + 7) / 8, units, n_experts), dim3(64), 0, st, m, (char*)ws, ws_stride);
}
void kr_launch_gpf_quant_x(const uint16_t* x, int M, int K, int8_t* xh, int8_t* xl, float* xs, float* xm, hipStream_t st) {
    const int thr = K / 8 < 1024 ? K / 8 : 1024;
    hipLaunchKernelGGL(kr_gpf_quant_x_kernel, dim3(M), dim3(thr), 0, st, x, K, xh, xl, xs, xm);
}
void kr_launch_gpf_act(const float* gu, int rows, int n, int gu_ld, int8_t* hh, int8_t* hl, float* hs, float* hm, hipStream_t st) {
    const int thr = n / 8 < 1024 ? n / 8 : 1024;
    hipLaunchKernelGGL(kr_gpf_act_kernel, dim3(rows), dim3(thr), 0, st, gu, n, gu_ld, hh, hl, hs, hm);
}
void kr_launch_gpf_gemm(const GgMat& m, const void* ws, size_t ws_stride, const int8_t* a_hi, const int8_t* a_lo, const float* a_scale, const float* a_sum,
                        const KrPfSort* sort, int topk, int gather_tokens, int max_tiles, int single_expert_rows, float* out, int out_ld, int col_off,
                        hipStream_t st) {
    GpfGemmArgs a{};
    a.m = m; a.ws = (const char*)ws; a.ws_stride = ws_stride; a.a_hi = a_hi; a.a_lo = a_lo; a.a_scale = a_scale; a.a_sum = a_sum; a.topk = topk;
    a.gather_tokens = gather_tokens;
    if (sort) { a.row_pair = sort->row_pair; a.tile_expert = sort->tile_expert; a.tile_row0 = sort->tile_row0; a.tile_rows = sort->tile_rows; a.n_tiles = sort->n_tiles; }
    a.out = out; a.out_ld = out_ld; a.col_off = col_off; a.single_expert = single_expert_rows > 0; a.total_rows = single_expert_rows;
    const int mt = single_expert_rows > 0 ? (single_expert_rows + GPF_BM - 1) / GPF_BM : max_tiles;
    const bool q4k = m.type == GG_Q4_K;
    const size_t lds = (size_t)2 * GPF_BM * GPF_LDA + (size_t)GPF_BN * ((q4k ? 128 : 256) + 16) + (size_t)(2 * 8 * GPF_BM + 3 * 8 * GPF_BN + GPF_BM) * 4;
    (void)kr_lds_optin(q4k ? (const void*)kr_gpf_gemm_kernel<GG_Q4_K> : (const void*)kr_gpf_gemm_kernel<GG_Q8_0>, 96 * 1024);   // per (kernel, device)
    dim3 grid(((mt + 7) / 8) * 8 * ((m.N + GPF_BN - 1) / GPF_BN));
    if (q4k) hipLaunchKernelGGL(kr_gpf_gemm_kernel<GG_Q4_K>, grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL(kr_gpf_gemm_kernel<GG_Q8_0>, grid, dim3(256), lds, st, a);
}

// ------------------------------------------------------------------------------------------
// Q4_K -> the tolerance GEMM's operand form (kr_prefill_h.hip, G = 1; KrMatDev::qs / qo), built once per weight set on first use:
//   nibbles n of every super-block in the INT4 lane-tiled layout of kr_kernels.h (k ascending inside a packed word, group pair = one super-block),
//   qs[tile][block][col][j] = f16((d * sc_j) / 4),   qo[tile][block][col][j] = f16(16 * (8 * d * sc_j - dmin * mn_j))     (gguf.rs:666-732)
// so that  w = d sc_j n - dmin mn_j = (d sc_j)(n - 8) + qo / 16.  The scale product d * sc_j is exact in f32 and rounds ONCE to f16 here (2^-12
// relative per sub-block: the stated cost of this form on top of the f16 activations).
// grid (tiles, blocks, experts), 64 threads = 8 rows x 8 lanes of the source tile
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) kr_gq_repack_kernel(const GgMat m, char* q_out, size_t q_stride, uint16_t* qs_out, uint16_t* qo_out, size_t qs_stride, int tile_off, int tiles_out) {
    __shared__ uint8_t qs[8][128];
    const int tile = blockIdx.x, blk = blockIdx.y, e = blockIdx.z, lane = threadIdx.x, r = lane >> 3, l = lane & 7;
    const int blocks = m.K / 256;
    const char* qsrc = reinterpret_cast<const char*>(m.q) + (size_t)e * m.q_stride;
    const char* hsrc = reinterpret_cast<const char*>(m.h) + (size_t)e * m.h_stride;
    const u32x4 w = *(reinterpret_cast<const u32x4*>(qsrc) + ((size_t)tile * blocks + blk) * 64 + lane);
    const uint32_t wj[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int p = 0; p < 4; p++) qs[r][32 * j + 2 * l + (p & 1) + 16 * (p >> 1)] = (uint8_t)(wj[j] >> (8 * p));     // record byte p of chunk j = qs[32j + e], e = 2l + (p & 1) + 16 (p >> 1)
    __syncthreads();
    // output record of (column r, lane l): packed words i = 2l, 2l + 1 of both 128-groups of the super-block
    uint32_t ow[4];
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
        for (int ii = 0; ii < 2; ii++) {
            const int i = 2 * l + ii;
            uint32_t word = 0;
#pragma unroll
            for (int mm = 0; mm < 8; mm++) {
                const int k = g * 128 + 8 * i + mm, sb = k >> 5, el = k & 31;
                const uint32_t b = qs[r][32 * (sb >> 1) + el];
                word |= ((sb & 1) ? (b >> 4) : (b & 0xFu)) << (4 * mm);
            }
            ow[g * 2 + ii] = word;
        }
    char* qdst = q_out + (size_t)e * q_stride;
    *(reinterpret_cast<u32x4*>(qdst) + ((size_t)(tile_off + tile) * blocks + blk) * 64 + lane) = u32x4{ow[0], ow[1], ow[2], ow[3]};
    // scale / offset tables: lane l of row r takes sub-block j = l
    const u32x4 hd = *(reinterpret_cast<const u32x4*>(hsrc) + ((size_t)tile * blocks + blk) * 8 + r);
    const float d = gpf_f16(hd.x & 0xFFFFu), dmin = gpf_f16(hd.x >> 16);
    const uint32_t w3[3] = {hd.y, hd.z, hd.w};
    auto B = [&](int i) -> uint32_t { return (w3[i >> 2] >> ((i & 3) * 8)) & 0xFFu; };
    const int j = l;
    int sc, mn;                                                    // get_scale_min_k4 (gguf_kernels.rs:640-648)
    if (j < 4) { sc = (int)(B(j) & 63u); mn = (int)(B(j + 4) & 63u); }
    else { sc = (int)((B(j + 4) & 0xFu) | ((B(j - 4) >> 6) << 4)); mn = (int)((B(j + 4) >> 4) | ((B(j) >> 6) << 4)); }
    const float sj = d * (float)sc, oj = dmin * (float)mn;
    const size_t ti = (((size_t)(tile_off + tile) * blocks + blk) * 8 + r) * 8 + j;
    const _Float16 hs = (_Float16)(sj * 0.25f), ho = (_Float16)(16.0f * (8.0f * sj - oj));
    (qs_out + (size_t)e * (qs_stride / 2))[ti] = __builtin_bit_cast(uint16_t, hs);
    (qo_out + (size_t)e * (qs_stride / 2))[ti] = __builtin_bit_cast(uint16_t, ho);
    (void)tiles_out;
}
void kr_launch_gq_repack(const GgMat& m, int n_experts, void* q_out, size_t q_stride, void* qs_out, void* qo_out, size_t qs_stride, int tile_off, int tiles_out, hipStream_t st) {
    hipLaunchKernelGGL(kr_gq_repack_kernel, dim3(m.N / 8, m.K / 256, n_experts), dim3(64), 0, st, m, (char*)q_out, q_stride, (uint16_t*)qs_out, (uint16_t*)qo_out, qs_stride, tile_off,
                       tiles_out);
}

// ------------------------------------------------------------------------------------------
// Q8_0 -> the tolerance GEMM's INT8 operand form (kr_prefill_h.hip, BITS = 8, G = 1): the int8 quants of every 128-k group in the INT8 lane-tiled
// layout of kr_kernels.h (lane record (column c, l8) = k 16 l8 .. 16 l8 + 15 of the group, natural order) and ONE f16 scale per 32-wide block,
// qs[tile][group pair][col][j] = f16(16 d_j), j = 0..7 over the pair's 256 k (0 for a group past K: the half-empty last stage of an odd group
// count).  w = d q exactly; 16 d rounds once to f16 (d is an f16: exact unless it overflows at d > 4094, which no Q8_0 block has).
// grid (tiles, groups of the PADDED pair count, experts), 64 threads = 8 rows x 8 lanes of the source tile (kr_gguf.hip layout)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) kr_gq8_repack_kernel(const GgMat m, char* q_out, size_t q_stride, uint16_t* qs_out, size_t qs_stride, int tile_off) {
    __shared__ uint8_t qb[8][128];
    const int tile = blockIdx.x, g = blockIdx.y, e = blockIdx.z, lane = threadIdx.x, r = lane >> 3, l = lane & 7;
    const int ng = m.K / 128, ngp = (ng + 1) / 2, nbg = (m.K / 32 + 3) / 4;
    uint16_t* sdst = qs_out + (size_t)e * (qs_stride / 2) + ((((size_t)(tile_off + tile) * ngp + (g >> 1)) * 8 + r) * 8 + (g & 1) * 4);
    if (g >= ng) { if (l < 4) sdst[l] = 0; return; }          // uniform per workgroup
    const char* qsrc = reinterpret_cast<const char*>(m.q) + (size_t)e * m.q_stride;
    const char* hsrc = reinterpret_cast<const char*>(m.h) + (size_t)e * m.h_stride;
    const u32x4 w = *(reinterpret_cast<const u32x4*>(qsrc) + ((size_t)tile * nbg + g) * 64 + lane);
    const uint32_t wu[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int p = 0; p < 4; p++) qb[r][32 * u + 2 * l + (p & 1) + 16 * (p >> 1)] = (uint8_t)(wu[u] >> (8 * p));   // bytes {2l, 2l+1, 16+2l, 17+2l} of block u
    __syncthreads();
    const u32x4 o = *reinterpret_cast<const u32x4*>(&qb[r][16 * l]);
    *(reinterpret_cast<u32x4*>(q_out + (size_t)e * q_stride) + ((size_t)(tile_off + tile) * ng + g) * 64 + lane) = o;
    if (l < 4) {
        const uint16_t db = *(reinterpret_cast<const uint16_t*>(hsrc) + (((size_t)tile * nbg + g) * 8 + r) * 4 + l);
        const _Float16 hs = (_Float16)(gpf_f16(db) * 16.0f);
        sdst[l] = __builtin_bit_cast(uint16_t, hs);
    }
}
void kr_launch_gq8_repack(const GgMat& m, int n_experts, void* q_out, size_t q_stride, void* qs_out, size_t qs_stride, int tile_off, hipStream_t st) {
    const int ng = m.K / 128, ngp = (ng + 1) / 2;
    hipLaunchKernelGGL(kr_gq8_repack_kernel, dim3(m.N / 8, 2 * ngp, n_experts), dim3(64), 0, st, m, (char*)q_out, q_stride, (uint16_t*)qs_out, qs_stride, tile_off);
}
