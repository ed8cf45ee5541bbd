// kr_mla_flash.hip -- prompt-pass MLA attention on the matrix cores (FAST / tolerance mode, kr_decode_set_attention_mode).
//
// The exact MLA prompt pass runs the decode launches with a token dimension: every (head, token) workgroup walks the whole latent cache
// (position x (kv_lora_rank + rope) elements), i.e. the cache is re-read nh x tokens times -- 2.2 k tok/s at 8192 tokens on the V2-Lite shape.
// In the absorbed form (decode.rs:2993-3252; the reference's GPU side: flashinfer MLA, attention.py:300-349) ALL heads of ALL tokens of a tile
// attend over the SAME rows:   s[h][p] = (q_abs[h] . ckv[p] + q_pe[h] . kpe[p]) * sm_scale,   out[h] = sum_p softmax(s)[p] * ckv[p]
// = flash attention with K rows [ckv | kpe] (576 dims) and V = the first kv_lora_rank dims of the same rows.  One workgroup takes 64 query
// rows (token-major: row = token * nh + head) and streams the cache once for all of them:
//   * waves (wq, wd): wq picks 32 query rows, wd one half of the kv_lora_rank output dims (the S^T tile is computed by both d-waves: with
//     K = 576 and an f32 O^T of 512 dims per row the registers of one wave cannot hold a full-width accumulator);
//   * S^T = K Q^T with v_mfma_f32_32x32x16_f16: A = staged rows [position][576] (LDS), B = the query tile (LDS, scale and log2 e folded in),
//     accumulator column = query row = lane % 32 -> lane-local online softmax (kr_attn_flash.hip);
//   * O^T += V^T P^T: A = the same latent rows written TRANSPOSED ([dim][position], two positions per dword) while the tile is staged --
//     one global fetch feeds both operands --, B = P^T from the S^T accumulators.
// q, p rounded to f16; ckv / kpe exact in f16 (FP16 or E4M3 caches).  f32 accumulation and softmax.
#include "kr_lds_optin.h"
#include "kr_device.h"
#include "kr_libm.h"
#include "kr_decode_ops.h"

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define MF_TK 32
#define MF_ROWS 64

__device__ __forceinline__ uint32_t mf_fp8x2_to_h2(uint32_t w, bool hi) {      // two E4M3 bytes -> packed f16 pair (exact): ONE v_cvt_scalef32_pk_f16_fp8 (scale 1)
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t h = hi ? __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, true) : __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, false);
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ uint32_t mf_f2h2(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b)); }

// grid (row tiles of 64 query rows); q_abs [C][nh][KLR], q_pe [C][nh][64] (f32, kr_mla_prep_kernel), caches [position][KLR] / [position][64]
// SPLIT (decode over a long cache, one token): grid (row tiles, chunks); a workgroup walks `chunk_tiles` tiles of the cache and leaves the
// un-normalised O rows and (max in log2 units, sum) per head for kr_fd_merge2_kernel (kr_attn_flash.hip); the position comes from a.step.
template <int KLR, bool FP8, bool SPLIT = false>
__global__ void __launch_bounds__(256) kr_mla_flash_kernel(const KrMlaArgs a, int C, int n_chunks = 0, int chunk_tiles = 0) {
    constexpr int RD = 64, DK = KLR + RD, KSTEPS = DK / 16, LDK = DK * 2 + 16, LDV = MF_TK * 2 + 16;
    constexpr int DBW = KLR / 64;                               // 32-dim output blocks per d-wave (two d-waves)
    constexpr int CB = FP8 ? 16 : 8;                            // dims per 16-byte global chunk
    constexpr int CPC = KLR / CB, CPRR = RD / CB, CPR = CPC + CPRR;      // chunks per row: latent part, rope part
    constexpr int NUN = (MF_TK / 2) * CPR, UPT = (NUN + 255) / 256;       // (position pair, chunk) units per tile / per thread
    extern __shared__ __attribute__((aligned(16))) char mf_smem[];
    char* Qs = mf_smem;                                         // [64 rows][LDK]        f16 (scaled)
    char* Ks = Qs + MF_ROWS * LDK;                              // [32 positions][LDK]   f16
    char* Vt = Ks + MF_TK * LDK;                                // [KLR dims][LDV]       f16, positions contiguous
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n31 = lane & 31, khalf = lane >> 5, wq = wave & 1, wd = wave >> 1;
    const int nrows = C * a.nh, row0 = blockIdx.x * MF_ROWS;
    const int pos0 = SPLIT ? a.step->pos : a.pos0;
    const int tok_first = row0 / a.nh, tok_last = min(C - 1, (row0 + MF_ROWS - 1) / a.nh);
    const int kv_end = pos0 + tok_last + 1, full_vis = pos0 + tok_first;
    const int n_tiles_all = (kv_end + MF_TK - 1) / MF_TK;
    const int tile_beg = SPLIT ? (int)blockIdx.y * chunk_tiles : 0, n_tiles = SPLIT ? min(n_tiles_all, tile_beg + chunk_tiles) : n_tiles_all;
    if (SPLIT && tile_beg >= n_tiles_all) return;               // chunk past the current length (the grid is sized for the whole cache)
    const int myrow = row0 + wq * 32 + n31;
    const bool row_ok = myrow < nrows;
    const int p_q = pos0 + (row_ok ? myrow / a.nh : 0);

    // ---- query tile -> LDS (f16, sm_scale * log2 e folded in); rows past the end are zero
    {
        const float sc = a.sm_scale * 1.4426950408889634f;
        for (int i = tid; i < MF_ROWS * (DK / 8); i += 256) {
            const int r = i / (DK / 8), c8 = i % (DK / 8), gr = row0 + r;
            u32x4 o = {0, 0, 0, 0};
            if (gr < nrows) {
                const float* src = c8 < KLR / 8 ? a.q_abs + (size_t)gr * KLR + c8 * 8 : a.q_pe + (size_t)gr * RD + (c8 - KLR / 8) * 8;
                const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
                o = u32x4{mf_f2h2(x0.x * sc, x0.y * sc), mf_f2h2(x0.z * sc, x0.w * sc), mf_f2h2(x1.x * sc, x1.y * sc), mf_f2h2(x1.z * sc, x1.w * sc)};
            }
            *reinterpret_cast<u32x4*>(Qs + r * LDK + c8 * 16) = o;
        }
    }
    v16f oacc[DBW];
#pragma unroll
    for (int db = 0; db < DBW; db++)
#pragma unroll
        for (int i = 0; i < 16; i++) oacc[db][i] = 0.0f;
    float m_run = -__builtin_inff(), l_run = 0.0f;

    const unsigned char* cc = reinterpret_cast<const unsigned char*>(a.ckv_cache);
    const unsigned char* cr = reinterpret_cast<const unsigned char*>(a.kpe_cache);
    constexpr int esz = FP8 ? 1 : 2;
    u32x4 pa[UPT], pb[UPT];
    auto load_tile = [&](int p0) {
#pragma unroll
        for (int j = 0; j < UPT; j++) {
            const int u = tid + j * 256, pp = u & 15, dc = u >> 4, p = p0 + 2 * pp;       // lanes walk the position pairs (transposed LDS writes)
            pa[j] = u32x4{0, 0, 0, 0}; pb[j] = u32x4{0, 0, 0, 0};
            if (u < NUN) {
                const unsigned char* base = dc < CPC ? cc + (size_t)dc * 16 : cr + (size_t)(dc - CPC) * 16;
                const size_t ld = (size_t)(dc < CPC ? KLR : RD) * esz;
                if (p < kv_end) pa[j] = *reinterpret_cast<const u32x4*>(base + (size_t)p * ld);
                if (p + 1 < kv_end) pb[j] = *reinterpret_cast<const u32x4*>(base + (size_t)(p + 1) * ld);
            }
        }
    };
    auto commit_tile = [&]() {
#pragma unroll
        for (int j = 0; j < UPT; j++) {
            const int u = tid + j * 256, pp = u & 15, dc = u >> 4;
            if (u < NUN) {
                constexpr int NW = FP8 ? 8 : 4;
                uint32_t ha[8], hb[8];
                if (FP8) {
                    const uint32_t wa[4] = {pa[j].x, pa[j].y, pa[j].z, pa[j].w}, wb[4] = {pb[j].x, pb[j].y, pb[j].z, pb[j].w};
#pragma unroll
                    for (int m = 0; m < 4; m++) { ha[2 * m] = mf_fp8x2_to_h2(wa[m], false); ha[2 * m + 1] = mf_fp8x2_to_h2(wa[m], true);
                                                  hb[2 * m] = mf_fp8x2_to_h2(wb[m], false); hb[2 * m + 1] = mf_fp8x2_to_h2(wb[m], true); }
                } else {
                    ha[0] = pa[j].x; ha[1] = pa[j].y; ha[2] = pa[j].z; ha[3] = pa[j].w; hb[0] = pb[j].x; hb[1] = pb[j].y; hb[2] = pb[j].z; hb[3] = pb[j].w;
                    ha[4] = ha[5] = ha[6] = ha[7] = 0; hb[4] = hb[5] = hb[6] = hb[7] = 0;
                }
                // row-major K rows (dims dc * CB ..)
                char* ka = Ks + (2 * pp) * LDK + dc * CB * 2; char* kb = ka + LDK;
                *reinterpret_cast<u32x4*>(ka) = u32x4{ha[0], ha[1], ha[2], ha[3]}; *reinterpret_cast<u32x4*>(kb) = u32x4{hb[0], hb[1], hb[2], hb[3]};
                if (FP8) { *reinterpret_cast<u32x4*>(ka + 16) = u32x4{ha[4], ha[5], ha[6], ha[7]}; *reinterpret_cast<u32x4*>(kb + 16) = u32x4{hb[4], hb[5], hb[6], hb[7]}; }
                // transposed value rows (latent part only): dims 2m, 2m+1 of the chunk, {row a, row b} in one dword
                if (dc < CPC) {
                    char* base = Vt + (size_t)(dc * CB) * LDV + pp * 4;
#pragma unroll
                    for (int m = 0; m < NW; m++) {
                        *reinterpret_cast<uint32_t*>(base + (2 * m) * LDV) = __builtin_amdgcn_perm(hb[m], ha[m], 0x05040100u);
                        *reinterpret_cast<uint32_t*>(base + (2 * m + 1) * LDV) = __builtin_amdgcn_perm(hb[m], ha[m], 0x07060302u);
                    }
                }
            }
        }
    };

    load_tile(tile_beg * MF_TK);
    for (int tile = tile_beg; tile < n_tiles; tile++) {
        const int p0 = tile * MF_TK;
        commit_tile();
        if (tile + 1 < n_tiles) load_tile(p0 + MF_TK);
        __syncthreads();
        // ---- S^T = K Q^T (32 positions x 32 query rows)
        v16f sacc;
#pragma unroll
        for (int i = 0; i < 16; i++) sacc[i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
            const v8h kf = *reinterpret_cast<const v8h*>(Ks + n31 * LDK + (16 * ks + 8 * khalf) * 2);
            const v8h qf = *reinterpret_cast<const v8h*>(Qs + (wq * 32 + n31) * LDK + (16 * ks + 8 * khalf) * 2);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf, sacc, 0, 0, 0);
            if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- online softmax of this lane's row (16 of the tile's 32 positions live here, the rest in lane ^ 32)
        const bool need_mask = p0 + MF_TK - 1 > full_vis;
        float mloc = -__builtin_inff();
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (need_mask) { const int p = p0 + (i & 3) + 8 * (i >> 2) + 4 * khalf; if (p > p_q || !row_ok) sacc[i] = -__builtin_inff(); }
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
            for (int db = 0; db < DBW; db++)
#pragma unroll
                for (int i = 0; i < 16; i++) oacc[db][i] *= alpha;
        }
        // ---- O^T += V^T P^T (this wave's half of the latent dims)
#pragma unroll
        for (int db = 0; db < DBW; db++)
#pragma unroll
            for (int kt = 0; kt < 2; kt++) {
                const char* vr = Vt + (size_t)(wd * (KLR / 2) + 32 * db + n31) * LDV + (16 * kt + 4 * khalf) * 2;
                const u32x2 v0 = *reinterpret_cast<const u32x2*>(vr), v1 = *reinterpret_cast<const u32x2*>(vr + 16);
                const u32x4 vv = {v0.x, v0.y, v1.x, v1.y};
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, vv), pf[kt], oacc[db], 0, 0, 0);
            }
        __syncthreads();
    }
    if (SPLIT) {
        if (row_ok) {
            float* out = a.fd_o + ((size_t)blockIdx.y * a.nh + myrow) * KLR + wd * (KLR / 2);          // [chunk][head][KLR]
#pragma unroll
            for (int db = 0; db < DBW; db++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++)
                    *reinterpret_cast<float4*>(out + 32 * db + 8 * g4 + 4 * khalf) = make_float4(oacc[db][4 * g4], oacc[db][4 * g4 + 1], oacc[db][4 * g4 + 2], oacc[db][4 * g4 + 3]);
            if (wd == 0 && khalf == 0) { float* ml = a.fd_ml + ((size_t)myrow * n_chunks + blockIdx.y) * 2; ml[0] = m_run; ml[1] = l_run; }
        }
        return;
    }
    if (row_ok) {
        const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
        float* out = a.attn_lat + (size_t)myrow * KLR + wd * (KLR / 2);
#pragma unroll
        for (int db = 0; db < DBW; db++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++)
                *reinterpret_cast<float4*>(out + 32 * db + 8 * g4 + 4 * khalf) =
                    make_float4(oacc[db][4 * g4] * inv, oacc[db][4 * g4 + 1] * inv, oacc[db][4 * g4 + 2] * inv, oacc[db][4 * g4 + 3] * inv);
    }
}

template <int KLR> static size_t kr_mla_flash_lds() { return (size_t)(MF_ROWS + MF_TK) * ((KLR + 64) * 2 + 16) + (size_t)KLR * (MF_TK * 2 + 16); }

// raise the dynamic-LDS window (per device, outside graph capture / before the first launch)
void kr_mla_flash_prepare() {
    (void)kr_lds_optin((const void*)kr_mla_flash_kernel<512, false, true>, kr_mla_flash_lds<512>());
    (void)kr_lds_optin((const void*)kr_mla_flash_kernel<512, true, true>, kr_mla_flash_lds<512>());
    (void)kr_lds_optin((const void*)kr_mla_flash_kernel<256, false, true>, kr_mla_flash_lds<256>());
    (void)kr_lds_optin((const void*)kr_mla_flash_kernel<256, true, true>, kr_mla_flash_lds<256>());
    (void)kr_lds_optin((const void*)kr_mla_flash_kernel<512, false>, kr_mla_flash_lds<512>());
    (void)kr_lds_optin((const void*)kr_mla_flash_kernel<512, true>, kr_mla_flash_lds<512>());
    (void)kr_lds_optin((const void*)kr_mla_flash_kernel<256, false>, kr_mla_flash_lds<256>());
    (void)kr_lds_optin((const void*)kr_mla_flash_kernel<256, true>, kr_mla_flash_lds<256>());
}
// prompt pass, n_tok tokens: a.q_abs / a.q_pe / a.attn_lat are the chunk's [n_tok][nh][.] buffers.  non-zero = geometry not covered
int kr_launch_mla_flash(const KrMlaArgs& a, int n_tok, hipStream_t st) {
    if (a.rd != 64 || (a.klr != 512 && a.klr != 256) || a.step) return 1;
    kr_mla_flash_prepare();
    dim3 grid((n_tok * a.nh + MF_ROWS - 1) / MF_ROWS);
    if (a.klr == 512) {
        if (a.kv_fp8) hipLaunchKernelGGL((kr_mla_flash_kernel<512, true>), grid, dim3(256), kr_mla_flash_lds<512>(), st, a, n_tok);
        else hipLaunchKernelGGL((kr_mla_flash_kernel<512, false>), grid, dim3(256), kr_mla_flash_lds<512>(), st, a, n_tok);
    } else {
        if (a.kv_fp8) hipLaunchKernelGGL((kr_mla_flash_kernel<256, true>), grid, dim3(256), kr_mla_flash_lds<256>(), st, a, n_tok);
        else hipLaunchKernelGGL((kr_mla_flash_kernel<256, false>), grid, dim3(256), kr_mla_flash_lds<256>(), st, a, n_tok);
    }
    return 0;
}

// decode over a long cache, FAST mode: split-KV form of the kernel above + the log-sum-exp merge of kr_attn_flash.hip.  nh <= 64 heads (one row
// tile).  non-zero = geometry not covered.  kr_mla_flash_prepare() must have run outside graph capture.
int kr_mla_flash_decode_chunk(int max_seq) { return max_seq > 16384 ? 128 : 64; }
size_t kr_mla_flash_decode_chunks(int max_seq) { const int ch = kr_mla_flash_decode_chunk(max_seq); return ((size_t)max_seq + ch - 1) / ch; }
void kr_launch_fd_merge2(const KrFdFlashArgs& a, int hd, int nch, int chunk, hipStream_t st);   // kr_attn_flash.hip
int kr_launch_mla_flash_decode(const KrMlaArgs& a, int max_seq, hipStream_t st) {
    if (a.rd != 64 || (a.klr != 512 && a.klr != 256) || !a.step || !a.fd_o || !a.fd_ml || a.nh > MF_ROWS) return 1;
    const int chunk = kr_mla_flash_decode_chunk(max_seq), nch = (int)kr_mla_flash_decode_chunks(max_seq);
    if (nch > 1024) return 1;
    dim3 grid(1, nch);
    if (a.klr == 512) {
        if (a.kv_fp8) hipLaunchKernelGGL((kr_mla_flash_kernel<512, true, true>), grid, dim3(256), kr_mla_flash_lds<512>(), st, a, 1, nch, chunk / MF_TK);
        else hipLaunchKernelGGL((kr_mla_flash_kernel<512, false, true>), grid, dim3(256), kr_mla_flash_lds<512>(), st, a, 1, nch, chunk / MF_TK);
    } else {
        if (a.kv_fp8) hipLaunchKernelGGL((kr_mla_flash_kernel<256, true, true>), grid, dim3(256), kr_mla_flash_lds<256>(), st, a, 1, nch, chunk / MF_TK);
        else hipLaunchKernelGGL((kr_mla_flash_kernel<256, false, true>), grid, dim3(256), kr_mla_flash_lds<256>(), st, a, 1, nch, chunk / MF_TK);
    }
    KrFdFlashArgs m{};
    m.step = a.step; m.fd_o = a.fd_o; m.fd_ml = a.fd_ml; m.nh = a.nh; m.nkv = 1; m.gated = 0; m.out = a.attn_lat; m.img_out = nullptr;
    kr_launch_fd_merge2(m, a.klr, nch, chunk, st);
    return 0;
}
