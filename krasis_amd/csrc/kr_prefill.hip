// kr_prefill.hip -- prefill (M >> 1) expert path for gfx950: token sort -> grouped GEMM on int8 MFMA -> combine.
//
// Replaces GpuPrefillManager.forward / fused_marlin_moe (python/krasis/gpu_prefill.py:64-239,4374-4484: moe_align_block_size,
// moe_wna16_marlin_gemm x2, silu_and_mul, moe_sum_reduce -- all third-party CUDA) with ONE numerics story: every (token, expert)
// row is computed with exactly the arithmetic of the reference's CPU engine (expert_forward_unified, src/moe.rs:184), i.e. the
// result of kr_moe_prefill is bit-identical to kr_moe_forward / KrasisEngine.forward_moe_direct on the same batch.
//
// How INT16 activations ride the int8 matrix cores exactly:
//   a (i16) = AH*256 + AL, AH = a >> 8 in [-128,127], AL' = (a & 255) - 128 in [-128,127]
//   sum_k w*a = 256 * mfma(AH, w) + mfma(AL', w) + 128 * sum_k w       (w = nibble - 8, sum_k w precomputed per group/column)
// Two v_mfma_i32_32x32x32_i8 per B fragment, i32 accumulators reset every 128-wide quantization group, then the same
// one-fma-per-group f32 chain as the reference: out = fma(f32(isum), bf16(w_scale) * a_scale, out).
//
// Tile: 64 rows (tokens routed to one expert) x 128 columns per workgroup of 4 waves; one group PAIR (256 k) per LDS stage.
#include "kr_lds_optin.h"
#include "kr_device.h"
#include "kr_kernels.h"
#include "kr_prefill.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define PF_BM 64

// ------------------------------------------------------------------------------------------
// token sort (moe_align_block_size equivalent): rows of the grouped GEMM = (token, slot) pairs grouped by expert
// ------------------------------------------------------------------------------------------
__global__ void kr_pf_count_kernel(const int32_t* __restrict__ ids, int n_pairs, int E, int* __restrict__ counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    const int e = ids[i];
    if (e >= 0 && e < E) atomicAdd(&counts[e], 1);
}

// The same two passes with the histogram PRIVATE to the workgroup (E <= 1024; 2048 pairs per workgroup): one global atomic per (workgroup, expert
// present in its slice) instead of one per pair -- with a skewed routing thousands of pairs hit the same counter (the per-pair form took 108 us per
// pass for the 81 920 pairs of an 8192-token chunk of the synthetic model, 26 us with uniform ids).  The scatter pass re-counts its slice, reserves a
// contiguous range per expert with ONE atomic on the cursor, then places its pairs with LDS atomics.  Row order inside an expert: any (see above).
#define KR_PF_WG_PAIRS 2048
__global__ void __launch_bounds__(256) kr_pf_count_wg_kernel(const int32_t* __restrict__ ids, int n_pairs, int E, int* __restrict__ counts) {
    __shared__ int s_cnt[1024];
    const int t = threadIdx.x, base = blockIdx.x * KR_PF_WG_PAIRS;
    for (int e = t; e < E; e += 256) s_cnt[e] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KR_PF_WG_PAIRS / 256; j++) {
        const int i = base + j * 256 + t;
        if (i < n_pairs) { const int e = ids[i]; if (e >= 0 && e < E) atomicAdd(&s_cnt[e], 1); }
    }
    __syncthreads();
    for (int e = t; e < E; e += 256) { const int c = s_cnt[e]; if (c) atomicAdd(&counts[e], c); }
}
__global__ void __launch_bounds__(256) kr_pf_scatter_wg_kernel(const int32_t* __restrict__ ids, int n_pairs, int E, const int* __restrict__ offsets, int* __restrict__ cursor,
                                                              int* __restrict__ row_pair, int* __restrict__ pair_row) {
    __shared__ int s_cnt[1024], s_base[1024];
    const int t = threadIdx.x, base = blockIdx.x * KR_PF_WG_PAIRS;
    for (int e = t; e < E; e += 256) s_cnt[e] = 0;
    __syncthreads();
    int my[KR_PF_WG_PAIRS / 256];
#pragma unroll
    for (int j = 0; j < KR_PF_WG_PAIRS / 256; j++) {
        const int i = base + j * 256 + t;
        int e = -1;
        if (i < n_pairs) { e = ids[i]; if (e < 0 || e >= E) e = -1; }
        my[j] = e;
        if (e >= 0) atomicAdd(&s_cnt[e], 1);
    }
    __syncthreads();
    for (int e = t; e < E; e += 256) { const int c = s_cnt[e]; s_base[e] = c ? offsets[e] + atomicAdd(&cursor[e], c) : 0; s_cnt[e] = 0; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KR_PF_WG_PAIRS / 256; j++) {
        const int i = base + j * 256 + t;
        if (i >= n_pairs) continue;
        const int e = my[j];
        if (e < 0) { pair_row[i] = -1; continue; }
        const int r = s_base[e] + atomicAdd(&s_cnt[e], 1);
        row_pair[r] = i; pair_row[i] = r;
    }
}

// one block: exclusive scan of counts -> row offsets; tile table (expert, first row, rows) for every 64-row tile
__global__ void __launch_bounds__(1024) kr_pf_scan_kernel(const int* __restrict__ counts, int E, int* __restrict__ offsets, int* __restrict__ cursor,
                                                         int* __restrict__ tile_expert, int* __restrict__ tile_row0, int* __restrict__ tile_rows,
                                                         int* __restrict__ n_tiles_out, int bm) {
    __shared__ int s_off[1025], s_tile[1025];
    const int t = threadIdx.x;
    if (t == 0) {
        int o = 0, tl = 0;
        for (int e = 0; e < E; e++) { s_off[e] = o; s_tile[e] = tl; o += counts[e]; tl += (counts[e] + bm - 1) / bm; }
        s_off[E] = o; s_tile[E] = tl; n_tiles_out[0] = tl; n_tiles_out[1] = o;   // [1]: rows in total (valid pairs)
    }
    __syncthreads();
    for (int e = t; e < E; e += 1024) {
        offsets[e] = s_off[e]; cursor[e] = 0;
        const int c = counts[e];
        for (int i = 0; i * bm < c; i++) {
            const int ti = s_tile[e] + i;
            tile_expert[ti] = e; tile_row0[ti] = s_off[e] + i * bm; tile_rows[ti] = (c - i * bm) < bm ? (c - i * bm) : bm;
        }
    }
}

// One workgroup sorts a chunk's (token, slot) pairs by expert: count (LDS atomics) -> exclusive scans of rows and 64-row tiles -> scatter, in ONE
// launch.  The three-launch form above costs 19 + 32 + 22 us per MoE layer of a 1024-token chunk (the scan walked the experts on one thread,
// one dependent global load each) -- 5 % of the FAST prompt pass; this form is used up to KR_PF_SORT1_MAX pairs and 1024 experts.
// The row order INSIDE an expert is the order in which the LDS atomics land: any order gives the same results (rows are independent; the
// combine goes through pair_row).
#define KR_PF_SORT1_MAX 32768
__global__ void __launch_bounds__(1024) kr_pf_sort1_kernel(const int32_t* __restrict__ ids, int n_pairs, int E, int* __restrict__ counts, int* __restrict__ offsets,
                                                          int* __restrict__ tile_expert, int* __restrict__ tile_row0, int* __restrict__ tile_rows,
                                                          int* __restrict__ n_tiles_out, int* __restrict__ row_pair, int* __restrict__ pair_row, int bm) {
    __shared__ int s_cnt[1024], s_off[1024], s_til[1024], s_cur[1024];
    const int t = threadIdx.x;
    s_cnt[t] = 0; s_cur[t] = 0;
    __syncthreads();
    for (int i = t; i < n_pairs; i += 1024) { const int e = ids[i]; if (e >= 0 && e < E) atomicAdd(&s_cnt[e], 1); }
    __syncthreads();
    const int c = t < E ? s_cnt[t] : 0, nt = (c + bm - 1) / bm;
    s_off[t] = c; s_til[t] = nt;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {             // inclusive Hillis-Steele scans of rows and tiles
        const int a = t >= d ? s_off[t - d] : 0, b = t >= d ? s_til[t - d] : 0;
        __syncthreads();
        s_off[t] += a; s_til[t] += b;
        __syncthreads();
    }
    const int off = s_off[t] - c, til = s_til[t] - nt;   // exclusive
    if (t == 1023) { n_tiles_out[0] = s_til[t]; n_tiles_out[1] = s_off[t]; }
    __syncthreads();
    s_off[t] = off;
    if (t < E) {
        counts[t] = c; offsets[t] = off;
        for (int i = 0; i < nt; i++) { tile_expert[til + i] = t; tile_row0[til + i] = off + i * bm; tile_rows[til + i] = (c - i * bm) < bm ? (c - i * bm) : bm; }
    }
    __syncthreads();
    for (int i = t; i < n_pairs; i += 1024) {
        const int e = ids[i];
        if (e < 0 || e >= E) { pair_row[i] = -1; continue; }
        const int r = s_off[e] + atomicAdd(&s_cur[e], 1);
        row_pair[r] = i; pair_row[i] = r;
    }
}

__global__ void kr_pf_scatter_kernel(const int32_t* __restrict__ ids, int n_pairs, int E, const int* __restrict__ offsets, int* __restrict__ cursor,
                                     int* __restrict__ row_pair, int* __restrict__ pair_row) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    const int e = ids[i];
    if (e < 0 || e >= E) { pair_row[i] = -1; return; }
    const int r = offsets[e] + atomicAdd(&cursor[e], 1);
    row_pair[r] = i; pair_row[i] = r;
}

// ------------------------------------------------------------------------------------------
// activation digits: quantize_activation_int16 (avx2.rs:234) per token, stored as two int8 planes + per-group scale
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void kr_store_digits(int8_t* hi, int8_t* lo, const int (&q)[8]) {
    u32x2 h, l;
    h.x = kr_pack4(q[0] >> 8, q[1] >> 8, q[2] >> 8, q[3] >> 8); h.y = kr_pack4(q[4] >> 8, q[5] >> 8, q[6] >> 8, q[7] >> 8);
    l.x = kr_pack4((q[0] & 255) - 128, (q[1] & 255) - 128, (q[2] & 255) - 128, (q[3] & 255) - 128);
    l.y = kr_pack4((q[4] & 255) - 128, (q[5] & 255) - 128, (q[6] & 255) - 128, (q[7] & 255) - 128);
    *reinterpret_cast<u32x2*>(hi) = h; *reinterpret_cast<u32x2*>(lo) = l;
}

// grid (M), block K/8 threads (<= 1024): thread = one 8-element chunk, 16 threads = one group
__global__ void kr_pf_quant_x_kernel(const uint16_t* __restrict__ x, int K, int8_t* __restrict__ xh, int8_t* __restrict__ xl, float* __restrict__ xs) {
    const int t = blockIdx.x;
    for (int c = threadIdx.x; c < K / 8; c += blockDim.x) {
        const u32x4 r = *reinterpret_cast<const u32x4*>(x + (size_t)t * K + (size_t)c * 8);
        float v[8];
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xFFFF0000u); v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xFFFF0000u);
        v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xFFFF0000u); v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xFFFF0000u);
        float mx = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) mx = fmaxf(mx, fabsf(v[i]));
        mx = kr_red16_max_f32(mx);
        const float scale = mx > 0.0f ? mx / 32767.0f : 1.0f, inv = mx > 0.0f ? 32767.0f / mx : 0.0f;
        int q[8];
        kr_quant8<false>(v, inv, q);
        kr_store_digits(xh + (size_t)t * K + c * 8, xl + (size_t)t * K + c * 8, q);
        if ((c & 15) == 0) xs[(size_t)t * (K / 128) + (c >> 4)] = scale;
    }
}

// hidden digits from gu rows: silu_quantize_int16_avx2 (avx2.rs:2310) or the GPT-OSS activation (moe.rs:268-287). grid (rows), block I/8
template <int ACT>
__global__ void kr_pf_act_kernel(const float* __restrict__ gu, int n, int gu_ld, float swiglu_limit, float alpha, int8_t* __restrict__ hh,
                                 int8_t* __restrict__ hl, float* __restrict__ hs) {
    const int row = blockIdx.x;
    const float* g = gu + (size_t)row * gu_ld;
    for (int c = threadIdx.x; c < n / 8; c += blockDim.x) {
        const float4 g0 = *reinterpret_cast<const float4*>(g + c * 8), g1 = *reinterpret_cast<const float4*>(g + c * 8 + 4);
        const float4 u0 = *reinterpret_cast<const float4*>(g + n + c * 8), u1 = *reinterpret_cast<const float4*>(g + n + c * 8 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, uu[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
        float h[8], mx = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (ACT == KR_ACT_GPTOSS) {
                float gate = gg[i], up = uu[i];
                if (gate > swiglu_limit) gate = swiglu_limit;
                if (up > swiglu_limit) up = swiglu_limit;
                if (up < -swiglu_limit) up = -swiglu_limit;
                h[i] = (up + 1.0f) * (gate * kr_sigmoid_poly5_scalar(gate * alpha));
            } else h[i] = (gg[i] * kr_sigmoid_poly5(gg[i])) * uu[i];
            mx = fmaxf(mx, fabsf(h[i]));
        }
        mx = kr_red16_max_f32(mx);
        const float scale = mx > 0.0f ? mx / 32767.0f : 1.0f, inv = mx > 0.0f ? 32767.0f / mx : 0.0f;
        int q[8];
        if (ACT == KR_ACT_SILU_FUSED) kr_quant8<true>(h, inv, q); else kr_quant8<false>(h, inv, q);
        kr_store_digits(hh + (size_t)row * n + c * 8, hl + (size_t)row * n + c * 8, q);
        if ((c & 15) == 0) hs[(size_t)row * (n / 128) + (c >> 4)] = scale;
    }
}

// per (expert, column, group): sum_k (nibble - 8), packed as i16 pairs in the layout of the scale pairs.  grid (N/8 tiles, experts), 64 thr
__global__ void __launch_bounds__(64) kr_pf_wsum_kernel(const KrMatDev m, uint32_t* __restrict__ wsum) {
    const int tile = blockIdx.x, e = blockIdx.y, lane = threadIdx.x, col = lane >> 3;
    const u32x4* q = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(m.q) + (size_t)e * m.q_stride) + (size_t)tile * m.ngp * 64 + lane;
    uint32_t* out = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(wsum) + (size_t)e * m.s_stride) + (size_t)tile * m.ngp * 8 + col;
    for (int gp = 0; gp < m.ngp; gp++) {
        const u32x4 w = q[(size_t)gp * 64];
        int s0 = __builtin_amdgcn_sdot4((int)(w.x & 0x0F0F0F0Fu), 0x01010101, 0, false) + __builtin_amdgcn_sdot4((int)((w.x >> 4) & 0x0F0F0F0Fu), 0x01010101, 0, false) +
                 __builtin_amdgcn_sdot4((int)(w.y & 0x0F0F0F0Fu), 0x01010101, 0, false) + __builtin_amdgcn_sdot4((int)((w.y >> 4) & 0x0F0F0F0Fu), 0x01010101, 0, false) - 128;
        int s1 = __builtin_amdgcn_sdot4((int)(w.z & 0x0F0F0F0Fu), 0x01010101, 0, false) + __builtin_amdgcn_sdot4((int)((w.z >> 4) & 0x0F0F0F0Fu), 0x01010101, 0, false) +
                 __builtin_amdgcn_sdot4((int)(w.w & 0x0F0F0F0Fu), 0x01010101, 0, false) + __builtin_amdgcn_sdot4((int)((w.w >> 4) & 0x0F0F0F0Fu), 0x01010101, 0, false) - 128;
        s0 = kr_red8_add_i32(s0); s1 = kr_red8_add_i32(s1);
        if ((lane & 7) == 0) out[gp * 8] = ((uint32_t)s0 & 0xFFFFu) | ((uint32_t)s1 << 16);
    }
}

// INT8-g128 twin: per (expert, column, group) sum_k w, same i16-pair layout.  grid (N/8 tiles, experts), 64 thr
__global__ void __launch_bounds__(64) kr_pf_wsum8_kernel(const KrMatDev m, uint32_t* __restrict__ wsum) {
    const int tile = blockIdx.x, e = blockIdx.y, lane = threadIdx.x, col = lane >> 3;
    const u32x4* q = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(m.q) + (size_t)e * m.q_stride) + (size_t)tile * m.ng * 64 + lane;
    uint32_t* out = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(wsum) + (size_t)e * m.s_stride) + (size_t)tile * m.ngp * 8 + col;
    for (int gp = 0; gp < m.ngp; gp++) {
        int sv[2] = {0, 0};
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int g = 2 * gp + h;
            if (g < m.ng) {
                const u32x4 w = q[(size_t)g * 64];
                int t = __builtin_amdgcn_sdot4((int)w.x, 0x01010101, 0, false); t = __builtin_amdgcn_sdot4((int)w.y, 0x01010101, t, false);
                t = __builtin_amdgcn_sdot4((int)w.z, 0x01010101, t, false); t = __builtin_amdgcn_sdot4((int)w.w, 0x01010101, t, false);
                sv[h] = t;
            }
            sv[h] = kr_red8_add_i32(sv[h]);
        }
        if ((lane & 7) == 0) out[gp * 8] = ((uint32_t)sv[0] & 0xFFFFu) | ((uint32_t)sv[1] << 16);
    }
}

// ------------------------------------------------------------------------------------------
// grouped GEMM on int8 MFMA
// ------------------------------------------------------------------------------------------
struct KrPfGemmArgs {
    KrMatDev m; const uint32_t* wsum;      // weights of the layer's experts (+ wsum in the scale-pair layout)
    const int8_t* a_hi; const int8_t* a_lo; const float* a_scale;   // digits [rows_or_tokens][K], scales [..][K/128]
    const int* row_pair; int topk; int gather_tokens;               // stage 1: A row = token of the pair; stage 2: A row = GEMM row
    const int* tile_expert; const int* tile_row0; const int* tile_rows; const int* n_tiles;
    float* out; int out_ld;                 // [rows][out_ld]
    int single_expert;                      // shared expert: every tile uses expert 0 of `m`, rows are tokens 0..M-1 in order
    int total_rows;
    int scatter_rows;                       // expert-parallel rows: GEMM row r is written to out row row_pair[r] (its place in the caller's order) -- no combine pass
    int out_bf16;                           // ... as bf16 (RNE), the dtype the rows travel back in
    int var_rows;                           // sorted expert tiles with fewer than 64 rows per expert on average: take the kernel that skips empty 32-row blocks
    // up to two MORE matrices that consume the same A rows (same K, same width): their column blocks follow those of `m` in the same launch
    // (q | k | v, qkvz | ba, shared gate_up | shared gate: a 64- or 1-column GEMM of its own is one latency-bound launch per chunk and layer)
    int n_extra; KrMatDev mx[2]; const uint32_t* wsumx[2]; float* outx[2]; int out_ldx[2];
    int sr, sc;                             // dense launches: super-tile shape per XCD (kr_pf_super_tile), set by the launcher
};

#include "kr_prefill_gemm2.inc"

// ------------------------------------------------------------------------------------------
// combine: out[t] = sum_s w[t][s] * eo[row(t,s)] in routing order (moe.rs:661-667); shared: rsf*out + shared (moe.rs:703-706)
// ------------------------------------------------------------------------------------------
// 4 columns per thread (16-byte loads of the f32 rows, 8-byte loads of bf16 rows): the expert rows are read once, 671 MB per layer of an 8192-token QCN
// chunk -- the pass is HBM / L2 bound and one column per thread left it at half the achievable rate.  Per column the sum is the same sequence of
// (mul, add) in routing order as before.  H % 4 == 0 (the GEMM path needs H % 128 == 0).
template <int ROWS>      // element type of the expert rows: 0 f32, 1 bf16, 2 f16 (tolerance GEMM)
__global__ void __launch_bounds__(256) kr_pf_combine_kernel(const void* __restrict__ eo_v, const int* __restrict__ pair_row, const float* __restrict__ wts,
                                                           int topk, int H, const float* __restrict__ shared_eo, float rsf, void* out, int out_bf16,
                                                           const float* __restrict__ row_mul) {
    // ROWS == 2: the rows hold the RAW accumulators of the down GEMM (sums of 2^-e-scaled hidden values times 16 w: O(1..100) whatever the row's
    // magnitude, so f16 neither overflows nor goes denormal); row_mul[r] = 2^e / 16 (a power of two: exact) restores the row here.
    const int t = blockIdx.y, j = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (j >= H) return;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int s = 0; s < topk; s++) {
        const int r = pair_row[(size_t)t * topk + s];
        if (r < 0) continue;
        const float w = wts[(size_t)t * topk + s];
        float v[4];
        if (ROWS == 1) {
            const u32x2 p = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(eo_v) + (size_t)r * H + j);
            v[0] = __uint_as_float(p.x << 16); v[1] = __uint_as_float(p.x & 0xFFFF0000u); v[2] = __uint_as_float(p.y << 16); v[3] = __uint_as_float(p.y & 0xFFFF0000u);
        } else if (ROWS == 2) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const h4 p = *reinterpret_cast<const h4*>(reinterpret_cast<const uint16_t*>(eo_v) + (size_t)r * H + j);
            const float rm = row_mul[r];
            v[0] = (float)p.x * rm; v[1] = (float)p.y * rm; v[2] = (float)p.z * rm; v[3] = (float)p.w * rm;
        } else {
            const float4 p = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(eo_v) + (size_t)r * H + j);
            v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = p.w;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] += w * v[i];
    }
    if (shared_eo) {
        const float4 sh = *reinterpret_cast<const float4*>(shared_eo + (size_t)t * H + j);
        acc[0] = rsf * acc[0] + sh.x; acc[1] = rsf * acc[1] + sh.y; acc[2] = rsf * acc[2] + sh.z; acc[3] = rsf * acc[3] + sh.w;
    }
    if (out_bf16) {
        u32x2 o;
        o.x = (uint32_t)kr_f32_to_bf16(acc[0]) | ((uint32_t)kr_f32_to_bf16(acc[1]) << 16); o.y = (uint32_t)kr_f32_to_bf16(acc[2]) | ((uint32_t)kr_f32_to_bf16(acc[3]) << 16);
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(out) + (size_t)t * H + j) = o;
    } else *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)t * H + j) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void kr_launch_pf_sort(const int32_t* ids, int M, int topk, int E, KrPfSort s, hipStream_t st, int bm) {
    const int n = M * topk;
    if (n <= KR_PF_SORT1_MAX && E <= 1024) {
        hipLaunchKernelGGL(kr_pf_sort1_kernel, dim3(1), dim3(1024), 0, st, ids, n, E, s.counts, s.offsets, s.tile_expert, s.tile_row0, s.tile_rows, s.n_tiles, s.row_pair, s.pair_row, bm);
        return;
    }
    (void)hipMemsetAsync(s.counts, 0, (size_t)E * 4, st);
    if (E <= 1024) {
        const int g = (n + KR_PF_WG_PAIRS - 1) / KR_PF_WG_PAIRS;
        hipLaunchKernelGGL(kr_pf_count_wg_kernel, dim3(g), dim3(256), 0, st, ids, n, E, s.counts);
        hipLaunchKernelGGL(kr_pf_scan_kernel, dim3(1), dim3(1024), 0, st, s.counts, E, s.offsets, s.cursor, s.tile_expert, s.tile_row0, s.tile_rows, s.n_tiles, bm);
        hipLaunchKernelGGL(kr_pf_scatter_wg_kernel, dim3(g), dim3(256), 0, st, ids, n, E, s.offsets, s.cursor, s.row_pair, s.pair_row);
        return;
    }
    hipLaunchKernelGGL(kr_pf_count_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ids, n, E, s.counts);
    hipLaunchKernelGGL(kr_pf_scan_kernel, dim3(1), dim3(1024), 0, st, s.counts, E, s.offsets, s.cursor, s.tile_expert, s.tile_row0, s.tile_rows, s.n_tiles, bm);
    hipLaunchKernelGGL(kr_pf_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ids, n, E, s.offsets, s.cursor, s.row_pair, s.pair_row);
}
void kr_launch_pf_quant_x(const uint16_t* x, int M, int K, int8_t* xh, int8_t* xl, float* xs, hipStream_t st) {
    const int thr = K / 8 < 1024 ? K / 8 : 1024;
    hipLaunchKernelGGL(kr_pf_quant_x_kernel, dim3(M), dim3(thr), 0, st, x, K, xh, xl, xs);
}
void kr_launch_pf_act(const float* gu, int rows, int n, int gu_ld, int act_mode, float swiglu_limit, float alpha, int8_t* hh, int8_t* hl, float* hs, hipStream_t st) {
    const int thr = n / 8 < 1024 ? n / 8 : 1024;
    if (act_mode == KR_ACT_GPTOSS) hipLaunchKernelGGL(kr_pf_act_kernel<KR_ACT_GPTOSS>, dim3(rows), dim3(thr), 0, st, gu, n, gu_ld, swiglu_limit, alpha, hh, hl, hs);
    else if (act_mode == KR_ACT_SILU_MUL) hipLaunchKernelGGL(kr_pf_act_kernel<KR_ACT_SILU_MUL>, dim3(rows), dim3(thr), 0, st, gu, n, gu_ld, swiglu_limit, alpha, hh, hl, hs);
    else hipLaunchKernelGGL(kr_pf_act_kernel<KR_ACT_SILU_FUSED>, dim3(rows), dim3(thr), 0, st, gu, n, gu_ld, swiglu_limit, alpha, hh, hl, hs);
}
void kr_launch_pf_wsum(const KrMatDev& m, int n_experts, uint32_t* wsum, hipStream_t st) {
    if (m.bits == 8) { hipLaunchKernelGGL(kr_pf_wsum8_kernel, dim3((m.N + 7) / 8, n_experts), dim3(64), 0, st, m, wsum); return; }
    hipLaunchKernelGGL(kr_pf_wsum_kernel, dim3((m.N + 7) / 8, n_experts), dim3(64), 0, st, m, wsum);
}
void kr_launch_pf_gemm(const KrMatDev& m, const uint32_t* wsum, const int8_t* a_hi, const int8_t* a_lo, const float* a_scale, const KrPfSort* sort, int topk,
                       int gather_tokens, int max_tiles, int single_expert_rows, float* out, int out_ld, hipStream_t st, int scatter_rows, int out_bf16, int var_rows) {
    KrPfGemmArgs a{};
    a.scatter_rows = scatter_rows; a.out_bf16 = out_bf16;
    a.var_rows = sort != nullptr && single_expert_rows <= 0 && var_rows;
    a.m = m; a.wsum = wsum; a.a_hi = a_hi; a.a_lo = a_lo; a.a_scale = a_scale; a.topk = topk; a.gather_tokens = gather_tokens;
    if (sort) { a.row_pair = sort->row_pair; a.tile_expert = sort->tile_expert; a.tile_row0 = sort->tile_row0; a.tile_rows = sort->tile_rows; a.n_tiles = sort->n_tiles; }
    a.out = out; a.out_ld = out_ld; a.single_expert = single_expert_rows > 0; a.total_rows = single_expert_rows;
    const int mt = single_expert_rows > 0 ? (single_expert_rows + PF_BM - 1) / PF_BM : max_tiles;
    // 32 columns per wave, two quantization groups per k stage: the best of the (64|32 columns) x (1|2 groups) shapes and of a 2 x 2 row-split
    // wave grid that were measured
    if (m.bits == 8) kr_pf_gemm2_launch<32, 2, 8>(a, mt, st);
    else kr_pf_gemm2_launch<32, 2, 4>(a, mt, st);
}
// dense GEMMs of up to three matrices over the same A rows (rows 0..M-1 in order), one launch; all matrices share K and the weight width
void kr_launch_pf_gemm_multi(const KrMatDev* mats, const uint32_t* const* wsums, float* const* outs, const int* out_lds, int n, const int8_t* a_hi, const int8_t* a_lo,
                             const float* a_scale, int M, hipStream_t st) {
    KrPfGemmArgs a{};
    a.m = mats[0]; a.wsum = wsums[0]; a.out = outs[0]; a.out_ld = out_lds[0]; a.a_hi = a_hi; a.a_lo = a_lo; a.a_scale = a_scale; a.topk = 1;
    a.single_expert = 1; a.total_rows = M; a.n_extra = n - 1;
    for (int i = 1; i < n; i++) { a.mx[i - 1] = mats[i]; a.wsumx[i - 1] = wsums[i]; a.outx[i - 1] = outs[i]; a.out_ldx[i - 1] = out_lds[i]; }
    const int mt = (M + PF_BM - 1) / PF_BM;
    if (mats[0].bits == 8) kr_pf_gemm2_launch<32, 2, 8>(a, mt, st);
    else kr_pf_gemm2_launch<32, 2, 4>(a, mt, st);
}
void kr_launch_pf_combine(const float* eo, const int* pair_row, const float* wts, int M, int topk, int H, const float* shared_eo, float rsf, void* out,
                          int out_bf16, hipStream_t st) {
    hipLaunchKernelGGL(kr_pf_combine_kernel<0>, dim3((H + 1023) / 1024, M), dim3(256), 0, st, (const void*)eo, pair_row, wts, topk, H, shared_eo, rsf, out, out_bf16, (const float*)nullptr);
}
void kr_launch_pf_combine_f16rows(const uint16_t* eo, const float* row_mul, const int* pair_row, const float* wts, int M, int topk, int H, const float* shared_eo, float rsf,
                                  void* out, int out_bf16, hipStream_t st) {
    hipLaunchKernelGGL(kr_pf_combine_kernel<2>, dim3((H + 1023) / 1024, M), dim3(256), 0, st, (const void*)eo, pair_row, wts, topk, H, shared_eo, rsf, out, out_bf16, row_mul);
}
void kr_launch_pf_combine_bf16rows(const uint16_t* eo, const int* pair_row, const float* wts, int M, int topk, int H, const float* shared_eo, float rsf, void* out,
                                   int out_bf16, hipStream_t st) {
    hipLaunchKernelGGL(kr_pf_combine_kernel<1>, dim3((H + 1023) / 1024, M), dim3(256), 0, st, (const void*)eo, pair_row, wts, topk, H, shared_eo, rsf, out, out_bf16, (const float*)nullptr);
}

// ------------------------------------------------------------------------------------------
// expert parallelism (kr_ep.cpp): destination rank + local expert id of every (token, slot) pair; rows gathered in destination order
// ------------------------------------------------------------------------------------------
__global__ void kr_ep_dest_kernel(const int32_t* __restrict__ ids, int n, int E_total, int per, int world, int full, int32_t* __restrict__ dest, int32_t* __restrict__ lid) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = ids[i];
    if (e < 0 || e >= E_total) { dest[i] = -1; lid[i] = -1; return; }
    int d = e / per; if (d > world - 1) d = world - 1;          // the last rank takes the remainder (gpu_prefill.py:353-359)
    dest[i] = d; lid[i] = full ? e : e - d * per;        // full: the owner holds every expert of the model under its global id
}
// Owner sort for a handful of destinations (world <= 64): the token sort's count / scatter kernels issue one global atomic per pair on
// counts[dest] / cursor[dest] -- with 8 destinations that is ~10 k atomics per address and layer (the same-address rate is ~90 per us).  Here a
// workgroup counts its 256 pairs per destination in LDS and touches each global counter ONCE.
__global__ void __launch_bounds__(256) kr_ep_count_kernel(const int32_t* __restrict__ dest, int n, int W, int* __restrict__ counts) {
    __shared__ int c[64];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < 64) c[threadIdx.x] = 0;
    __syncthreads();
    const int d = i < n ? dest[i] : -1;
    if (d >= 0) atomicAdd(&c[d], 1);
    __syncthreads();
    if ((int)threadIdx.x < W && c[threadIdx.x]) atomicAdd(&counts[threadIdx.x], c[threadIdx.x]);
}
__global__ void __launch_bounds__(256) kr_ep_scatter_kernel(const int32_t* __restrict__ dest, int n, int W, const int* __restrict__ offsets, int* __restrict__ cursor,
                                                           int* __restrict__ row_pair, int* __restrict__ pair_row) {
    __shared__ int c[64], base[64];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < 64) c[threadIdx.x] = 0;
    __syncthreads();
    const int d = i < n ? dest[i] : -1;
    int r = 0;
    if (d >= 0) r = atomicAdd(&c[d], 1);
    __syncthreads();
    if ((int)threadIdx.x < W) base[threadIdx.x] = c[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], c[threadIdx.x]) : 0;
    __syncthreads();
    if (i >= n) return;
    if (d < 0) { pair_row[i] = -1; return; }
    const int row = offsets[d] + base[d] + r;
    row_pair[row] = i; pair_row[i] = row;
}
void kr_launch_ep_sort(const int32_t* dest, int n, int W, KrPfSort s, hipStream_t st) {
    (void)hipMemsetAsync(s.counts, 0, (size_t)W * 4, st);
    hipLaunchKernelGGL(kr_ep_count_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dest, n, W, s.counts);
    hipLaunchKernelGGL(kr_pf_scan_kernel, dim3(1), dim3(1024), 0, st, s.counts, W, s.offsets, s.cursor, s.tile_expert, s.tile_row0, s.tile_rows, s.n_tiles, PF_BM);
    hipLaunchKernelGGL(kr_ep_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dest, n, W, s.offsets, s.cursor, s.row_pair, s.pair_row);
}

// send row r = x[token of pair row_pair[r]] (bf16 [H]) and its local expert id; grid (n_rows), H/8 threads
__global__ void kr_ep_gather_kernel(const uint16_t* __restrict__ x, const int* __restrict__ row_pair, const int32_t* __restrict__ lid, int topk, int H,
                                    const int* __restrict__ n_rows, uint16_t* __restrict__ rows, int32_t* __restrict__ row_lid) {
    const int r = blockIdx.x;
    if (r >= n_rows[0]) { if (threadIdx.x == 0) row_lid[r] = -1; return; }     // past the last routed pair: a row no expert takes
    const int pair = row_pair[r];
    const u32x4* src = reinterpret_cast<const u32x4*>(x + (size_t)(pair / topk) * H);
    u32x4* dst = reinterpret_cast<u32x4*>(rows + (size_t)r * H);
    for (int c = threadIdx.x; c < H / 8; c += blockDim.x) dst[c] = src[c];
    if (threadIdx.x == 0) row_lid[r] = lid[pair];
}
// f32 rows -> bf16 rows (RNE) for the return leg
__global__ void kr_ep_rows_bf16_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = kr_f32_to_bf16(in[i]);
}
void kr_launch_ep_dest(const int32_t* ids, int n, int E_total, int per, int world, int full, int32_t* dest, int32_t* lid, hipStream_t st) {
    hipLaunchKernelGGL(kr_ep_dest_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ids, n, E_total, per, world, full, dest, lid);
}
void kr_launch_ep_gather(const uint16_t* x, const int* row_pair, const int32_t* lid, int topk, int H, const int* n_rows, int max_rows, uint16_t* rows, int32_t* row_lid,
                         hipStream_t st) {
    if (max_rows <= 0) return;
    hipLaunchKernelGGL(kr_ep_gather_kernel, dim3(max_rows), dim3(H / 8 < 256 ? H / 8 : 256), 0, st, x, row_pair, lid, topk, H, n_rows, rows, row_lid);
}
// out[i] = parts[0][i] + parts[1][i] + ... in rank order (the loopback transport's all-reduce; RCCL does its own)
__global__ void kr_ep_sum_f32_kernel(const float* __restrict__ parts, int W, size_t n, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float a = parts[i];
        for (int r = 1; r < W; r++) a += parts[(size_t)r * n + i];
        out[i] = a;
    }
}
void kr_launch_ep_sum_f32(const float* parts, int W, size_t n, float* out, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(kr_ep_sum_f32_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, st, parts, W, n, out);
}
void kr_launch_ep_rows_bf16(const float* in, uint16_t* out, size_t n, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(kr_ep_rows_bf16_kernel, dim3(2048), dim3(256), 0, st, in, out, n);
}
