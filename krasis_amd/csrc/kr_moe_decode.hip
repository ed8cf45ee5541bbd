// kr_moe_decode.hip -- memory-bound M=1..few MoE expert path for gfx950 (MI355X).
//
// Replaces (bit-exactly) the reference's CPU decode experts:
//   moe_forward_unified (src/moe.rs:572-715) -> expert_forward_unified (src/moe.rs:184-380)
//   -> expert_matmul_int4/int8_transposed_integer (src/kernel/avx2.rs:1066,1477),
//   quantize_activation_int16 (avx2.rs:234), silu_quantize_int16_avx2 (avx2.rs:2310).
//
// Shape of the computation on the GPU:
//   * a wave owns 8 output columns; each column is served by 8 lanes that split every 128-wide
//     quantization group 16/16/.. between them.  One `global_load_dwordx4` per lane per group pair
//     => every wave-level load is 1 KiB of contiguous, streamed-once (non-temporal) HBM.
//   * integer dot products are exact: a(i16) = AH*256 + AL, nibbles stay unsigned,
//     sum(q*a) = 256*sdot4(q,AH) + udot4(q,AL), minus 8*sum(a) for the INT4 offset.
//   * the per-group i32 sums are reduced over the 8 lanes with DPP adds and folded into the
//     column's f32 accumulator with ONE fma per group, in group order -- the same chain as
//     _mm256_fmadd_ps(group_f32, w_scale*a_scale, out) in the reference, hence bit-identical.
#include "kr_device.h"
#include "kr_kernels.h"

#define KR_BLOCK 256
#define KR_WAVES (KR_BLOCK / 64)

// ------------------------------------------------------------------------------------------
// prologues: build the INT16 activation image in LDS (whole workgroup cooperates)
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ KrActLds kr_carve_lds(u32x4* smem, int K, bool want_i8) {
    KrActLds L;
    L.planes = smem;
    L.asum16 = reinterpret_cast<int*>(smem + K / 8);
    L.ascale = reinterpret_cast<float*>(L.asum16 + K / 16);
    // keep the INT8 image 16-byte aligned: asum16 (K/16 ints) + ascale (K/128 floats) rounded up
    const int tail_words = K / 16 + ((K / 128 + 3) & ~3);
    L.planes8 = want_i8 ? (smem + K / 8 + (tail_words + 3) / 4) : nullptr;
    return L;
}
static inline size_t kr_lds_bytes(int K, bool want_i8) {
    const int tail_words = K / 16 + ((K / 128 + 3) & ~3);
    size_t b = (size_t)(K / 8) * 16 + (size_t)((tail_words + 3) / 4) * 16;
    if (want_i8) b += (size_t)(K / 16) * 32;
    return b;
}

__device__ __forceinline__ void kr_load8(const uint16_t* x, int c, float (&v)[8]) {
    const u32x4 r = *reinterpret_cast<const u32x4*>(x + (size_t)c * 8);
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xFFFF0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xFFFF0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xFFFF0000u);
    v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xFFFF0000u);
}
__device__ __forceinline__ void kr_load8(const float* x, int c, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(x + (size_t)c * 8);
    const float4 b = *reinterpret_cast<const float4*>(x + (size_t)c * 8 + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// group max (16 chunks = 16 consecutive lanes) -> scale / inverse scale, avx2.rs:257-258
__device__ __forceinline__ void kr_group_scale(float mx_local, float& scale, float& inv) {
    const float mx = kr_red16_max_f32(mx_local);
    scale = mx > 0.0f ? mx / 32767.0f : 1.0f;
    inv = mx > 0.0f ? 32767.0f / mx : 0.0f;
}

// quantize_activation_int16 / _f32 (avx2.rs:234,274): per-128 scale, round half away from zero
template <typename T, bool I8>
__device__ __forceinline__ void kr_prologue_quant(const T* x, int K, const KrActLds& L) {
    const int nchunks = K / 8;
    for (int c = threadIdx.x; c < nchunks; c += KR_BLOCK) {
        float v[8];
        kr_load8(x, c, v);
        float mx = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) mx = fmaxf(mx, fabsf(v[i]));
        float scale, inv;
        kr_group_scale(mx, scale, inv);
        int q[8];
        kr_quant8<false>(v, inv, q);
        kr_store_chunk<I8>(L, c, q);
        if ((c & 15) == 0) L.ascale[c >> 4] = scale;
    }
}

// decode graph: f32 hidden, optionally rounded to bf16 first (decode.rs:3307-3309 feeds bf16(hidden) to the routed experts)
template <bool I8>
__device__ __forceinline__ void kr_prologue_quant_f32(const float* x, int K, const KrActLds& L, bool round_bf16) {
    const int nchunks = K / 8;
    for (int c = threadIdx.x; c < nchunks; c += KR_BLOCK) {
        float v[8];
        kr_load8(x, c, v);
        float mx = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (round_bf16) v[i] = kr_bf16_to_f32(kr_f32_to_bf16(v[i]));
            mx = fmaxf(mx, fabsf(v[i]));
        }
        float scale, inv;
        kr_group_scale(mx, scale, inv);
        int q[8];
        kr_quant8<false>(v, inv, q);
        kr_store_chunk<I8>(L, c, q);
        if ((c & 15) == 0) L.ascale[c >> 4] = scale;
    }
}

// hidden = act(gate, up) then per-128 INT16 quantization, from gu = [gate(n) | up(n)] in global memory
template <int ACT, bool I8>
__device__ __forceinline__ void kr_prologue_hidden(const float* gu, int n, float swiglu_limit, float alpha, const KrActLds& L) {
    const int nchunks = n / 8;
    for (int c = threadIdx.x; c < nchunks; c += KR_BLOCK) {
        float g[8], u[8], h[8];
        kr_load8(gu, c, g);
        kr_load8(gu + n, c, u);
        float mx = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (ACT == KR_ACT_GPTOSS) {  // moe.rs:272-280
                float gate = g[i], up = u[i];
                if (gate > swiglu_limit) gate = swiglu_limit;
                if (up > swiglu_limit) up = swiglu_limit;
                if (up < -swiglu_limit) up = -swiglu_limit;
                const float glu = gate * kr_sigmoid_poly5_scalar(gate * alpha);
                h[i] = (up + 1.0f) * glu;
            } else {                      // avx2.rs:2331-2333 / decode.rs:1731-1733
                const float silu = g[i] * kr_sigmoid_poly5(g[i]);
                h[i] = silu * u[i];
            }
            mx = fmaxf(mx, fabsf(h[i]));
        }
        float scale, inv;
        kr_group_scale(mx, scale, inv);
        int q[8];
        if (ACT == KR_ACT_SILU_FUSED) kr_quant8<true>(h, inv, q);   // _mm256_cvtps_epi32
        else kr_quant8<false>(h, inv, q);                            // f32::round
        kr_store_chunk<I8>(L, c, q);
        if ((c & 15) == 0) L.ascale[c >> 4] = scale;
    }
}

// ------------------------------------------------------------------------------------------
// the streaming matvec tile: 8 columns per wave, 8 lanes per column
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ void kr_dot_word_i4(uint32_t w, const u32x4 r, int& accH, uint32_t& accL) {
    const uint32_t lo = w & 0x0F0F0F0Fu;         // k = 0,2,4,6
    const uint32_t hi = (w >> 4) & 0x0F0F0F0Fu;  // k = 1,3,5,7
    accH = __builtin_amdgcn_sdot4((int)lo, (int)r.x, accH, false);
    accH = __builtin_amdgcn_sdot4((int)hi, (int)r.y, accH, false);
    accL = __builtin_amdgcn_udot4(lo, r.z, accL, false);
    accL = __builtin_amdgcn_udot4(hi, r.w, accL, false);
}

// one quantization group, INT4: lane's two packed words -> exact i32 partial of sum((q-8)*a)
__device__ __forceinline__ int kr_group_i4(uint32_t w0, uint32_t w1, int g, int l8, const KrActLds& L) {
    const int chunk = g * 16 + 2 * l8;
    const u32x4 r0 = L.planes[chunk], r1 = L.planes[chunk + 1];
    int accH = 0; uint32_t accL = 0;
    kr_dot_word_i4(w0, r0, accH, accL);
    kr_dot_word_i4(w1, r1, accH, accL);
    return (accH << 8) + (int)accL - 8 * L.asum16[g * 8 + l8];
}

// one quantization group, INT8: lane's 16 weights (natural k order)
__device__ __forceinline__ int kr_group_i8(const u32x4 w, int g, int l8, const KrActLds& L) {
    const u32x4 ah = L.planes8[(g * 8 + l8) * 2], al = L.planes8[(g * 8 + l8) * 2 + 1];
    int accH = 0, accL = 0, accW = 0;
    accH = __builtin_amdgcn_sdot4((int)w.x, (int)ah.x, accH, false);
    accH = __builtin_amdgcn_sdot4((int)w.y, (int)ah.y, accH, false);
    accH = __builtin_amdgcn_sdot4((int)w.z, (int)ah.z, accH, false);
    accH = __builtin_amdgcn_sdot4((int)w.w, (int)ah.w, accH, false);
    accL = __builtin_amdgcn_sdot4((int)w.x, (int)al.x, accL, false);
    accL = __builtin_amdgcn_sdot4((int)w.y, (int)al.y, accL, false);
    accL = __builtin_amdgcn_sdot4((int)w.z, (int)al.z, accL, false);
    accL = __builtin_amdgcn_sdot4((int)w.w, (int)al.w, accL, false);
    accW = __builtin_amdgcn_sdot4((int)w.x, 0x01010101, accW, false);
    accW = __builtin_amdgcn_sdot4((int)w.y, 0x01010101, accW, false);
    accW = __builtin_amdgcn_sdot4((int)w.z, 0x01010101, accW, false);
    accW = __builtin_amdgcn_sdot4((int)w.w, 0x01010101, accW, false);
    return (accH << 8) + accL + (accW << 7);  // AL = AL' + 128
}

__device__ __forceinline__ float kr_chain(float acc, int isum, uint32_t sbits, float a_scale, bool fused) {
    const float comb = __uint_as_float(sbits << 16) * a_scale;       // bf16(w_scale) * a_scale, avx2.rs:1171
    const float gf = (float)isum;
    return fused ? __builtin_fmaf(gf, comb, acc) : (acc + gf * comb); // avx2.rs:1175 / :1201
}

#define KR_PRE 8   // group pairs (INT4) / groups (INT8) whose weights are fetched BEFORE the activation prologue runs

// The first KR_PRE weight records of a tile are requested before the workgroup builds its activation image, so the HBM
// latency of the stream overlaps the prologue (K = 2048 is covered entirely: 8 x 16 B per lane in flight).
struct KrPre { u32x4 w[KR_PRE]; uint32_t sc[KR_PRE]; };

template <int BITS>
__device__ __forceinline__ void kr_preload(KrPre& p, const void* qbase, const uint32_t* sbase, const KrMatDev& m, int tile, int lane, int unit0) {
    const int units = BITS == 4 ? m.ngp : m.ng;
    const u32x4* q = reinterpret_cast<const u32x4*>(qbase) + (size_t)tile * units * 64 + lane;
    const uint32_t* s = sbase + (size_t)tile * m.ngp * 8 + (lane >> 3);
#pragma unroll
    for (int u = 0; u < KR_PRE; u++)
        if (unit0 + u < units) { p.w[u] = kr_ldg_nt(q + (size_t)(unit0 + u) * 64); p.sc[u] = kr_ldg_nt(s + (BITS == 4 ? (unit0 + u) : ((unit0 + u) >> 1)) * 8); }
}

template <int BITS>
__device__ __forceinline__ float kr_matvec_tile(KrPre& p, bool preloaded, const void* qbase, const uint32_t* sbase, const KrMatDev& m, int tile,
                                               const KrActLds& L, int lane) {
    const int l8 = lane & 7, col = lane >> 3;
    const bool fused = (tile * 8 + col) < m.n_fma;
    const int units = BITS == 4 ? m.ngp : m.ng;
    float acc = 0.0f;
    for (int u0 = 0; u0 < units; u0 += KR_PRE) {
        if (u0 > 0 || !preloaded) kr_preload<BITS>(p, qbase, sbase, m, tile, lane, u0);
#pragma unroll
        for (int u = 0; u < KR_PRE; u++) {
            const int un = u0 + u;
            if (un < units) {
                if (BITS == 4) {
                    const int g0 = 2 * un, g1 = g0 + 1;
                    const int i0 = kr_red8_add_i32(kr_group_i4(p.w[u].x, p.w[u].y, g0, l8, L));
                    acc = kr_chain(acc, i0, p.sc[u] & 0xFFFFu, L.ascale[g0], fused);
                    if (g1 < m.ng) {
                        const int i1 = kr_red8_add_i32(kr_group_i4(p.w[u].z, p.w[u].w, g1, l8, L));
                        acc = kr_chain(acc, i1, p.sc[u] >> 16, L.ascale[g1], fused);
                    }
                } else {
                    const int i0 = kr_red8_add_i32(kr_group_i8(p.w[u], un, l8, L));
                    acc = kr_chain(acc, i0, (un & 1) ? (p.sc[u] >> 16) : (p.sc[u] & 0xFFFFu), L.ascale[un], fused);
                }
            }
        }
    }
    return acc;
}

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------

extern __shared__ __attribute__((aligned(16))) u32x4 kr_smem[];

struct KrSlot { const void* q13; const uint32_t* s13; const void* q2; const uint32_t* s2; int inter; bool shared; bool valid; };

__device__ __forceinline__ KrSlot kr_resolve_slot(const KrMoeArgs& a, int b, int slot) {
    KrSlot r;
    r.shared = slot >= a.topk;
    if (r.shared) {
        r.valid = true; r.inter = a.I_shared;
        r.q13 = a.sw13.q; r.s13 = a.sw13.s; r.q2 = a.sw2.q; r.s2 = a.sw2.s;
    } else {
        const int e = a.ids[(size_t)b * a.topk + slot];
        r.valid = e >= 0; r.inter = a.I;
        const size_t ee = (size_t)(e < 0 ? 0 : e);
        r.q13 = reinterpret_cast<const char*>(a.w13.q) + ee * a.w13.q_stride;
        r.s13 = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.w13.s) + ee * a.w13.s_stride);
        r.q2 = reinterpret_cast<const char*>(a.w2.q) + ee * a.w2.q_stride;
        r.s2 = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.w2.s) + ee * a.w2.s_stride);
    }
    return r;
}

// stage 1: gu[b][slot][0..2I) = W13 . q(act[b])      grid = (tile groups, n_slots, B)
// The shared slot may carry one extra tile: the shared expert's sigmoid-gate row (decode.rs:3379-3390), N = 1.
template <int BITS>
__global__ void __launch_bounds__(KR_BLOCK) kr_moe_w13_kernel(const KrMoeArgs a, int tiles_per_wave) {
    const int slot = blockIdx.y, b = blockIdx.z;
    const KrSlot sl = kr_resolve_slot(a, b, slot);
    if (!sl.valid) return;
    const KrMatDev& m = sl.shared ? a.sw13 : a.w13;
    const int ntiles = (m.N + 7) / 8;
    const bool with_gate = sl.shared && a.sgate.q != nullptr;
    const int tile0 = (blockIdx.x * KR_WAVES) * tiles_per_wave;
    if (tile0 >= ntiles + (with_gate ? 1 : 0)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int first = tile0 + wave * tiles_per_wave;
    KrPre pre;
    const bool gate_wave = with_gate && first == ntiles;
    if (first < ntiles) kr_preload<BITS>(pre, sl.q13, sl.s13, m, first, lane, 0);
    else if (gate_wave) kr_preload<4>(pre, a.sgate.q, a.sgate.s, a.sgate, 0, lane, 0);
    const KrActLds L = kr_carve_lds(kr_smem, a.H, BITS == 8);
    if (a.act_f32) kr_prologue_quant_f32<BITS == 8>(a.act_f32 + (size_t)b * a.H, a.H, L, !(sl.shared && a.shared_decode));
    else kr_prologue_quant<uint16_t, BITS == 8>(a.act + (size_t)b * a.H, a.H, L);
    __syncthreads();
    float* gu = a.gu + ((size_t)b * a.n_slots + slot) * a.gu_ld;
    for (int t = 0; t < tiles_per_wave; t++) {
        const int tile = first + t;
        if (tile < ntiles) {
            const float acc = kr_matvec_tile<BITS>(pre, t == 0, sl.q13, sl.s13, m, tile, L, lane);
            const int col = tile * 8 + (lane >> 3);
            if ((lane & 7) == 0 && col < m.N) gu[col] = acc;
        } else if (with_gate && tile == ntiles) {
            const float acc = kr_matvec_tile<4>(pre, t == 0, a.sgate.q, a.sgate.s, a.sgate, 0, L, lane);
            if (lane == 0) a.gate_out[b] = acc;
        }
    }
}

// stage 2: eo[b][slot][0..H) = W2 . q(act_fn(gu[b][slot]))
template <int BITS, int ACT>
__global__ void __launch_bounds__(KR_BLOCK) kr_moe_w2_kernel(const KrMoeArgs a, int tiles_per_wave) {
    const int slot = blockIdx.y, b = blockIdx.z;
    const KrSlot sl = kr_resolve_slot(a, b, slot);
    if (!sl.valid) return;
    const KrMatDev& m = sl.shared ? a.sw2 : a.w2;
    const int ntiles = (m.N + 7) / 8;
    const int tile0 = (blockIdx.x * KR_WAVES) * tiles_per_wave;
    if (tile0 >= ntiles) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int first = tile0 + wave * tiles_per_wave;
    KrPre pre;
    if (first < ntiles) kr_preload<BITS>(pre, sl.q2, sl.s2, m, first, lane, 0);
    const KrActLds L = kr_carve_lds(kr_smem, sl.inter, BITS == 8);
    const float* gu = a.gu + ((size_t)b * a.n_slots + slot) * a.gu_ld;
    if (sl.shared && a.shared_decode) kr_prologue_hidden<KR_ACT_SILU_MUL, BITS == 8>(gu, sl.inter, a.swiglu_limit, a.alpha, L);
    else kr_prologue_hidden<ACT, BITS == 8>(gu, sl.inter, a.swiglu_limit, a.alpha, L);
    __syncthreads();
    float* eo = a.eo + ((size_t)b * a.n_slots + slot) * a.H;
    for (int t = 0; t < tiles_per_wave; t++) {
        const int tile = first + t;
        if (tile >= ntiles) break;
        const float acc = kr_matvec_tile<BITS>(pre, t == 0, sl.q2, sl.s2, m, tile, L, lane);
        const int col = tile * 8 + (lane >> 3);
        if ((lane & 7) == 0 && col < m.N) eo[col] = acc;
    }
}

// stage 3: out[b][j] = sum_i w_i * eo_i[j] in routing order (moe.rs:661-667); then rsf*out + shared (moe.rs:703-706)
__global__ void __launch_bounds__(KR_BLOCK) kr_moe_combine_kernel(const KrMoeArgs a) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * KR_BLOCK + threadIdx.x;
    if (j >= a.H) return;
    const float* eo = a.eo + (size_t)b * a.n_slots * a.H;
    float acc = 0.0f;
    for (int s = 0; s < a.topk; s++) {
        if (a.ids[(size_t)b * a.topk + s] < 0) continue;
        const float w = a.wts[(size_t)b * a.topk + s];
        acc += w * eo[(size_t)s * a.H + j];
    }
    if (a.n_slots > a.topk) acc = a.rsf * acc + eo[(size_t)a.topk * a.H + j];
    if (a.out_bf16) reinterpret_cast<uint16_t*>(a.out)[(size_t)b * a.H + j] = kr_f32_to_bf16(acc);
    else reinterpret_cast<float*>(a.out)[(size_t)b * a.H + j] = acc;
}

// y_i[N_i] = W_i . quant(x[K]) for up to KR_MAX_MULTI matrices that share the same input vector (q|k|v, qkvz|ba, ...):
// one launch, one activation prologue per workgroup, tiles of all matrices in one grid.
template <typename T, int BITS>
__global__ void __launch_bounds__(KR_BLOCK) kr_matvec_kernel(const KrMultiMat mm, const T* x, int tiles_per_wave, int act_mode) {
    const int total = mm.tile_end[mm.n - 1];
    const int tile0 = (blockIdx.x * KR_WAVES) * tiles_per_wave;
    if (tile0 >= total) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int first = tile0 + wave * tiles_per_wave;
    int mi = 0;
    while (mi + 1 < mm.n && first >= mm.tile_end[mi]) mi++;
    KrPre pre;
    if (first < total) kr_preload<BITS>(pre, mm.m[mi].q, mm.m[mi].s, mm.m[mi], first - (mi ? mm.tile_end[mi - 1] : 0), lane, 0);
    const int K = mm.m[0].ng * 128;
    const KrActLds L = kr_carve_lds(kr_smem, K, BITS == 8);
    if (act_mode == KR_ACT_SILU_MUL) kr_prologue_hidden<KR_ACT_SILU_MUL, BITS == 8>(reinterpret_cast<const float*>(x), K, 0.0f, 0.0f, L);
    else kr_prologue_quant<T, BITS == 8>(x, K, L);
    __syncthreads();
    for (int t = 0; t < tiles_per_wave; t++) {
        const int gt = first + t;
        if (gt >= total) break;
        while (mi + 1 < mm.n && gt >= mm.tile_end[mi]) mi++;
        const KrMatDev& m = mm.m[mi];
        const int tile = gt - (mi ? mm.tile_end[mi - 1] : 0);
        const float acc = kr_matvec_tile<BITS>(pre, t == 0, m.q, m.s, m, tile, L, lane);
        const int col = tile * 8 + (lane >> 3);
        if ((lane & 7) == 0 && col < m.N) mm.y[mi][col] = acc;
    }
}

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------

static int kr_pick_tpw(int K, int ntiles) {
    // keep >= ~1 KiB/lane-instruction streams but give short-K matrices more columns per wave
    int tpw = K >= 2048 ? 1 : (K >= 1024 ? 2 : 4);
    while (tpw > 1 && (ntiles + KR_WAVES * tpw - 1) / (KR_WAVES * tpw) < 32) tpw >>= 1;
    return tpw;
}

void kr_launch_moe_w13(const KrMoeArgs& a, hipStream_t st) {
    const bool has_shared = a.n_slots > a.topk;
    int nt = (a.w13.N + 7) / 8;
    if (has_shared && (a.sw13.N + 7) / 8 + (a.sgate.q ? 1 : 0) > nt) nt = (a.sw13.N + 7) / 8 + (a.sgate.q ? 1 : 0);
    const int tpw = kr_pick_tpw(a.H, nt);
    dim3 grid((nt + KR_WAVES * tpw - 1) / (KR_WAVES * tpw), a.n_slots, a.B);
    const size_t lds = kr_lds_bytes(a.H, a.w13.bits == 8);
    if (a.w13.bits == 4) hipLaunchKernelGGL(kr_moe_w13_kernel<4>, grid, dim3(KR_BLOCK), lds, st, a, tpw);
    else hipLaunchKernelGGL(kr_moe_w13_kernel<8>, grid, dim3(KR_BLOCK), lds, st, a, tpw);
}

void kr_launch_moe_w2(const KrMoeArgs& a, hipStream_t st) {
    const bool has_shared = a.n_slots > a.topk;
    const int nt = (a.H + 7) / 8;
    const int imax = has_shared && a.I_shared > a.I ? a.I_shared : a.I;
    const int tpw = kr_pick_tpw(a.I, nt);
    dim3 grid((nt + KR_WAVES * tpw - 1) / (KR_WAVES * tpw), a.n_slots, a.B);
    const size_t lds = kr_lds_bytes(imax, a.w2.bits == 8);
#define KR_W2(B_, A_) hipLaunchKernelGGL((kr_moe_w2_kernel<B_, A_>), grid, dim3(KR_BLOCK), lds, st, a, tpw)
    if (a.w2.bits == 4) {
        if (a.act_mode == KR_ACT_SILU_FUSED) KR_W2(4, KR_ACT_SILU_FUSED);
        else if (a.act_mode == KR_ACT_GPTOSS) KR_W2(4, KR_ACT_GPTOSS);
        else KR_W2(4, KR_ACT_SILU_MUL);
    } else {
        if (a.act_mode == KR_ACT_SILU_FUSED) KR_W2(8, KR_ACT_SILU_FUSED);
        else if (a.act_mode == KR_ACT_GPTOSS) KR_W2(8, KR_ACT_GPTOSS);
        else KR_W2(8, KR_ACT_SILU_MUL);
    }
#undef KR_W2
}

void kr_launch_moe_combine(const KrMoeArgs& a, hipStream_t st) {
    dim3 grid((a.H + KR_BLOCK - 1) / KR_BLOCK, a.B);
    hipLaunchKernelGGL(kr_moe_combine_kernel, grid, dim3(KR_BLOCK), 0, st, a);
}

void kr_launch_moe_decode(const KrMoeArgs& a, hipStream_t st) {
    kr_launch_moe_w13(a, st);
    kr_launch_moe_w2(a, st);
    kr_launch_moe_combine(a, st);
}

void kr_launch_multi_matvec(const KrMatDev* mats, float* const* ys, int n, const void* x, int x_is_f32, hipStream_t st, int act_mode) {
    KrMultiMat mm{};
    mm.n = n;
    int total = 0;
    for (int i = 0; i < n; i++) { mm.m[i] = mats[i]; mm.y[i] = ys[i]; total += (mats[i].N + 7) / 8; mm.tile_end[i] = total; }
    const int tpw = kr_pick_tpw(mats[0].K, total);
    dim3 grid((total + KR_WAVES * tpw - 1) / (KR_WAVES * tpw));
    const int bits = mats[0].bits;
    const size_t lds = kr_lds_bytes(mats[0].ng * 128, bits == 8);
    if (x_is_f32) {
        if (bits == 4) hipLaunchKernelGGL((kr_matvec_kernel<float, 4>), grid, dim3(KR_BLOCK), lds, st, mm, (const float*)x, tpw, act_mode);
        else hipLaunchKernelGGL((kr_matvec_kernel<float, 8>), grid, dim3(KR_BLOCK), lds, st, mm, (const float*)x, tpw, act_mode);
    } else {
        if (bits == 4) hipLaunchKernelGGL((kr_matvec_kernel<uint16_t, 4>), grid, dim3(KR_BLOCK), lds, st, mm, (const uint16_t*)x, tpw, act_mode);
        else hipLaunchKernelGGL((kr_matvec_kernel<uint16_t, 8>), grid, dim3(KR_BLOCK), lds, st, mm, (const uint16_t*)x, tpw, act_mode);
    }
}

void kr_launch_matvec(const KrMatDev& m, const void* x, int x_is_f32, float* y, hipStream_t st, int act_mode) {
    kr_launch_multi_matvec(&m, &y, 1, x, x_is_f32, st, act_mode);
}

// ------------------------------------------------------------------------------------------
// synthetic fill + bf16 reduce
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ uint64_t kr_splitmix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void kr_fill_words_kernel(uint32_t* q, size_t n_words, uint64_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x)
        q[i] = (uint32_t)kr_splitmix(seed + i);
}
// bf16 scales 0.005 + u*0.045, truncated to bf16 (decode.rs:4386-4392), two per word
__global__ void kr_fill_scales_kernel(uint32_t* s, size_t n_words, uint64_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t r = kr_splitmix(seed ^ 0xABCDEF12345ull ^ (i << 1));
        const float f0 = 0.005f + ((float)(uint32_t)r / 4294967295.0f) * 0.045f;
        const float f1 = 0.005f + ((float)(uint32_t)(r >> 32) / 4294967295.0f) * 0.045f;
        s[i] = (__float_as_uint(f0) >> 16) | (__float_as_uint(f1) & 0xFFFF0000u);
    }
}

void kr_launch_fill_synth(void* q, size_t q_bytes, uint32_t* s, size_t s_words, uint64_t seed, hipStream_t st) {
    hipLaunchKernelGGL(kr_fill_words_kernel, dim3(2048), dim3(256), 0, st, (uint32_t*)q, q_bytes / 4, seed);
    hipLaunchKernelGGL(kr_fill_scales_kernel, dim3(512), dim3(256), 0, st, s, s_words, seed);
}

// uniform f32 in [-amp, amp] and random FP16 KV patterns (decode.rs:4371-4375, 4402-4411), counter-hash based
__global__ void kr_fill_uniform_f32_kernel(float* x, size_t n, float amp, uint64_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int64_t b = (int64_t)kr_splitmix(seed * 0x100000001B3ull + i);
        x[i] = (float)((double)b / 9223372036854775807.0) * amp;
    }
}
__global__ void kr_fill_fp16_kv_kernel(uint16_t* x, size_t n, uint64_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t bits = kr_splitmix(seed * 0x100000001B3ull + i);
        const uint16_t sign = (uint16_t)((bits >> 15) & 1), ex = (uint16_t)(((bits >> 5) & 0xF) + 8), mant = (uint16_t)(bits & 0x3FF);
        x[i] = (uint16_t)((sign << 15) | (ex << 10) | mant);
    }
}
void kr_launch_fill_uniform_f32(float* x, size_t n, float amp, uint64_t seed, hipStream_t st) {
    hipLaunchKernelGGL(kr_fill_uniform_f32_kernel, dim3(2048), dim3(256), 0, st, x, n, amp, seed);
}
void kr_launch_fill_fp16_kv(uint16_t* x, size_t n, uint64_t seed, hipStream_t st) {
    hipLaunchKernelGGL(kr_fill_fp16_kv_kernel, dim3(1024), dim3(256), 0, st, x, n, seed);
}

// reduce_sum_bf16 (moe.rs:2505): f32 accumulate in input order, RNE to bf16
__global__ void kr_reduce_sum_bf16_kernel(const uint16_t* const* in, int n_in, uint16_t* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (n_in == 1) { out[i] = in[0][i]; continue; }
        float s = 0.0f;
        for (int p = 0; p < n_in; p++) s = s + kr_bf16_to_f32(in[p][i]);
        out[i] = kr_f32_to_bf16(s);
    }
}
void kr_launch_reduce_sum_bf16(const uint16_t* const* tbl, int n_in, uint16_t* out, size_t n, hipStream_t st) {
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(kr_reduce_sum_bf16_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, tbl, n_in, out, n);
}
