// kr_router.hip -- MoE router for gfx950: gate GEMV + scoring + wavefront top-k, bit-exact ids.
//
// Two reference behaviours are reproduced:
//   rule DECODE  -- decode graph: moe_route_matmul_avx2 (src/decode.rs:1385) f32 gate x f32 hidden with two
//                   8-lane fma accumulators, moe_route_score_topk (decode.rs:4088) with the degree-4 poly sigmoid,
//                   topk_indices (decode.rs:1495) = size-k min-heap, strict '>' replacement, stable desc sort.
//   rule ENGINE  -- forward_moe_routed (src/moe.rs:3081-3246): bf16 gate x bf16 act, one sequential f32 sum per
//                   expert, libm sigmoid/softmax, k passes of argmax (lowest index wins ties).
//
// Layout: the gate matrix is stored "chain-major" so that the 16 virtual AVX lanes of an expert row are 16 GPU
// lanes, each walking its own fma chain with 16-byte loads:
//   f32 : G[e/4][c4 < H/64][(e%4)*16 + j][u < 4]  = gate[e][16*(4*c4+u) + j]
//   bf16: G[e/4][c8 < H/128][(e%4)*16 + j][u < 8] = gate[e][16*(8*c8+u) + j]   (used when every value is bf16-exact)
#include "kr_device.h"
#include "kr_libm.h"
#include "kr_router.h"
#include "kr_topk.h"

#ifdef KR_TIMING   // tools/probes/route_timing.hip: wall-clock stamps (10 ns units) of one wave, no-op in the product build
__device__ unsigned long long kr_stamps[32];
#define KR_STAMP(i) do { if ((threadIdx.x & 63) == 0) kr_stamps[i] = wall_clock64(); } while (0)
#else
#define KR_STAMP(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------
// logits, rule DECODE
// ------------------------------------------------------------------------------------------
template <bool GATE_BF16>
__global__ void __launch_bounds__(256) kr_route_logits_decode_kernel(const void* __restrict__ gate_cm, const float* __restrict__ x,
                                                                    const float* __restrict__ bias, float* __restrict__ logits,
                                                                    int E, int H) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [16][H/16 + 4] chain-major copy of x
    const int m = blockIdx.y;
    const int ld = H / 16 + 4;
    const float* xr = x + (size_t)m * H;
    for (int i = threadIdx.x; i < H; i += 256) xs[(i & 15) * ld + (i >> 4)] = xr[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int eb = blockIdx.x * 4 + wave;  // block of 4 experts
    if (eb * 4 >= E) return;
    const int j = lane & 15;
    float acc = 0.0f;
    const float* xj = xs + j * ld;
    if (GATE_BF16) {
        const int nc = H / 128;
        const u32x4* g = reinterpret_cast<const u32x4*>(gate_cm) + (size_t)eb * nc * 64 + lane;
        for (int c0 = 0; c0 < nc; c0 += 4) {
            u32x4 w[4];
#pragma unroll
            for (int u = 0; u < 4; u++) if (c0 + u < nc) w[u] = kr_ldg_nt(g + (size_t)(c0 + u) * 64);
#pragma unroll
            for (int u = 0; u < 4; u++) if (c0 + u < nc) {
                const float* xx = xj + (c0 + u) * 8;
                const uint32_t ww[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    acc = __builtin_fmaf(__uint_as_float(ww[p] << 16), xx[2 * p], acc);
                    acc = __builtin_fmaf(__uint_as_float(ww[p] & 0xFFFF0000u), xx[2 * p + 1], acc);
                }
            }
        }
    } else {
        const int nc = H / 64;
        const u32x4* g = reinterpret_cast<const u32x4*>(gate_cm) + (size_t)eb * nc * 64 + lane;
        for (int c0 = 0; c0 < nc; c0 += 8) {
            u32x4 w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) if (c0 + u < nc) w[u] = kr_ldg_nt(g + (size_t)(c0 + u) * 64);
#pragma unroll
            for (int u = 0; u < 8; u++) if (c0 + u < nc) {
                const float* xx = xj + (c0 + u) * 4;
                acc = __builtin_fmaf(__uint_as_float(w[u].x), xx[0], acc);
                acc = __builtin_fmaf(__uint_as_float(w[u].y), xx[1], acc);
                acc = __builtin_fmaf(__uint_as_float(w[u].z), xx[2], acc);
                acc = __builtin_fmaf(__uint_as_float(w[u].w), xx[3], acc);
            }
        }
    }
    // acc0 + acc1 (decode.rs:1419), then the hsum tree (:1420-1427): (l0+l4 + l1+l5) + (l2+l6 + l3+l7)
    float v = acc;
    v = v + __shfl_xor(v, 8);
    v = v + __shfl_xor(v, 4);
    v = v + __shfl_xor(v, 1);
    v = v + __shfl_xor(v, 2);
    const int e = eb * 4 + (lane >> 4);
    if (j == 0 && e < E) {
        if (bias) v += bias[e];  // decode.rs:3292-3294
        logits[(size_t)m * E + e] = v;
    }
}

// ------------------------------------------------------------------------------------------
// logits, rule ENGINE: bf16 x bf16 products are exact in f32; the sum is one sequential chain (moe.rs:3088-3096)
// gate layout: [e/64][H/8][64 lanes][8 bf16]
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) kr_route_logits_engine_kernel(const u32x4* __restrict__ gate_rm, const uint16_t* __restrict__ act,
                                                                   float* __restrict__ logits, int E, int H) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int m = blockIdx.y;
    for (int i = threadIdx.x; i < H; i += 64) xs[i] = kr_bf16_to_f32(act[(size_t)m * H + i]);
    __syncthreads();
    const int lane = threadIdx.x;
    const int e = blockIdx.x * 64 + lane;
    const u32x4* g = gate_rm + (size_t)blockIdx.x * (H / 8) * 64 + lane;
    float sum = 0.0f;
    for (int c0 = 0; c0 < H / 8; c0 += 8) {
        u32x4 w[8];
#pragma unroll
        for (int u = 0; u < 8; u++) w[u] = kr_ldg_nt(g + (size_t)(c0 + u) * 64);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float* xx = xs + (c0 + u) * 8;
            const uint32_t ww[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
            for (int p = 0; p < 4; p++) {
                sum += xx[2 * p] * __uint_as_float(ww[p] << 16);
                sum += xx[2 * p + 1] * __uint_as_float(ww[p] & 0xFFFF0000u);
            }
        }
    }
    if (e < E) logits[(size_t)m * E + e] = sum;
}

// ------------------------------------------------------------------------------------------
// scoring + top-k: one wave per token
// ------------------------------------------------------------------------------------------
struct KrRouteSelArgs {
    const float* logits;  // [m,E] (bias already added for rule DECODE)
    const float* esc;     // e_score_corr / correction_bias [E] or null
    int32_t* ids; float* w;
    int E, topk, scoring, norm, rule;
    int gptoss;           // rule ENGINE: swiglu_limit > 0 branch (moe.rs:3101)
};

// the selection is run by ONE wave (its own launch, or the last workgroup of the fused kernel whose other waves have left):
// LDS traffic of a single wave is ordered by a fence, no workgroup barrier involved
__device__ __forceinline__ void kr_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }

// softmax scoring without a correction bias: x -> fl(x * inv) is monotone, so the top-k of the exponentials is the top-k of the
// probabilities unless rounding makes two of the leading k+1 equal.  With DUAL a second wave of the workgroup selects on the
// exponentials while lane 0 of the first walks the 512-term sequential sum; equal scaled leaders fall back to the full selection.
__device__ __forceinline__ bool kr_route_dual_ok(const KrRouteSelArgs& a) {
    const bool decode = a.rule == 1;
    const bool raw_topk = decode ? (a.scoring == 2) : (a.gptoss != 0);
    return !raw_topk && a.scoring != 0 && a.esc == nullptr;
}
template <int NV, bool DUAL = false>
__device__ __forceinline__ void kr_route_select_body(const KrRouteSelArgs& a, const float* lg, int32_t* ids, float* w, float* sm) {
    float* scores = sm;            // [E]
    float* sel = sm + a.E;         // [E]   (serial tie fallback only)
    float* pv = sel + a.E;         // [33]
    int* pi = reinterpret_cast<int*>(pv + 33);      // [33]
    float* hv = reinterpret_cast<float*>(pi + 33);  // [32]
    int* hi = reinterpret_cast<int*>(hv + 32);      // [32]
    float* red = reinterpret_cast<float*>(hi + 32); // [2]
    const int lane = threadIdx.x & 63, E = a.E, k = a.topk;
    const bool decode = a.rule == 1;
    const bool raw_topk = decode ? (a.scoring == 2) : (a.gptoss != 0);
    float sc[NV], sl[NV];
    bool pre_selected = false;
    KR_STAMP(8);
#pragma unroll
    for (int i = 0; i < NV; i++) { const int e = lane * NV + i; sc[i] = e < E ? lg[e] : 0.0f; }
    if (raw_topk) {
#pragma unroll
        for (int i = 0; i < NV; i++) { const int e = lane * NV + i; if (e < E && !decode && a.esc) sc[i] += a.esc[e]; }
    } else if (a.scoring == 0) {
        const int e8 = (E / 8) * 8;
#pragma unroll
        for (int i = 0; i < NV; i++) { const int e = lane * NV + i; if (e < E) sc[i] = (decode && e < e8) ? kr_sigmoid_poly4(sc[i]) : 1.0f / (1.0f + kr_expf(-sc[i])); }
    } else {
        float mx = -__builtin_inff();
#pragma unroll
        for (int i = 0; i < NV; i++) { const int e = lane * NV + i; if (e < E) mx = fmaxf(mx, sc[i]); }
        mx = kr_red16_max_f32(mx);
        mx = fmaxf(fmaxf(__shfl(mx, 0), __shfl(mx, 16)), fmaxf(__shfl(mx, 32), __shfl(mx, 48)));
        const bool dual = DUAL && kr_route_dual_ok(a);
        const int wv_id = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < NV; i++) { const int e = lane * NV + i; if (e < E) { sc[i] = kr_expf(sc[i] - mx); if (!dual || wv_id == 0) scores[e] = sc[i]; } }
        if (dual) {
            if (wv_id == 0) {
                kr_wave_sync();
                if (lane == 0) red[0] = kr_seq_sum(scores, E);
            } else {
                const int npu = a.topk + 1 <= E ? a.topk + 1 : a.topk;
                kr_topk_wave_reg<NV>(sc, E, npu, pv, pi);        // on the exponentials
            }
            __syncthreads();                                     // the two surviving waves
            if (wv_id != 0) return;
            pre_selected = true;
        } else {
        kr_wave_sync();
        KR_STAMP(12);
        if (lane == 0) red[0] = kr_seq_sum(scores, E);   // decode.rs:4156 / moe.rs:3201: sum in index order
        KR_STAMP(13);
        kr_wave_sync();
        }
        const float se = red[0];
        if (decode) { const float inv = 1.0f / se;
#pragma unroll
            for (int i = 0; i < NV; i++) sc[i] *= inv; }
        else {
#pragma unroll
            for (int i = 0; i < NV; i++) sc[i] /= se; }
    }
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int e = lane * NV + i;
        sl[i] = (!raw_topk && a.esc && e < E) ? sc[i] + a.esc[e] : sc[i];
        if (e < E) { scores[e] = sc[i]; sel[e] = sl[i]; }
    }
    const int np = k + 1 <= E ? k + 1 : k;
    KR_STAMP(9);
    if (pre_selected) {   // scale the leaders found on the exponentials; they stand if strictly decreasing after rounding
        kr_wave_sync();
        const float se = red[0];
        float cv = lane < np ? pv[lane] : 0.0f;
        cv = decode ? cv * (1.0f / se) : cv / se;
        const float nx = __shfl_down(cv, 1);
        const bool strict = __ballot(lane + 1 < np && !(cv > nx)) == 0ull;
        if (strict) { if (lane < np) pv[lane] = cv; }
        else pre_selected = false;
        kr_wave_sync();
    }
    if (!pre_selected) kr_topk_wave_reg<NV>(sl, E, np, pv, pi);
    KR_STAMP(10);
    kr_wave_sync();
    // ---- epilogue: lanes 0..k-1 each own one selected expert; only the (short) sums stay sequential ----
    bool tie = false;
    if (decode) {
        const float pa = lane < np ? pv[lane] : 0.0f, pb = lane + 1 < np ? pv[lane + 1] : 0.0f;
        tie = __ballot(lane + 1 < np && pa == pb) != 0ull;
    }
    if (tie) {                                             // heap order governs ties (decode.rs:1531): exact serial emulation
        if (lane == 0) kr_topk_heap_serial(sel, E, k, hv, hi, pi);
        kr_wave_sync();
    }
    const int my = lane < k ? pi[lane] : 0;                // engine rule: lowest index wins == (value desc, index asc)
    float wv = lane < k ? scores[my] : 0.0f;
    if (raw_topk) {
        float mx = lane < k ? wv : -__builtin_inff();
        mx = kr_red16_max_f32(mx);
        mx = fmaxf(fmaxf(__shfl(mx, 0), __shfl(mx, 16)), fmaxf(__shfl(mx, 32), __shfl(mx, 48)));
        wv = lane < k ? kr_expf(wv - mx) : 0.0f;
    }
    if (raw_topk || a.norm) {
        if (lane < k) hv[lane] = wv;
        kr_wave_sync();
        if (lane == 0) { float se = 0.0f; for (int i = 0; i < k; i++) se += hv[i]; red[1] = se; }   // sequential, routing order
        kr_wave_sync();
        const float se = red[1];
        if (raw_topk) wv = decode ? wv * (1.0f / se) : wv / se;
        else if (se > 0.0f) wv = wv / se;
    }
    if (lane < k) { ids[lane] = my; w[lane] = wv; }
}

__global__ void __launch_bounds__(64) kr_route_select_kernel(const KrRouteSelArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int m = blockIdx.x;
    const float* lg = a.logits + (size_t)m * a.E;
    int32_t* ids = a.ids + (size_t)m * a.topk;
    float* w = a.w + (size_t)m * a.topk;
    const int nv = (a.E + 63) / 64;
    if (nv <= 1) kr_route_select_body<1>(a, lg, ids, w, sm);
    else if (nv <= 2) kr_route_select_body<2>(a, lg, ids, w, sm);
    else if (nv <= 4) kr_route_select_body<4>(a, lg, ids, w, sm);
    else if (nv <= 8) kr_route_select_body<8>(a, lg, ids, w, sm);
    else if (nv <= 16) kr_route_select_body<16>(a, lg, ids, w, sm);
    else kr_route_select_body<32>(a, lg, ids, w, sm);
}

// ------------------------------------------------------------------------------------------
// decode graph, M = 1: [fused add + RMSNorm] -> gate GEMV (rule DECODE) -> scoring + top-k, ONE launch.
//   * every workgroup rebuilds the normalised hidden vector itself (the reduction order is fixed -- 8 fma lanes + hsum,
//     decode.rs:1235-1252 -- so all workgroups get identical bits); workgroup 0 publishes hidden/residual for the expert kernels.
//     Inputs and outputs are different buffers because other workgroups may still be reading the inputs.
//   * the workgroup that finishes last (device-scope counter) runs the selection, so the logits never wait for another launch.
// ------------------------------------------------------------------------------------------
struct KrRouteFusedArgs {
    const void* gate_cm; const float* bias; float* logits; unsigned* counter;
    KrRouteSelArgs sel;
    const float* x;            // normalised hidden (norm_w == nullptr), else unused
    const float *hid_in, *res_in, *norm_w; float *hid_out, *res_out; float eps; int bias_one;   // folded fused_add_rmsnorm (decode.rs:1199)
    void *img_f32, *img_bf16;  // optional INT16 images of the normalised hidden for the expert launches (f32 quant: shared expert; bf16-rounded: routed)
    int H;
};

template <bool GATE_BF16, int NV>
__global__ void __launch_bounds__(256) kr_route_fused_decode_kernel(const KrRouteFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ int s_last;
    const int H = a.H, E = a.sel.E, ld = H / 16 + 4;
    float* xs = sm;                  // [16][ld] chain-major copy of x
    float* r = sm + 16 * ld;         // [H + 4] residual sum (norm fold) -- reused as selection scratch by the last workgroup
    const int ldt = H / 8 + 4;
    float* rt = r + H + 4;           // [8][ldt] lane-major copy of the residual sum for the sum-of-squares chain
    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63;
    const int eb = blockIdx.x * 4 + wave;  // block of 4 experts
    KR_STAMP(0);
    // the wave's gate rows do not depend on the hidden vector: the first KR_GPF 16-byte chunks per lane are in flight while the norm runs
    constexpr int KR_GPF = 16;
    const int ncg = GATE_BF16 ? H / 128 : H / 64;
    const int ebc = eb * 4 < E ? eb : (E - 1) / 4;      // waves past the last expert block re-read it (requests are never masked: the compiler's load count stays exact)
    const u32x4* gp = reinterpret_cast<const u32x4*>(a.gate_cm) + (size_t)ebc * ncg * 64 + lane;
    u32x4 gw[KR_GPF];
    // The norm's inputs are requested FIRST: a wave's memory counter is in-order, so behind the 16 gate chunks per lane they came back last and the norm
    // (the serial part of this launch) started when the whole gate prefetch had landed (round 5, docs/design/11-round5-decode-latency.md).
    const bool vec4 = a.norm_w && (H & 1023) == 0;
    float4 hv[4], rv[4];
    if (vec4) {
        const float4* h4 = reinterpret_cast<const float4*>(a.hid_in); const float4* r4 = reinterpret_cast<const float4*>(a.res_in);
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = u * 256 < H / 4 ? u * 256 + t : t; hv[u] = h4[i]; rv[u] = r4[i]; }      // clamped, never masked
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int u = 0; u < KR_GPF; u++) gw[u] = kr_ldg_nt(gp + (size_t)(u < ncg ? u : ncg - 1) * 64);
    if (a.norm_w) {
        if (vec4) {   // float4 loads, all in flight before the first add
            const float4* h4 = reinterpret_cast<const float4*>(a.hid_in); const float4* r4 = reinterpret_cast<const float4*>(a.res_in);
            for (int i0 = 0; i0 < H / 4; i0 += 1024) {
                if (i0 > 0) {
#pragma unroll
                    for (int u = 0; u < 4; u++) if (i0 + u * 256 < H / 4) { hv[u] = h4[i0 + u * 256 + t]; rv[u] = r4[i0 + u * 256 + t]; }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) if (i0 + u * 256 < H / 4) {
                    const float4 v = {hv[u].x + rv[u].x, hv[u].y + rv[u].y, hv[u].z + rv[u].z, hv[u].w + rv[u].w};
                    reinterpret_cast<float4*>(r)[i0 + u * 256 + t] = v;
                    { const int e = (i0 + u * 256 + t) * 4, bb = e >> 3, l0 = e & 7;    // elements e .. e + 3: lanes l0 .. l0 + 3 of block bb
                      rt[l0 * ldt + bb] = v.x; rt[(l0 + 1) * ldt + bb] = v.y; rt[(l0 + 2) * ldt + bb] = v.z; rt[(l0 + 3) * ldt + bb] = v.w; }
                    if (blockIdx.x == 0) reinterpret_cast<float4*>(a.res_out)[i0 + u * 256 + t] = v;
                }
            }
        } else
            for (int i = t; i < H; i += 256) { const float v = a.hid_in[i] + a.res_in[i]; r[i] = v; rt[(i & 7) * ldt + (i >> 3)] = v; if (blockIdx.x == 0) a.res_out[i] = v; }
        __syncthreads();
        if (t < 8) {
            float ss = kr_sumsq_lane_t(rt, ldt, H, t);
            ss = ss + __shfl_xor(ss, 4); ss = ss + __shfl_xor(ss, 1); ss = ss + __shfl_xor(ss, 2);
            if (t == 0) {
                for (int q = (H / 8) * 8; q < H; q++) ss += r[q] * r[q];
                r[H] = 1.0f / sqrtf(ss / (float)H + a.eps);
            }
        }
        __syncthreads();
        const float rms = r[H];
        for (int i = t; i < H; i += 256) {
            const float v = (r[i] * rms) * (a.bias_one ? (a.norm_w[i] + 1.0f) : a.norm_w[i]);
            xs[(i & 15) * ld + (i >> 4)] = v;
            r[i] = v;
            if (blockIdx.x == 0) a.hid_out[i] = v;
        }
        if (a.img_f32) {   // every workgroup holds the normalised vector: (image, 128-group) pairs are dealt round-robin, 16 lanes each
            __syncthreads();
            const int ng = H / 128;
            const KrActLds Lf = kr_carve_lds(reinterpret_cast<u32x4*>(a.img_f32), H, false), Lb = kr_carve_lds(reinterpret_cast<u32x4*>(a.img_bf16), H, false);
            for (int u = blockIdx.x * 16 + (t >> 4); u < 2 * ng; u += gridDim.x * 16) {
                const int g = u >> 1, c = g * 16 + (t & 15);
                const bool rb = u & 1;
                float v8[8];
                kr_load8(r, c, v8);
                float mx = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; i++) { if (rb) v8[i] = kr_bf16_to_f32(kr_f32_to_bf16(v8[i])); mx = fmaxf(mx, fabsf(v8[i])); }
                float scale, inv;
                kr_group_scale(mx, scale, inv);
                int q8[8];
                kr_quant8<false>(v8, inv, q8);
                const KrActLds& Lg = rb ? Lb : Lf;
                kr_store_chunk<false>(Lg, c, q8);
                if ((c & 15) == 0) Lg.ascale[c >> 4] = scale;
            }
        }
    } else {
        for (int i = t; i < H; i += 256) xs[(i & 15) * ld + (i >> 4)] = a.x[i];
    }
    __syncthreads();
    KR_STAMP(1);
    if (eb * 4 < E) {
        const int j = lane & 15;
        float acc = 0.0f;
        const float* xj = xs + j * ld;
        if (GATE_BF16) {
            const int nc = ncg;
            const u32x4* g = gp;
#pragma unroll
            for (int u = 0; u < KR_GPF; u++) if (u < nc) {
                const float* xx = xj + u * 8;
                const uint32_t ww[4] = {gw[u].x, gw[u].y, gw[u].z, gw[u].w};
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    acc = __builtin_fmaf(__uint_as_float(ww[p] << 16), xx[2 * p], acc);
                    acc = __builtin_fmaf(__uint_as_float(ww[p] & 0xFFFF0000u), xx[2 * p + 1], acc);
                }
            }
            for (int c0 = KR_GPF; c0 < nc; c0 += 4) {
                u32x4 w[4];
#pragma unroll
                for (int u = 0; u < 4; u++) if (c0 + u < nc) w[u] = kr_ldg_nt(g + (size_t)(c0 + u) * 64);
#pragma unroll
                for (int u = 0; u < 4; u++) if (c0 + u < nc) {
                    const float* xx = xj + (c0 + u) * 8;
                    const uint32_t ww[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        acc = __builtin_fmaf(__uint_as_float(ww[p] << 16), xx[2 * p], acc);
                        acc = __builtin_fmaf(__uint_as_float(ww[p] & 0xFFFF0000u), xx[2 * p + 1], acc);
                    }
                }
            }
        } else {
            const int nc = ncg;
            const u32x4* g = gp;
#pragma unroll
            for (int u = 0; u < KR_GPF; u++) if (u < nc) {
                const float* xx = xj + u * 4;
                acc = __builtin_fmaf(__uint_as_float(gw[u].x), xx[0], acc);
                acc = __builtin_fmaf(__uint_as_float(gw[u].y), xx[1], acc);
                acc = __builtin_fmaf(__uint_as_float(gw[u].z), xx[2], acc);
                acc = __builtin_fmaf(__uint_as_float(gw[u].w), xx[3], acc);
            }
            for (int c0 = KR_GPF; c0 < nc; c0 += 8) {
                u32x4 w[8];
#pragma unroll
                for (int u = 0; u < 8; u++) if (c0 + u < nc) w[u] = kr_ldg_nt(g + (size_t)(c0 + u) * 64);
#pragma unroll
                for (int u = 0; u < 8; u++) if (c0 + u < nc) {
                    const float* xx = xj + (c0 + u) * 4;
                    acc = __builtin_fmaf(__uint_as_float(w[u].x), xx[0], acc);
                    acc = __builtin_fmaf(__uint_as_float(w[u].y), xx[1], acc);
                    acc = __builtin_fmaf(__uint_as_float(w[u].z), xx[2], acc);
                    acc = __builtin_fmaf(__uint_as_float(w[u].w), xx[3], acc);
                }
            }
        }
        float v = acc;
        v = v + __shfl_xor(v, 8);
        v = v + __shfl_xor(v, 4);
        v = v + __shfl_xor(v, 1);
        v = v + __shfl_xor(v, 2);
        const int e = eb * 4 + (lane >> 4);
        if (j == 0 && e < E) {
            if (a.bias) v += a.bias[e];
            __hip_atomic_store(a.logits + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- last workgroup selects ----
    KR_STAMP(2);
    __syncthreads();
    if (t == 0) {
        const unsigned prev = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == gridDim.x - 1;
        if (s_last) __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch / graph replay
    }
    __syncthreads();
    const bool dual = kr_route_dual_ok(a.sel);
    if (!s_last || wave > (dual ? 1 : 0)) return;
    KR_STAMP(3);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);   // every workgroup's logits are visible (released by its counter increment)
    kr_route_select_body<NV, true>(a.sel, a.logits, a.sel.ids, a.sel.w, sm);
    KR_STAMP(11);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void kr_launch_route_logits_decode(const void* gate_cm, int gate_bf16, const float* x, const float* bias, float* logits,
                                   int m, int E, int H, hipStream_t st) {
    dim3 grid((E / 4 + 3) / 4, m);
    const size_t lds = (size_t)16 * (H / 16 + 4) * 4;
    if (gate_bf16) hipLaunchKernelGGL(kr_route_logits_decode_kernel<true>, grid, dim3(256), lds, st, gate_cm, x, bias, logits, E, H);
    else hipLaunchKernelGGL(kr_route_logits_decode_kernel<false>, grid, dim3(256), lds, st, gate_cm, x, bias, logits, E, H);
}
void kr_launch_route_logits_engine(const void* gate_rm, const uint16_t* act, float* logits, int m, int E, int H, hipStream_t st) {
    dim3 grid((E + 63) / 64, m);
    hipLaunchKernelGGL(kr_route_logits_engine_kernel, grid, dim3(64), (size_t)H * 4, st, (const u32x4*)gate_rm, act, logits, E, H);
}
void kr_launch_route_select(const float* logits, const float* esc, int32_t* ids, float* w, int m, int E, int topk, int scoring,
                            int norm, int rule, int gptoss, hipStream_t st) {
    KrRouteSelArgs a{logits, esc, ids, w, E, topk, scoring, norm, rule, gptoss};
    const size_t lds = (size_t)(2 * E + 33 + 33 + 32 + 32 + 4) * 4;
    hipLaunchKernelGGL(kr_route_select_kernel, dim3(m), dim3(64), lds, st, a);
}

int kr_launch_route_fused_decode(const void* gate_cm, int gate_bf16, const float* bias, float* logits, unsigned* counter, const float* esc,
                                 int32_t* ids, float* w, int E, int H, int topk, int scoring, int norm_topk, const float* x,
                                 const float* hid_in, const float* res_in, const float* norm_w, float* hid_out, float* res_out, float eps,
                                 int bias_one, hipStream_t st, void* img_f32, void* img_bf16) {
    KrRouteFusedArgs a{};
    a.img_f32 = (norm_w && img_f32 && img_bf16) ? img_f32 : nullptr; a.img_bf16 = a.img_f32 ? img_bf16 : nullptr;
    a.gate_cm = gate_cm; a.bias = bias; a.logits = logits; a.counter = counter;
    a.sel = KrRouteSelArgs{logits, esc, ids, w, E, topk, scoring, norm_topk, 1 /* KR_ROUTE_RULE_DECODE */, 0};
    a.x = x; a.hid_in = hid_in; a.res_in = res_in; a.norm_w = norm_w; a.hid_out = hid_out; a.res_out = res_out; a.eps = eps; a.bias_one = bias_one; a.H = H;
    const int nv = (E + 63) / 64;
    if (H % 128 || topk > 32) return 1;
    dim3 grid((E / 4 + 3) / 4);
    const size_t sel_f = (size_t)(2 * E + 33 + 33 + 32 + 32 + 4);
    const size_t norm_f = (size_t)16 * (H / 16 + 4) + (size_t)H + 4 + (size_t)8 * (H / 8 + 4);
    const size_t lds = (norm_f > sel_f ? norm_f : sel_f) * 4;
#define KR_RF(B, N) hipLaunchKernelGGL((kr_route_fused_decode_kernel<B, N>), grid, dim3(256), lds, st, a)
    if (gate_bf16) { if (nv <= 1) KR_RF(true, 1); else if (nv <= 2) KR_RF(true, 2); else if (nv <= 4) KR_RF(true, 4); else if (nv <= 8) KR_RF(true, 8); else return 1; }
    else { if (nv <= 1) KR_RF(false, 1); else if (nv <= 2) KR_RF(false, 2); else if (nv <= 4) KR_RF(false, 4); else if (nv <= 8) KR_RF(false, 8); else return 1; }
#undef KR_RF
    return 0;
}
