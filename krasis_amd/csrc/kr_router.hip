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
__device__ __forceinline__ float kr_sigmoid_poly4(float x) {  // decode.rs:4110-4131
    const float t = (0.0f - x) * 1.4426950408889634f;
    const float n = floorf(t);
    const int ni = (int)n;
    const float f = t - n;
    const float p = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(__builtin_fmaf(0.009518f, f, 0.0558011f), f, 0.2402265f), f, 0.6931472f), f, 1.0f);
    return 1.0f / (1.0f + p * __int_as_float((ni + 127) << 23));
}

// better(a, b): a precedes b in (value desc, index asc)
__device__ __forceinline__ bool kr_better(float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia < ib); }

// exact emulation of topk_indices (decode.rs:1495-1535), run by one lane
__device__ void kr_topk_heap_serial(const float* v, int n, int k, float* hv, int* hi, int32_t* out) {
    for (int i = 0; i < k; i++) { hv[i] = v[i]; hi[i] = i; }
    for (int i = 1; i < k; i++) {  // stable ascending insertion sort
        float xv = hv[i]; int xi = hi[i]; int j = i - 1;
        while (j >= 0 && hv[j] > xv) { hv[j + 1] = hv[j]; hi[j + 1] = hi[j]; j--; }
        hv[j + 1] = xv; hi[j + 1] = xi;
    }
    for (int i = k; i < n; i++) {
        if (v[i] > hv[0]) {
            hv[0] = v[i]; hi[0] = i;
            int pos = 0;
            for (;;) {
                const int left = 2 * pos + 1, right = 2 * pos + 2; int smallest = pos;
                if (left < k && hv[left] < hv[smallest]) smallest = left;
                if (right < k && hv[right] < hv[smallest]) smallest = right;
                if (smallest == pos) break;
                const float tv = hv[pos]; const int ti = hi[pos];
                hv[pos] = hv[smallest]; hi[pos] = hi[smallest]; hv[smallest] = tv; hi[smallest] = ti;
                pos = smallest;
            }
        }
    }
    for (int i = 1; i < k; i++) {  // stable descending
        float xv = hv[i]; int xi = hi[i]; int j = i - 1;
        while (j >= 0 && hv[j] < xv) { hv[j + 1] = hv[j]; hi[j + 1] = hi[j]; j--; }
        hv[j + 1] = xv; hi[j + 1] = xi;
    }
    for (int i = 0; i < k; i++) out[i] = hi[i];
}

// ---- wave-wide (value desc, index asc) arg-best over 64 lanes on packed 64-bit keys ----
// key = orderable(value) << 32 | (0xFFFFFFFF - index); larger key == better; 0 == empty / already taken.
__device__ __forceinline__ uint64_t kr_make_key(float v, int e) {
    if (v == 0.0f) v = 0.0f;                       // -0 and +0 compare equal in the reference
    uint32_t u = __float_as_uint(v);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;    // monotone map float -> uint
    return ((uint64_t)u << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)e);
}
__device__ __forceinline__ float kr_key_value(uint64_t k) {
    uint32_t u = (uint32_t)(k >> 32);
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    return __uint_as_float(u);
}
__device__ __forceinline__ int kr_key_index(uint64_t k) { return (int)(0xFFFFFFFFu - (uint32_t)k); }
__device__ __forceinline__ uint64_t kr_key_max(uint64_t a, uint64_t b) { return a > b ? a : b; }
#define KR_DPP64(k, ctrl) (((uint64_t)(uint32_t)KR_DPP((int)((k) >> 32), ctrl) << 32) | (uint32_t)KR_DPP((int)(uint32_t)(k), ctrl))
__device__ __forceinline__ uint64_t kr_wave_key_max(uint64_t k) {
    k = kr_key_max(k, KR_DPP64(k, KR_DPP_XOR1));
    k = kr_key_max(k, KR_DPP64(k, KR_DPP_XOR2));
    k = kr_key_max(k, KR_DPP64(k, KR_DPP_HALF_MIRROR));
    k = kr_key_max(k, KR_DPP64(k, KR_DPP_MIRROR));    // every lane of a 16-lane row holds the row maximum
    uint64_t r[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
        r[j] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(k >> 32), 16 * j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k, 16 * j);
    return kr_key_max(kr_key_max(r[0], r[1]), kr_key_max(r[2], r[3]));
}

// first kp1 elements in (value desc, index asc) order; element e lives in lane e%64, slot e/64 (registers)
template <int NV>
__device__ __forceinline__ void kr_topk_wave_reg(const float (&val)[NV], int n, int kp1, float* pv, int* pi) {
    const int lane = threadIdx.x & 63;
    uint64_t key[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) { const int e = i * 64 + lane; key[i] = e < n ? kr_make_key(val[i], e) : 0ull; }
    for (int t = 0; t < kp1; t++) {
        uint64_t best = 0ull;
#pragma unroll
        for (int i = 0; i < NV; i++) best = kr_key_max(best, key[i]);
        const uint64_t w = kr_wave_key_max(best);
#pragma unroll
        for (int i = 0; i < NV; i++) if (key[i] == w) key[i] = 0ull;   // keys are unique (index embedded): removes exactly the winner
        if (lane == 0) { pv[t] = kr_key_value(w); pi[t] = kr_key_index(w); }
    }
}

struct KrRouteSelArgs {
    const float* logits;  // [m,E] (bias already added for rule DECODE)
    const float* esc;     // e_score_corr / correction_bias [E] or null
    int32_t* ids; float* w;
    int E, topk, scoring, norm, rule;
    int gptoss;           // rule ENGINE: swiglu_limit > 0 branch (moe.rs:3101)
};

// strictly sequential f32 sum of x[0..n) (reference order), register-batched so LDS latency is paid once per 16 values
__device__ __forceinline__ float kr_seq_sum(const float* x, int n) {
    float s = 0.0f; int e = 0;
    for (; e + 16 <= n; e += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = x[e + u];
#pragma unroll
        for (int u = 0; u < 16; u++) s += v[u];
    }
    for (; e < n; e++) s += x[e];
    return s;
}

template <int NV>
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
#pragma unroll
    for (int i = 0; i < NV; i++) { const int e = i * 64 + lane; sc[i] = e < E ? lg[e] : 0.0f; }
    if (raw_topk) {
#pragma unroll
        for (int i = 0; i < NV; i++) { const int e = i * 64 + lane; if (e < E && !decode && a.esc) sc[i] += a.esc[e]; }
    } else if (a.scoring == 0) {
        const int e8 = (E / 8) * 8;
#pragma unroll
        for (int i = 0; i < NV; i++) { const int e = i * 64 + lane; if (e < E) sc[i] = (decode && e < e8) ? kr_sigmoid_poly4(sc[i]) : 1.0f / (1.0f + kr_expf(-sc[i])); }
    } else {
        float mx = -__builtin_inff();
#pragma unroll
        for (int i = 0; i < NV; i++) { const int e = i * 64 + lane; if (e < E) mx = fmaxf(mx, sc[i]); }
        mx = kr_red16_max_f32(mx);
        mx = fmaxf(fmaxf(__shfl(mx, 0), __shfl(mx, 16)), fmaxf(__shfl(mx, 32), __shfl(mx, 48)));
#pragma unroll
        for (int i = 0; i < NV; i++) { const int e = i * 64 + lane; if (e < E) { sc[i] = kr_expf(sc[i] - mx); scores[e] = sc[i]; } }
        __syncthreads();
        if (lane == 0) red[0] = kr_seq_sum(scores, E);   // decode.rs:4156 / moe.rs:3201: sum in index order
        __syncthreads();
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
        const int e = i * 64 + lane;
        sl[i] = (!raw_topk && a.esc && e < E) ? sc[i] + a.esc[e] : sc[i];
        if (e < E) { scores[e] = sc[i]; sel[e] = sl[i]; }
    }
    const int np = k + 1 <= E ? k + 1 : k;
    kr_topk_wave_reg<NV>(sl, E, np, pv, pi);
    __syncthreads();
    if (lane == 0) {
        bool tie = false;
        if (decode) for (int i = 0; i + 1 < np; i++) tie |= (pv[i] == pv[i + 1]);
        if (tie) kr_topk_heap_serial(sel, E, k, hv, hi, ids);  // heap order governs ties (decode.rs:1531)
        else for (int i = 0; i < k; i++) ids[i] = pi[i];       // engine rule: lowest index wins == (value desc, index asc)
        if (raw_topk) {
            float mx = -__builtin_inff();
            for (int i = 0; i < k; i++) mx = fmaxf(mx, scores[ids[i]]);
            float se = 0.0f;
            for (int i = 0; i < k; i++) { const float v = kr_expf(scores[ids[i]] - mx); w[i] = v; se += v; }
            if (decode) { const float inv = 1.0f / se; for (int i = 0; i < k; i++) w[i] *= inv; }
            else for (int i = 0; i < k; i++) w[i] /= se;
        } else {
            for (int i = 0; i < k; i++) w[i] = scores[ids[i]];
            if (a.norm) {
                float s = 0.0f;
                for (int i = 0; i < k; i++) s += w[i];
                if (s > 0.0f) for (int i = 0; i < k; i++) w[i] /= s;
            }
        }
    }
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
