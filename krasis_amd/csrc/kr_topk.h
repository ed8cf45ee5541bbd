// kr_topk.h -- device helpers shared by the router kernels (kr_router.hip) and the tolerance-mode decode kernels (kr_decode_fast.hip):
// the decode-graph sigmoid (decode.rs:4110-4131), the serial emulation of topk_indices (decode.rs:1495-1535) used when leaders tie, and the
// wave-wide top-k in (value desc, index asc) order.
#pragma once
#include "kr_device.h"

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
static __device__ void kr_topk_heap_serial(const float* v, int n, int k, float* hv, int* hi, int32_t* out) {
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

// ---- wave-wide top-k in (value desc, index asc) order ----
// Element e lives in lane e / NV, slot e % NV ("lane-major"), so index order == (lane asc, slot asc): a 32-bit orderable image of the
// value is enough -- among equal values the first slot wins inside a lane (strict '>' scan) and the lowest lane wins across lanes (ballot + ffs).
// key 0 == empty / already taken.
__device__ __forceinline__ uint32_t kr_make_key(float v) {
    if (v == 0.0f) v = 0.0f;                       // -0 and +0 compare equal in the reference
    uint32_t u = __float_as_uint(v);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;    // monotone map float -> uint
    return u;
}
__device__ __forceinline__ float kr_key_value(uint32_t u) {
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    return __uint_as_float(u);
}
__device__ __forceinline__ uint32_t kr_umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t kr_wave_umax(uint32_t k) {
    k = kr_umax(k, (uint32_t)KR_DPP((int)k, KR_DPP_XOR1));
    k = kr_umax(k, (uint32_t)KR_DPP((int)k, KR_DPP_XOR2));
    k = kr_umax(k, (uint32_t)KR_DPP((int)k, KR_DPP_HALF_MIRROR));
    k = kr_umax(k, (uint32_t)KR_DPP((int)k, KR_DPP_MIRROR));    // every lane of a 16-lane row holds the row maximum
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)k, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)k, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)k, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)k, 48);
    return kr_umax(kr_umax(r0, r1), kr_umax(r2, r3));
}

// first kp1 elements in (value desc, index asc) order; element e = lane * NV + slot (registers)
template <int NV>
__device__ __forceinline__ void kr_topk_wave_reg(const float (&val)[NV], int n, int kp1, float* pv, int* pi) {
    const int lane = threadIdx.x & 63;
    uint32_t key[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) { const int e = lane * NV + i; key[i] = e < n ? kr_make_key(val[i]) : 0u; }
    for (int t = 0; t < kp1; t++) {
        uint32_t hk = 0u; int hs = 0;
#pragma unroll
        for (int i = 0; i < NV; i++) if (key[i] > hk) { hk = key[i]; hs = i; }
        const uint32_t wmax = kr_wave_umax(hk);
        const uint64_t mask = __ballot(hk == wmax);
        const int win = __builtin_ctzll(mask);
        const int slot = __builtin_amdgcn_readlane(hs, win);
#pragma unroll
        for (int i = 0; i < NV; i++) if (lane == win && i == slot) key[i] = 0u;
        if (lane == 0) { pv[t] = kr_key_value(wmax); pi[t] = win * NV + slot; }
    }
}
