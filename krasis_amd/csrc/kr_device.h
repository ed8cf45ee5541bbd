// kr_device.h -- device-side helpers shared by the gfx950 kernels (wave64 only).
//
// Numerics contract: every kernel reproduces the evaluation order of the reference's CPU
// kernels so results are bit-identical to the oracle.  This translation unit is
// compiled with -ffp-contract=off; fused multiply-adds appear only where the reference issues
// _mm256_fmadd_ps and are written as __builtin_fmaf explicitly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KR_WAVE 64
#define KR_DPP_XOR1 0xB1        // quad_perm [1,0,3,2]
#define KR_DPP_XOR2 0x4E        // quad_perm [2,3,0,1]
#define KR_DPP_HALF_MIRROR 0x141 // lane i <-> 7-i inside each 8-lane half row
#define KR_DPP_MIRROR 0x140      // lane i <-> 15-i inside each 16-lane row

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float kr_bf16_to_f32(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
// marlin.rs:25 -- RNE without NaN special-casing
__device__ __forceinline__ uint16_t kr_f32_to_bf16(float f) {
    uint32_t b = __float_as_uint(f);
    b += 0x7FFFu + ((b >> 16) & 1u);
    return (uint16_t)(b >> 16);
}

#define KR_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, false)

// sum / max over each aligned group of 8 lanes; every lane of the group receives the result
__device__ __forceinline__ int kr_red8_add_i32(int v) {
    v += KR_DPP(v, KR_DPP_XOR1);
    v += KR_DPP(v, KR_DPP_XOR2);
    v += KR_DPP(v, KR_DPP_HALF_MIRROR);
    return v;
}
__device__ __forceinline__ float kr_red16_max_f32(float v) {
    v = fmaxf(v, __int_as_float(KR_DPP(__float_as_int(v), KR_DPP_XOR1)));
    v = fmaxf(v, __int_as_float(KR_DPP(__float_as_int(v), KR_DPP_XOR2)));
    v = fmaxf(v, __int_as_float(KR_DPP(__float_as_int(v), KR_DPP_HALF_MIRROR)));
    v = fmaxf(v, __int_as_float(KR_DPP(__float_as_int(v), KR_DPP_MIRROR)));
    return v;
}
__device__ __forceinline__ float kr_red4_max_f32(float v) {
    v = fmaxf(v, __int_as_float(KR_DPP(__float_as_int(v), KR_DPP_XOR1)));
    v = fmaxf(v, __int_as_float(KR_DPP(__float_as_int(v), KR_DPP_XOR2)));
    return v;
}

// streamed-once weight loads: non-temporal 16 B / 4 B
__device__ __forceinline__ u32x4 kr_ldg_nt(const u32x4* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ uint32_t kr_ldg_nt(const uint32_t* p) { return __builtin_nontemporal_load(p); }

// fast_exp_avx2 (avx2.rs:2235): 2^(x*log2e) with the degree-5 polynomial, fma Horner
__device__ __forceinline__ float kr_fast_exp_poly5(float x) {
    const float t = x * 1.4426950408889634f;
    const float n = floorf(t);
    const int ni = (int)n;
    const float f = t - n;
    float p = __builtin_fmaf(0.0013333558f, f, 0.009618129f);
    p = __builtin_fmaf(p, f, 0.0555041f);
    p = __builtin_fmaf(p, f, 0.2402265f);
    p = __builtin_fmaf(p, f, 0.6931472f);
    p = __builtin_fmaf(p, f, 1.0f);
    return p * __int_as_float((ni + 127) << 23);
}
// fast_sigmoid_avx2 (avx2.rs:2277) with the reciprocal done as an IEEE divide (oracle mode KRO_SIG_POLY5_DIV;
// the reference's rcpps+Newton step is host-CPU specific and differs from this by <= 2 ulp)
__device__ __forceinline__ float kr_sigmoid_poly5(float x) {
    float c = 0.0f - x;
    c = c < 20.0f ? c : 20.0f;
    c = c > -20.0f ? c : -20.0f;
    return 1.0f / (1.0f + kr_fast_exp_poly5(c));
}
// scalar twin fast_sigmoid (moe.rs:1216): non-fused Horner
__device__ __forceinline__ float kr_sigmoid_poly5_scalar(float x) {
    float nx = -x;
    nx = nx < -20.0f ? -20.0f : nx;
    nx = nx > 20.0f ? 20.0f : nx;
    const float t = nx * 1.4426950408889634f;
    const float n = floorf(t);
    const float f = t - n;
    const float p = 1.0f + f * (0.6931472f + f * (0.2402265f + f * (0.0555041f + f * (0.009618129f + f * 0.0013333558f))));
    return 1.0f / (1.0f + p * __int_as_float(((int)n + 127) << 23));
}

// ---------------------------------------------------------------------------------------------
// INT16 activation image in LDS (gs = 128).  For every 8 consecutive k ("chunk") one 16-byte record
//   { AH_even, AH_odd, AL_even, AL_odd }  with  a = AH*256 + AL,  AH = a >> 8 (signed), AL = a & 255
//   *_even packs k = 0,2,4,6 of the chunk, *_odd packs k = 1,3,5,7 -- the order in which
//   (w & 0x0F0F0F0F) and ((w >> 4) & 0x0F0F0F0F) present the nibbles of a packed INT4 word.
// plus   asum16[k/16] = sum of the 16 quantized activations (for the "-8" nibble offset)
// and    ascale[k/128] = the per-group activation scale.
// INT8 weights use the natural-order image { AH[0..3] .. } described at kr_act_image_i8.
// ---------------------------------------------------------------------------------------------
struct KrActLds {
    u32x4* planes;   // [K/8]
    int* asum16;     // [K/16]
    float* ascale;   // [K/128]
    u32x4* planes8;  // INT8 image: [K/16][2] = {AH(16 k), AL'(16 k)} natural order, AL' = AL - 128
};

__device__ __forceinline__ uint32_t kr_pack4(int b0, int b1, int b2, int b3) {
    return (uint32_t)(b0 & 0xFF) | ((uint32_t)(b1 & 0xFF) << 8) | ((uint32_t)(b2 & 0xFF) << 16) | ((uint32_t)(b3 & 0xFF) << 24);
}

// Quantize 8 values (one chunk) given the group's inverse scale; ROUND_EVEN selects
// _mm256_cvtps_epi32 semantics (avx2.rs:2357) vs f32::round (avx2.rs:264,300).
template <bool ROUND_EVEN>
__device__ __forceinline__ void kr_quant8(const float (&x)[8], float inv, int (&q)[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float s = x[i] * inv;
        int v = ROUND_EVEN ? __float2int_rn(s) : (int)roundf(s);
        v = v > 32767 ? 32767 : v;
        v = v < -32768 ? -32768 : v;
        q[i] = v;
    }
}

// Store one chunk's records.  `chunk` is the global chunk index (k/8).  Pair sums for asum16 are
// completed with the neighbouring lane (chunks are assigned lane-consecutively by every caller).
template <bool WANT_I8_IMAGE>
__device__ __forceinline__ void kr_store_chunk(const KrActLds& L, int chunk, const int (&q)[8]) {
    u32x4 r;
    r.x = kr_pack4(q[0] >> 8, q[2] >> 8, q[4] >> 8, q[6] >> 8);
    r.y = kr_pack4(q[1] >> 8, q[3] >> 8, q[5] >> 8, q[7] >> 8);
    r.z = kr_pack4(q[0], q[2], q[4], q[6]);
    r.w = kr_pack4(q[1], q[3], q[5], q[7]);
    L.planes[chunk] = r;
    int s = q[0] + q[1] + q[2] + q[3] + q[4] + q[5] + q[6] + q[7];
    s += KR_DPP(s, KR_DPP_XOR1);
    if ((chunk & 1) == 0) L.asum16[chunk >> 1] = s;
    if (WANT_I8_IMAGE) {
        // natural order: half-record per chunk: AH[0..7] (2 dwords) and AL'[0..7] (2 dwords)
        uint32_t* p = reinterpret_cast<uint32_t*>(L.planes8) + (chunk >> 1) * 8 + (chunk & 1) * 2;
        p[0] = kr_pack4(q[0] >> 8, q[1] >> 8, q[2] >> 8, q[3] >> 8);
        p[1] = kr_pack4(q[4] >> 8, q[5] >> 8, q[6] >> 8, q[7] >> 8);
        p[4] = kr_pack4((q[0] & 255) - 128, (q[1] & 255) - 128, (q[2] & 255) - 128, (q[3] & 255) - 128);
        p[5] = kr_pack4((q[4] & 255) - 128, (q[5] & 255) - 128, (q[6] & 255) - 128, (q[7] & 255) - 128);
    }
}

// ---------------------------------------------------------------------------------------------
// KV-cache element codecs.  FP16 is the reference's CPU-decode cache (VCVTPS2PH RNE, decode.rs:4464-4478); FP8-E4M3 (OCP "fn": bias 7,
// no infinities, 0x7F/0xFF = NaN, max 448) is the reference's GPU cache dtype (python/krasis/kv_cache.py:38-135, torch.float8_e4m3fn).
// f32 -> e4m3 follows torch's conversion: round to nearest even, |x| beyond the largest finite value becomes NaN (no saturation).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t kr_f32_to_e4m3(float f) {
    const uint32_t b = __float_as_uint(f), sign = (b >> 24) & 0x80u, a = b & 0x7FFFFFFFu;
    if (a >= 0x43F00000u) return (uint8_t)(sign | 0x7Fu);              // |x| >= 480 (or NaN/inf): NaN, like c10::Float8_e4m3fn
    if (a < 0x3C800000u) {                                             // |x| < 2^-6: subnormal result, RNE via the magic-add trick
        const float t = __uint_as_float(a) + __uint_as_float(0x46800000u);   // + 2^14: mantissa LSB == 2^-9
        return (uint8_t)(sign | ((__float_as_uint(t) - 0x46800000u) & 0xFFu));
    }
    uint32_t r = a + 0x7FFFFu + ((a >> 20) & 1u);                      // RNE at bit 20
    r = (r - 0x3C000000u) >> 20;                                       // rebias 127 -> 7, keep 4+3 bits
    return (uint8_t)(sign | (r & 0x7Fu));
}
__device__ __forceinline__ float kr_e4m3_to_f32(uint8_t x) {
    // gfx950's v_cvt_f32_fp8 reads OCP E4M3 (the "fn" format of torch.float8_e4m3fn): exact for all 254 finite codes (subnormals, both zeros,
    // +-448; tests/test_decode_gpu.py runs every one through it), NaN for 0x7F / 0xFF
    return __builtin_amdgcn_cvt_f32_fp8((int)x, 0);
}
__device__ __forceinline__ float kr_kv_load(const void* base, size_t i, int fp8) {
    if (fp8) return kr_e4m3_to_f32(reinterpret_cast<const uint8_t*>(base)[i]);
    const uint16_t h = reinterpret_cast<const uint16_t*>(base)[i];
    _Float16 hv; __builtin_memcpy(&hv, &h, 2);
    return (float)hv;
}
__device__ __forceinline__ void kr_kv_store(void* base, size_t i, float v, int fp8) {
    if (fp8) { reinterpret_cast<uint8_t*>(base)[i] = kr_f32_to_e4m3(v); return; }
    const _Float16 hv = (_Float16)v;                                    // v_cvt_f16_f32: round to nearest even
    uint16_t h; __builtin_memcpy(&h, &hv, 2);
    reinterpret_cast<uint16_t*>(base)[i] = h;
}

// strictly sequential f32 sum of x[0..n) (reference order); the next 16 values are fetched from LDS (4 x ds_read_b128) while the
// current 16 are being added, so the chain of dependent adds is the only latency left.  x must be 16-byte aligned.
#define KR_ADD16(s, a0, a1, a2, a3) do { s += a0.x; s += a0.y; s += a0.z; s += a0.w; s += a1.x; s += a1.y; s += a1.z; s += a1.w; \
                                           s += a2.x; s += a2.y; s += a2.z; s += a2.w; s += a3.x; s += a3.y; s += a3.z; s += a3.w; } while (0)
__device__ __forceinline__ float kr_seq_sum(const float* x, int n, float s = 0.0f) {   // s: the running sum so far (tiled callers)
    int e = 0;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    if (n >= 32) {   // two register sets (A, B): B's ds_reads are in flight while A is added and vice versa
        float4 a0 = x4[0], a1 = x4[1], a2 = x4[2], a3 = x4[3];
        for (; e + 64 <= n; e += 32) {
            const float4 b0 = x4[e / 4 + 4], b1 = x4[e / 4 + 5], b2 = x4[e / 4 + 6], b3 = x4[e / 4 + 7];
            __builtin_amdgcn_sched_barrier(0);
            KR_ADD16(s, a0, a1, a2, a3);
            a0 = x4[e / 4 + 8]; a1 = x4[e / 4 + 9]; a2 = x4[e / 4 + 10]; a3 = x4[e / 4 + 11];
            __builtin_amdgcn_sched_barrier(0);
            KR_ADD16(s, b0, b1, b2, b3);
        }
        // a = block at e (loaded); at least 32 and fewer than 64 values remain
        const float4 b0 = x4[e / 4 + 4], b1 = x4[e / 4 + 5], b2 = x4[e / 4 + 6], b3 = x4[e / 4 + 7];
        KR_ADD16(s, a0, a1, a2, a3);
        KR_ADD16(s, b0, b1, b2, b3);
        e += 32;
    }
    for (; e < n; e++) s += x[e];
    return s;
}

// The reference's 8-lane sum of squares (lane l owns elements 8 b + l, b ascending; fused_add_rmsnorm_avx2, decode.rs:1235-1252) over
// a LANE-MAJOR copy xt[l * ld + b] = x[8 b + l] (ld % 4 == 0; ld = n / 8 + 4 keeps the 8 lanes on distinct banks): the chain lane reads
// its elements four at a time -- 64 ds_read_b128 instead of 256 ds_read_b32 for n = 2048, and a lone wave pays per instruction.
// Call with the first 8 lanes of a wave; returns the lane's partial (the caller folds the 8 lanes with the reference's hsum tree).
__device__ __forceinline__ float kr_sumsq_lane_t(const float* xt, int ld, int n, int l) {
    const float4* p = reinterpret_cast<const float4*>(xt + l * ld);
    const int nb = n / 8;
    float acc = 0.0f;
    int b = 0;
    for (; b + 32 <= nb; b += 32) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = p[(b >> 2) + u];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc = __builtin_fmaf(v[u].x, v[u].x, acc); acc = __builtin_fmaf(v[u].y, v[u].y, acc);
            acc = __builtin_fmaf(v[u].z, v[u].z, acc); acc = __builtin_fmaf(v[u].w, v[u].w, acc);
        }
    }
    for (; b < nb; b++) { const float v = xt[l * ld + b]; acc = __builtin_fmaf(v, v, acc); }
    return acc;
}

// ---------------------------------------------------------------------------------------------
// INT16 activation image: carving, and the f32 quantizer shared by every kernel that PRODUCES an image for a later matvec launch
// (the image can live in LDS or, pre-built by the producer of the activation, in global memory with the identical byte layout).
// ---------------------------------------------------------------------------------------------
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
__host__ __device__ static inline size_t kr_lds_bytes(int K, bool want_i8) {
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


// quantize_activation_int16_f32 (avx2.rs:274) of chunks [c0, c1) (8 values each; c0, c1 multiples of 16 = whole 128-groups), optionally after
// rounding to bf16 (decode.rs:3307-3309 feeds bf16(hidden) to the routed experts).  All threads of the workgroup take part (stride blockDim).
template <bool I8>
__device__ __forceinline__ void kr_quant_range_f32(const float* x, int c0, int c1, const KrActLds& L, bool round_bf16) {
    for (int c = c0 + (int)threadIdx.x; c < c1; c += (int)blockDim.x) {
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
