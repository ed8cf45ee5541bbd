/*
 * krasis_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See krasis_oracle.h for scope and rules.  Build: make -C oracle
 *
 * Compile flags matter: -ffp-contract=off (Rust never contracts a*b+c), fmaf() is
 * used exactly where the reference uses _mm256_fmadd_ps / _mm256_fnmadd_ps.
 * Rust `x.round()` = roundf (half away from zero); `_mm256_cvtps_epi32` = lrintf
 * under the default rounding mode (half to even); `as i32` saturates.
 */
#include "krasis_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* scalar helpers                                                      */
/* ------------------------------------------------------------------ */

static inline float bits_f32(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline uint32_t f32_bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

float kro_bf16_to_f32(uint16_t v) { return bits_f32((uint32_t)v << 16); }

uint16_t kro_f32_to_bf16(float v) {
    uint32_t bits = f32_bits(v);
    uint32_t round = bits + (0x7FFFu + ((bits >> 16) & 1u)); /* wrapping add */
    return (uint16_t)(round >> 16);
}

float kro_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    if (exp == 0) {
        if (man == 0) return bits_f32(sign);
        /* subnormal: man * 2^-24 */
        float f = (float)man * 5.9604644775390625e-8f;
        return (sign ? -f : f);
    }
    if (exp == 31) return bits_f32(sign | 0x7F800000u | (man << 13));
    return bits_f32(sign | ((exp + 112u) << 23) | (man << 13));
}

uint16_t kro_f32_to_f16(float v) {
    /* round-to-nearest-even, IEEE (matches VCVTPS2PH with _MM_FROUND_TO_NEAREST_INT) */
    __m128 x = _mm_set_ss(v);
    __m128i h = _mm_cvtps_ph(x, _MM_FROUND_TO_NEAREST_INT);
    return (uint16_t)_mm_extract_epi16(h, 0);
}

static inline int32_t sat_i32(float x) { /* Rust `as i32` */
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}
static inline int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float maxf_rust(float a, float b) { /* f32::max: NaN-ignoring */
    if (a != a) return b;
    if (b != b) return a;
    return a > b ? a : b;
}

/* horizontal sum of 8 virtual lanes, order of hsum_avx2 (avx2.rs:168, gguf_kernels.rs:666) */
static inline float hsum8(const float* l) {
    float s0 = l[0] + l[4], s1 = l[1] + l[5], s2 = l[2] + l[6], s3 = l[3] + l[7];
    float t0 = s0 + s1, t1 = s2 + s3;
    return t0 + t1;
}
static inline float hmax8(const float* l) { /* avx2.rs:182 */
    float m0 = fmaxf(l[0], l[4]), m1 = fmaxf(l[1], l[5]), m2 = fmaxf(l[2], l[6]), m3 = fmaxf(l[3], l[7]);
    float a = fmaxf(m0, m2), b = fmaxf(m1, m3);
    return fmaxf(a, b);
}

/* ------------------------------------------------------------------ */
/* A: codecs                                                           */
/* ------------------------------------------------------------------ */

size_t kro_ggml_block_size(int t) {
    switch (t) {
        case KRO_F32: case KRO_F16: case KRO_BF16: return 1;
        case KRO_Q4_0: case KRO_Q5_0: case KRO_Q8_0: return 32;
        case KRO_Q4_K: case KRO_Q5_K: case KRO_Q6_K: return 256;
        default: return 0;
    }
}
size_t kro_ggml_block_bytes(int t) {
    switch (t) {
        case KRO_F32: return 4;
        case KRO_F16: case KRO_BF16: return 2;
        case KRO_Q4_0: return 18; case KRO_Q5_0: return 22; case KRO_Q8_0: return 34;
        case KRO_Q4_K: return 144; case KRO_Q5_K: return 176; case KRO_Q6_K: return 210;
        default: return 0;
    }
}

void kro_get_scale_min_k4(int j, const uint8_t* s, uint8_t* sc, uint8_t* mn) {
    if (j < 4) { *sc = s[j] & 63; *mn = s[j + 4] & 63; }
    else {
        *sc = (uint8_t)((s[j + 4] & 0xF) | ((s[j - 4] >> 6) << 4));
        *mn = (uint8_t)((s[j + 4] >> 4) | ((s[j] >> 6) << 4));
    }
}

static inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

int kro_dequantize(int t, const uint8_t* data, size_t n, float* out) {
    switch (t) {
    case KRO_F32: memcpy(out, data, n * 4); return 0;
    case KRO_F16: for (size_t i = 0; i < n; i++) out[i] = kro_f16_to_f32(rd16(data + 2 * i)); return 0;
    case KRO_BF16: for (size_t i = 0; i < n; i++) out[i] = kro_bf16_to_f32(rd16(data + 2 * i)); return 0;
    case KRO_Q8_0: { /* gguf.rs:574 */
        size_t nb = n / 32;
        for (size_t i = 0; i < nb; i++) {
            const uint8_t* b = data + i * 34; float d = kro_f16_to_f32(rd16(b));
            for (int j = 0; j < 32; j++) out[i * 32 + j] = d * (float)(int8_t)b[2 + j];
        } return 0; }
    case KRO_Q5_0: { /* gguf.rs:599 */
        size_t nb = n / 32;
        for (size_t i = 0; i < nb; i++) {
            const uint8_t* b = data + i * 22; float d = kro_f16_to_f32(rd16(b));
            uint32_t qh = rd32(b + 2); const uint8_t* qs = b + 6;
            for (int j = 0; j < 32; j++) {
                uint8_t q4 = j < 16 ? (qs[j] & 0x0F) : ((qs[j - 16] >> 4) & 0x0F);
                uint8_t q5 = (uint8_t)((qh >> j) & 1);
                int q = (int)(q4 | (q5 << 4)) - 16;
                out[i * 32 + j] = d * (float)q;
            }
        } return 0; }
    case KRO_Q4_0: { /* gguf.rs:635 */
        size_t nb = n / 32;
        for (size_t i = 0; i < nb; i++) {
            const uint8_t* b = data + i * 18; float d = kro_f16_to_f32(rd16(b)); const uint8_t* qs = b + 2;
            for (int j = 0; j < 32; j++) {
                uint8_t nib = j < 16 ? (qs[j] & 0x0F) : ((qs[j - 16] >> 4) & 0x0F);
                out[i * 32 + j] = d * (float)((int)nib - 8);
            }
        } return 0; }
    case KRO_Q4_K: { /* gguf.rs:681 */
        size_t nb = n / 256;
        for (size_t i = 0; i < nb; i++) {
            const uint8_t* b = data + i * 144;
            float d = kro_f16_to_f32(rd16(b)), dmin = kro_f16_to_f32(rd16(b + 2));
            const uint8_t* sc = b + 4; const uint8_t* qs = b + 16;
            size_t base = i * 256; int is = 0; int qo = 0;
            for (int j = 0; j < 256; j += 64) {
                uint8_t s1, m1, s2, m2;
                kro_get_scale_min_k4(is, sc, &s1, &m1); kro_get_scale_min_k4(is + 1, sc, &s2, &m2);
                float d1 = d * (float)s1, mm1 = dmin * (float)m1, d2 = d * (float)s2, mm2 = dmin * (float)m2;
                for (int l = 0; l < 32; l++) out[base + j + l] = d1 * (float)(qs[qo + l] & 0xF) - mm1;
                for (int l = 0; l < 32; l++) out[base + j + 32 + l] = d2 * (float)(qs[qo + l] >> 4) - mm2;
                qo += 32; is += 2;
            }
        } return 0; }
    case KRO_Q5_K: { /* gguf.rs:740 */
        size_t nb = n / 256;
        for (size_t i = 0; i < nb; i++) {
            const uint8_t* b = data + i * 176;
            float d = kro_f16_to_f32(rd16(b)), dmin = kro_f16_to_f32(rd16(b + 2));
            const uint8_t* sc = b + 4; const uint8_t* qh = b + 16; const uint8_t* qs = b + 48;
            size_t base = i * 256; int is = 0; int qo = 0; uint8_t u1 = 1, u2 = 2;
            for (int j = 0; j < 256; j += 64) {
                uint8_t s1, m1, s2, m2;
                kro_get_scale_min_k4(is, sc, &s1, &m1); kro_get_scale_min_k4(is + 1, sc, &s2, &m2);
                float d1 = d * (float)s1, mm1 = dmin * (float)m1, d2 = d * (float)s2, mm2 = dmin * (float)m2;
                size_t ob = base + (size_t)is * 32;
                for (int l = 0; l < 32; l++) {
                    uint32_t q4 = qs[qo + l] & 0xF, q5 = (qh[l] & u1) ? 16u : 0u;
                    out[ob + l] = d1 * (float)(q4 + q5) - mm1;
                }
                for (int l = 0; l < 32; l++) {
                    uint32_t q4 = qs[qo + l] >> 4, q5 = (qh[l] & u2) ? 16u : 0u;
                    out[ob + 32 + l] = d2 * (float)(q4 + q5) - mm2;
                }
                qo += 32; u1 = (uint8_t)(u1 << 2); u2 = (uint8_t)(u2 << 2); is += 2;
            }
        } return 0; }
    case KRO_Q6_K: { /* gguf.rs:813 */
        size_t nb = n / 256;
        for (size_t i = 0; i < nb; i++) {
            const uint8_t* b = data + i * 210;
            const uint8_t* ql_all = b; const uint8_t* qh_all = b + 128; const uint8_t* sc_all = b + 192;
            float d = kro_f16_to_f32(rd16(b + 208));
            size_t base = i * 256; int oo = 0, qlo = 0, qho = 0, sco = 0;
            for (int h = 0; h < 2; h++) {
                const uint8_t* ql = ql_all + qlo; const uint8_t* qh = qh_all + qho; const uint8_t* sc = sc_all + sco;
                for (int l = 0; l < 32; l++) {
                    int8_t q1 = (int8_t)((ql[l] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                    int8_t q2 = (int8_t)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                    int8_t q3 = (int8_t)((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                    int8_t q4 = (int8_t)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                    /* NOTE: scale index sc[0],sc[2],sc[4],sc[6] for all l (gguf.rs:852-855) */
                    out[base + oo + l]      = d * (float)(int8_t)sc[0] * (float)q1;
                    out[base + oo + l + 32] = d * (float)(int8_t)sc[2] * (float)q2;
                    out[base + oo + l + 64] = d * (float)(int8_t)sc[4] * (float)q3;
                    out[base + oo + l + 96] = d * (float)(int8_t)sc[6] * (float)q4;
                }
                oo += 128; qlo += 64; qho += 32; sco += 8;
            }
        } return 0; }
    default: return -1;
    }
}

/* ------------------------------------------------------------------ */
/* A: weight quantizers + layouts                                      */
/* ------------------------------------------------------------------ */

void kro_quantize_int4(const uint16_t* w, int rows, int cols, int gs, uint32_t* packed, uint16_t* scales) {
    int ng = cols / gs, pc = cols / 8;
    for (int r = 0; r < rows; r++) {
        size_t ro = (size_t)r * cols;
        for (int g = 0; g < ng; g++) {
            float amax = 0.0f;
            for (int i = 0; i < gs; i++) amax = maxf_rust(amax, fabsf(kro_bf16_to_f32(w[ro + g * gs + i])));
            float scale = (amax == 0.0f) ? 1.0f : amax / 7.0f;
            scales[(size_t)r * ng + g] = kro_f32_to_bf16(scale);
        }
        for (int g = 0; g < ng; g++) {
            float scale = kro_bf16_to_f32(scales[(size_t)r * ng + g]);
            float inv = (scale == 0.0f) ? 0.0f : 1.0f / scale;
            for (int i = 0; i < gs; i += 8) {
                uint32_t word = 0;
                for (int j = 0; j < 8; j++) {
                    float val = kro_bf16_to_f32(w[ro + g * gs + i + j]);
                    float rq = roundf(val * inv);
                    if (rq < -8.0f) rq = -8.0f; if (rq > 7.0f) rq = 7.0f; /* f32::clamp; NaN stays NaN -> as i8 = 0 */
                    int8_t q = (rq != rq) ? 0 : (int8_t)rq;
                    uint8_t u4 = (uint8_t)(q + 8) & 0xF;
                    word |= (uint32_t)u4 << (j * 4);
                }
                packed[(size_t)r * pc + (g * gs + i) / 8] = word;
            }
        }
    }
}

void kro_quantize_int8(const uint16_t* w, int rows, int cols, int gs, int8_t* data, uint16_t* scales) {
    int ng = cols / gs;
    for (int r = 0; r < rows; r++) {
        size_t ro = (size_t)r * cols;
        for (int g = 0; g < ng; g++) {
            float amax = 0.0f;
            for (int i = 0; i < gs; i++) amax = maxf_rust(amax, fabsf(kro_bf16_to_f32(w[ro + g * gs + i])));
            float scale = (amax == 0.0f) ? 1.0f : amax / 127.0f;
            scales[(size_t)r * ng + g] = kro_f32_to_bf16(scale);
        }
        for (int g = 0; g < ng; g++) {
            float scale = kro_bf16_to_f32(scales[(size_t)r * ng + g]);
            float inv = (scale == 0.0f) ? 0.0f : 1.0f / scale;
            for (int i = 0; i < gs; i++) {
                float rq = roundf(kro_bf16_to_f32(w[ro + g * gs + i]) * inv);
                if (rq < -128.0f) rq = -128.0f; if (rq > 127.0f) rq = 127.0f;
                data[ro + g * gs + i] = (rq != rq) ? 0 : (int8_t)rq;
            }
        }
    }
}

void kro_dequantize_int4(const uint32_t* packed, const uint16_t* scales, int rows, int cols, int gs, float* out) {
    int ng = cols / gs, pc = cols / 8;
    for (int r = 0; r < rows; r++) for (int g = 0; g < ng; g++) {
        float s = kro_bf16_to_f32(scales[(size_t)r * ng + g]);
        for (int i = 0; i < gs; i += 8) {
            int c = g * gs + i; uint32_t word = packed[(size_t)r * pc + c / 8];
            for (int j = 0; j < 8; j++) out[(size_t)r * cols + c + j] = (float)((int)((word >> (j * 4)) & 0xF) - 8) * s;
        }
    }
}
void kro_dequantize_int8(const int8_t* data, const uint16_t* scales, int rows, int cols, int gs, float* out) {
    int ng = cols / gs;
    for (int r = 0; r < rows; r++) for (int g = 0; g < ng; g++) {
        float s = kro_bf16_to_f32(scales[(size_t)r * ng + g]);
        for (int i = 0; i < gs; i++) { size_t o = (size_t)r * cols + g * gs + i; out[o] = (float)data[o] * s; }
    }
}

void kro_transpose_int4(const uint32_t* packed, const uint16_t* scales, int n, int k, int gs, uint32_t* pt, uint16_t* st) {
    int pk = k / 8, ng = k / gs;
    for (int r = 0; r < n; r++) for (int c = 0; c < pk; c++) pt[(size_t)c * n + r] = packed[(size_t)r * pk + c];
    for (int r = 0; r < n; r++) for (int c = 0; c < ng; c++) st[(size_t)c * n + r] = scales[(size_t)r * ng + c];
}
void kro_transpose_int8(const int8_t* data, const uint16_t* scales, int n, int k, int gs, int8_t* dt, uint16_t* st) {
    int ng = k / gs;
    for (int r = 0; r < n; r++) for (int c = 0; c < k; c++) dt[(size_t)c * n + r] = data[(size_t)r * k + c];
    for (int r = 0; r < n; r++) for (int c = 0; c < ng; c++) st[(size_t)c * n + r] = scales[(size_t)r * ng + c];
}

void kro_quantize_f32_to_transposed_int4(const float* w, int rows, int cols, int gs, uint32_t* pt, uint16_t* st) {
    int pk = cols / 8, ng = cols / gs;
    for (int r = 0; r < rows; r++) {
        size_t rb = (size_t)r * cols;
        for (int g = 0; g < ng; g++) {
            float mx = 0.0f;
            for (int i = 0; i < gs; i++) mx = maxf_rust(mx, fabsf(w[rb + g * gs + i]));
            float scale = mx > 0.0f ? mx / 7.0f : 1.0f;
            float inv = mx > 0.0f ? 7.0f / mx : 0.0f;
            st[(size_t)g * rows + r] = kro_f32_to_bf16(scale);
            for (int p = 0; p < gs / 8; p++) {
                uint32_t word = 0;
                for (int j = 0; j < 8; j++) {
                    float val = w[rb + g * gs + p * 8 + j];
                    int32_t q = clampi(sat_i32(roundf(val * inv)), -8, 7);
                    word |= (uint32_t)(q + 8) << (j * 4);
                }
                pt[(size_t)(g * (gs / 8) + p) * rows + r] = word;
            }
        }
    }
    (void)pk;
}
void kro_quantize_f32_to_transposed_int8(const float* w, int rows, int cols, int gs, int8_t* dt, uint16_t* st) {
    int ng = cols / gs;
    for (int r = 0; r < rows; r++) {
        size_t rb = (size_t)r * cols;
        for (int g = 0; g < ng; g++) {
            float mx = 0.0f;
            for (int i = 0; i < gs; i++) mx = maxf_rust(mx, fabsf(w[rb + g * gs + i]));
            float scale = mx > 0.0f ? mx / 127.0f : 1.0f;
            float inv = mx > 0.0f ? 127.0f / mx : 0.0f;
            st[(size_t)g * rows + r] = kro_f32_to_bf16(scale);
            for (int i = 0; i < gs; i++) {
                int32_t q = clampi(sat_i32(roundf(w[rb + g * gs + i] * inv)), -128, 127);
                dt[(size_t)(g * gs + i) * rows + r] = (int8_t)q;
            }
        }
    }
}

#define KRO_TILE_N 256
size_t kro_repack_tiled_u32(const uint32_t* src, int k_rows, int n, uint32_t* dst) {
    int nt = (n + KRO_TILE_N - 1) / KRO_TILE_N; size_t total = (size_t)nt * k_rows * KRO_TILE_N;
    memset(dst, 0, total * 4);
    for (int t = 0; t < nt; t++) {
        int n0 = t * KRO_TILE_N, n1 = n0 + KRO_TILE_N < n ? n0 + KRO_TILE_N : n;
        for (int kr = 0; kr < k_rows; kr++)
            memcpy(dst + ((size_t)t * k_rows + kr) * KRO_TILE_N, src + (size_t)kr * n + n0, (size_t)(n1 - n0) * 4);
    }
    return total;
}
size_t kro_repack_tiled_u16(const uint16_t* src, int k_rows, int n, uint16_t* dst) {
    int nt = (n + KRO_TILE_N - 1) / KRO_TILE_N; size_t total = (size_t)nt * k_rows * KRO_TILE_N;
    memset(dst, 0, total * 2);
    for (int t = 0; t < nt; t++) {
        int n0 = t * KRO_TILE_N, n1 = n0 + KRO_TILE_N < n ? n0 + KRO_TILE_N : n;
        for (int kr = 0; kr < k_rows; kr++)
            memcpy(dst + ((size_t)t * k_rows + kr) * KRO_TILE_N, src + (size_t)kr * n + n0, (size_t)(n1 - n0) * 2);
    }
    return total;
}

/* ------------------------------------------------------------------ */
/* B: activation quantizers                                            */
/* ------------------------------------------------------------------ */

static void quant_group_round(const float* x, int n, int16_t* q, float* scale_out, int32_t* sum_out) {
    /* avx2.rs:246-266 / gguf_kernels.rs:153-170: scalar, `.round()` */
    float mx = 0.0f;
    for (int i = 0; i < n; i++) mx = maxf_rust(mx, fabsf(x[i]));
    float scale = mx > 0.0f ? mx / 32767.0f : 1.0f;
    float inv = mx > 0.0f ? 32767.0f / mx : 0.0f;
    *scale_out = scale;
    int32_t sum = 0;
    for (int i = 0; i < n; i++) {
        int32_t v = clampi(sat_i32(roundf(x[i] * inv)), -32768, 32767);
        q[i] = (int16_t)v; sum += v;
    }
    if (sum_out) *sum_out = sum;
}

void kro_quant_act_int16_f32(const float* x, int k, int gs, int16_t* q, float* scales) {
    for (int g = 0; g < k / gs; g++) quant_group_round(x + g * gs, gs, q + g * gs, scales + g, NULL);
}
void kro_quant_act_int16_bf16(const uint16_t* x, int k, int gs, int16_t* q, float* scales) {
    float* tmp = (float*)malloc(sizeof(float) * (size_t)k);
    for (int i = 0; i < k; i++) tmp[i] = kro_bf16_to_f32(x[i]);
    kro_quant_act_int16_f32(tmp, k, gs, q, scales);
    free(tmp);
}
void kro_gguf_quant_f32(const float* x, int k, int16_t* q, float* scales, int32_t* sums) {
    for (int g = 0; g < k / 32; g++) quant_group_round(x + g * 32, 32, q + g * 32, scales + g, sums + g);
}
void kro_gguf_quant_bf16(const uint16_t* x, int k, int16_t* q, float* scales, int32_t* sums) {
    float* tmp = (float*)malloc(sizeof(float) * (size_t)k);
    for (int i = 0; i < k; i++) tmp[i] = kro_bf16_to_f32(x[i]);
    kro_gguf_quant_f32(tmp, k, q, scales, sums);
    free(tmp);
}

/* fast_exp_avx2 (avx2.rs:2235): 2^(x*log2e), degree-5 poly, fma Horner */
static inline float fast_exp_poly5(float x) {
    const float log2e = 1.4426950408889634f;
    float t = x * log2e;
    float n = floorf(t);
    int32_t ni = (int32_t)lrintf(n);
    float f = t - n;
    float p = fmaf(0.0013333558f, f, 0.009618129f);
    p = fmaf(p, f, 0.0555041f);
    p = fmaf(p, f, 0.2402265f);
    p = fmaf(p, f, 0.6931472f);
    p = fmaf(p, f, 1.0f);
    float pow2n = bits_f32((uint32_t)(ni + 127) << 23);
    return p * pow2n;
}

float kro_sigmoid(float x, int mode) {
    if (mode == KRO_SIG_LIBM) return 1.0f / (1.0f + expf(-x));
    if (mode == KRO_SIG_POLY5_SCALAR) { /* moe.rs:1216 */
        float neg_x = -x; if (neg_x < -20.0f) neg_x = -20.0f; if (neg_x > 20.0f) neg_x = 20.0f;
        float t = neg_x * 1.4426950408889634f;
        float n = floorf(t); float f = t - n;
        float pow2f = 1.0f + f * (0.6931472f + f * (0.2402265f + f * (0.0555041f + f * (0.009618129f + f * 0.0013333558f))));
        float e = pow2f * bits_f32((uint32_t)(((int32_t)n + 127)) << 23);
        return 1.0f / (1.0f + e);
    }
    /* avx2.rs:2277: neg = 0 - x; clamp to [-20, 20] */
    float neg_x = 0.0f - x;
    float c = neg_x < 20.0f ? neg_x : 20.0f; /* _mm256_min_ps(neg_x, 20) */
    c = c > -20.0f ? c : -20.0f;             /* _mm256_max_ps(., -20) */
    float e = fast_exp_poly5(c);
    float denom = 1.0f + e;
    if (mode == KRO_SIG_POLY5_RCPNR) {
        float rcp = _mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(denom)));
        return rcp * fmaf(-denom, rcp, 2.0f); /* fnmadd(denom, rcp, two) */
    }
    return 1.0f / denom; /* KRO_SIG_POLY5_DIV */
}

void kro_silu_quant_int16(const float* gate, const float* up, int n, int gs, int sig_mode,
                          float* hidden_f32, int16_t* q, float* scales) {
    float* h = hidden_f32 ? hidden_f32 : (float*)malloc(sizeof(float) * (size_t)n);
    for (int g = 0; g < n / gs; g++) {
        int st = g * gs;
        float lanes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < gs; i++) {
            float gv = gate[st + i];
            float silu = gv * kro_sigmoid(gv, sig_mode);
            float hv = silu * up[st + i];
            h[st + i] = hv;
            float a = fabsf(hv);
            lanes[i & 7] = (lanes[i & 7] > a) ? lanes[i & 7] : a; /* _mm256_max_ps(max_abs_vec, abs_h) */
        }
        float mx = hmax8(lanes);
        float scale = mx > 0.0f ? mx / 32767.0f : 1.0f;
        float inv = mx > 0.0f ? 32767.0f / mx : 0.0f;
        scales[g] = scale;
        for (int i = 0; i < gs; i++) {
            float s = h[st + i] * inv;
            long v = lrintf(s); /* _mm256_cvtps_epi32: RNE */
            if (v > 32767) v = 32767; if (v < -32768) v = -32768; /* _mm_packs_epi32 saturation */
            q[st + i] = (int16_t)v;
        }
    }
    if (!hidden_f32) free(h);
}

void kro_fast_silu_mul(const float* gate, const float* up, int n, int sig_mode, float* out) {
    int n8 = n / 8;
    for (int i = 0; i < n8 * 8; i++) { float g = gate[i]; out[i] = (g * kro_sigmoid(g, sig_mode)) * up[i]; }
    for (int i = n8 * 8; i < n; i++) { float x = gate[i]; float s = 1.0f / (1.0f + expf(-x)); out[i] = x * s * up[i]; }
}

/* ------------------------------------------------------------------ */
/* C: matvecs                                                          */
/* ------------------------------------------------------------------ */

void kro_matvec_int4_t(const uint32_t* packed, const uint16_t* scales, const int16_t* a, const float* a_s,
                       int k, int n, int gs, float* out) {
    int ng = k / gs, ppg = gs / 8;
    int n_fma = (n / 8) * 8; /* tiles are multiples of 256, so only the global tail (<8 cols) takes the scalar path */
    for (int c = 0; c < n; c++) {
        float acc = 0.0f;
        for (int g = 0; g < ng; g++) {
            int32_t isum = 0;
            for (int p = 0; p < ppg; p++) {
                int kr = g * ppg + p; uint32_t word = packed[(size_t)kr * n + c];
                for (int j = 0; j < 8; j++) isum += ((int32_t)((word >> (j * 4)) & 0xF) - 8) * (int32_t)a[kr * 8 + j];
            }
            float combined = kro_bf16_to_f32(scales[(size_t)g * n + c]) * a_s[g];
            if (c < n_fma) acc = fmaf((float)isum, combined, acc);   /* avx2.rs:1171-1175 */
            else acc += (float)isum * combined;                      /* avx2.rs:1201 */
        }
        out[c] = acc;
    }
}

void kro_matvec_int8_t(const int8_t* data, const uint16_t* scales, const int16_t* a, const float* a_s,
                       int k, int n, int gs, float* out) {
    int ng = k / gs; int n_fma = (n / 8) * 8;
    for (int c = 0; c < n; c++) {
        float acc = 0.0f;
        for (int g = 0; g < ng; g++) {
            int32_t isum = 0;
            for (int i = 0; i < gs; i++) { int kp = g * gs + i; isum += (int32_t)data[(size_t)kp * n + c] * (int32_t)a[kp]; }
            float combined = kro_bf16_to_f32(scales[(size_t)g * n + c]) * a_s[g];
            if (c < n_fma) acc = fmaf((float)isum, combined, acc);
            else acc += (float)isum * combined;
        }
        out[c] = acc;
    }
}

void kro_matvec_int4_rowmajor(const uint32_t* packed, const uint16_t* scales, const int16_t* a, const float* a_s,
                              int k, int n, int gs, float* out) {
    int ng = k / gs, pk = k / 8;
    for (int r = 0; r < n; r++) {
        float acc = 0.0f;
        for (int g = 0; g < ng; g++) {
            float combined = kro_bf16_to_f32(scales[(size_t)r * ng + g]) * a_s[g];
            int32_t s = 0;
            for (int p = 0; p < gs / 8; p++) {
                int kb = g * gs + p * 8; uint32_t word = packed[(size_t)r * pk + kb / 8];
                for (int j = 0; j < 8; j++) s += ((int32_t)((word >> (j * 4)) & 0xF) - 8) * (int32_t)a[kb + j];
            }
            acc += (float)s * combined; /* avx2.rs:343 */
        }
        out[r] = acc;
    }
}

/* virtual-lane integer partials: lane l of _mm256_madd_epi16 over 16 i16 = w[2l]*a[2l] + w[2l+1]*a[2l+1] */
static void q4k_row(const uint8_t* row, int nblk, const int16_t* a, const float* a_s, const int32_t* a_sum, float* out) {
    float lanes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float corr = 0.0f;
    for (int b = 0; b < nblk; b++) {
        const uint8_t* blk = row + (size_t)b * 144;
        float d = kro_f16_to_f32(rd16(blk)), dmin = kro_f16_to_f32(rd16(blk + 2));
        const uint8_t* sp = blk + 4; const uint8_t* quants = blk + 16;
        int ab = b * 256, gb = b * 8;
        for (int j = 0; j < 4; j++) {
            uint8_t sc_lo, mn_lo, sc_hi, mn_hi;
            kro_get_scale_min_k4(2 * j, sp, &sc_lo, &mn_lo); kro_get_scale_min_k4(2 * j + 1, sp, &sc_hi, &mn_hi);
            const uint8_t* qs = quants + j * 32;
            int lo_g = gb + j * 2, hi_g = lo_g + 1;
            float as_lo = a_s[lo_g], as_hi = a_s[hi_g];
            int32_t ilo[8], ihi[8];
            for (int l = 0; l < 8; l++) {
                const int16_t* al = a + ab + j * 64;
                ilo[l] = (int32_t)(qs[2 * l] & 0xF) * al[2 * l] + (int32_t)(qs[2 * l + 1] & 0xF) * al[2 * l + 1]
                       + (int32_t)(qs[16 + 2 * l] & 0xF) * al[16 + 2 * l] + (int32_t)(qs[16 + 2 * l + 1] & 0xF) * al[16 + 2 * l + 1];
                const int16_t* ah = a + ab + j * 64 + 32;
                ihi[l] = (int32_t)(qs[2 * l] >> 4) * ah[2 * l] + (int32_t)(qs[2 * l + 1] >> 4) * ah[2 * l + 1]
                       + (int32_t)(qs[16 + 2 * l] >> 4) * ah[16 + 2 * l] + (int32_t)(qs[16 + 2 * l + 1] >> 4) * ah[16 + 2 * l + 1];
            }
            float comb_lo = d * (float)sc_lo * as_lo;
            for (int l = 0; l < 8; l++) lanes[l] = fmaf((float)ilo[l], comb_lo, lanes[l]);
            corr += dmin * (float)mn_lo * as_lo * (float)a_sum[lo_g];
            float comb_hi = d * (float)sc_hi * as_hi;
            for (int l = 0; l < 8; l++) lanes[l] = fmaf((float)ihi[l], comb_hi, lanes[l]);
            corr += dmin * (float)mn_hi * as_hi * (float)a_sum[hi_g];
        }
    }
    *out = hsum8(lanes) - corr;
}

static void q8_0_row(const uint8_t* row, int nblk, const int16_t* a, const float* a_s, float* out) {
    float lanes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < nblk; b++) {
        const uint8_t* blk = row + (size_t)b * 34; float d = kro_f16_to_f32(rd16(blk));
        const int8_t* qs = (const int8_t*)(blk + 2);
        float comb = d * a_s[b];
        const int16_t* ab = a + b * 32;
        for (int l = 0; l < 8; l++) {
            int32_t v = (int32_t)qs[2 * l] * ab[2 * l] + (int32_t)qs[2 * l + 1] * ab[2 * l + 1]
                      + (int32_t)qs[16 + 2 * l] * ab[16 + 2 * l] + (int32_t)qs[16 + 2 * l + 1] * ab[16 + 2 * l + 1];
            lanes[l] = fmaf((float)v, comb, lanes[l]);
        }
    }
    *out = hsum8(lanes);
}

static void q4_0_row(const uint8_t* row, int nblk, const int16_t* a, const float* a_s, const int32_t* a_sum, float* out) {
    float lanes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float corr = 0.0f;
    for (int b = 0; b < nblk; b++) {
        const uint8_t* blk = row + (size_t)b * 18; float d = kro_f16_to_f32(rd16(blk));
        const uint8_t* qs = blk + 2; const int16_t* ab = a + b * 32;
        for (int l = 0; l < 8; l++) {
            int32_t v = (int32_t)(qs[2 * l] & 0xF) * ab[2 * l] + (int32_t)(qs[2 * l + 1] & 0xF) * ab[2 * l + 1]
                      + (int32_t)(qs[2 * l] >> 4) * ab[16 + 2 * l] + (int32_t)(qs[2 * l + 1] >> 4) * ab[16 + 2 * l + 1];
            lanes[l] = fmaf((float)v, d * a_s[b], lanes[l]);
        }
        corr += d * 8.0f * a_s[b] * (float)a_sum[b];
    }
    *out = hsum8(lanes) - corr;
}

void kro_gguf_matvec_f32(int t, const uint8_t* w, const float* x, int n, int k, float* out) {
    size_t bs = kro_ggml_block_size(t), bb = kro_ggml_block_bytes(t);
    size_t nblk = (size_t)k / bs, row_bytes = nblk * bb;
    for (int r = 0; r < n; r++) {
        const uint8_t* row = w + (size_t)r * row_bytes; float sum = 0.0f;
        switch (t) {
        case KRO_Q4_K: /* gguf_kernels.rs:495-532 */
            for (size_t b = 0; b < nblk; b++) {
                const uint8_t* blk = row + b * 144; const float* in = x + b * 256;
                float d = kro_f16_to_f32(rd16(blk)), dmin = kro_f16_to_f32(rd16(blk + 2));
                float bsum = 0.0f;
                for (int j = 0; j < 4; j++) {
                    uint8_t sl, ml, sh, mh; kro_get_scale_min_k4(2 * j, blk + 4, &sl, &ml); kro_get_scale_min_k4(2 * j + 1, blk + 4, &sh, &mh);
                    float d_lo = d * (float)sl, min_lo = dmin * (float)ml, d_hi = d * (float)sh, min_hi = dmin * (float)mh;
                    const uint8_t* qs = blk + 16 + j * 32;
                    for (int l = 0; l < 32; l++) bsum += (d_lo * (float)(qs[l] & 0xF) - min_lo) * in[j * 64 + l];
                    for (int l = 0; l < 32; l++) bsum += (d_hi * (float)((qs[l] >> 4) & 0xF) - min_hi) * in[j * 64 + 32 + l];
                }
                sum += bsum;
            } break;
        case KRO_Q8_0:
            for (size_t b = 0; b < nblk; b++) {
                const uint8_t* blk = row + b * 34; float d = kro_f16_to_f32(rd16(blk));
                for (int j = 0; j < 32; j++) sum += d * (float)(int8_t)blk[2 + j] * x[b * 32 + j];
            } break;
        case KRO_Q4_0:
            for (size_t b = 0; b < nblk; b++) {
                const uint8_t* blk = row + b * 18; float d = kro_f16_to_f32(rd16(blk)); const uint8_t* qs = blk + 2;
                for (int j = 0; j < 32; j++) {
                    uint8_t nib = j < 16 ? (qs[j] & 0x0F) : ((qs[j - 16] >> 4) & 0x0F);
                    sum += d * (float)((int)nib - 8) * x[b * 32 + j];
                }
            } break;
        case KRO_Q5_0:
            for (size_t b = 0; b < nblk; b++) {
                const uint8_t* blk = row + b * 22; float d = kro_f16_to_f32(rd16(blk)); uint32_t qh = rd32(blk + 2); const uint8_t* qs = blk + 6;
                for (int j = 0; j < 32; j++) {
                    uint8_t q4 = j < 16 ? (qs[j] & 0x0F) : ((qs[j - 16] >> 4) & 0x0F);
                    int q = (int)(q4 | (((qh >> j) & 1) << 4)) - 16;
                    sum += d * (float)q * x[b * 32 + j];
                }
            } break;
        case KRO_Q6_K: /* gguf_kernels.rs:594 -- scale index is + l/16, unlike gguf.rs dequant */
            for (size_t b = 0; b < nblk; b++) {
                const uint8_t* blk = row + b * 210; const uint8_t* ql = blk; const uint8_t* qh = blk + 128; const uint8_t* sc = blk + 192;
                float d = kro_f16_to_f32(rd16(blk + 208)); const float* in = x + b * 256;
                for (int h = 0; h < 2; h++) {
                    int qlo = h * 64, qho = h * 32, sco = h * 8, io = h * 128;
                    for (int l = 0; l < 32; l++) {
                        int is = l / 16;
                        int q0 = (int)(ql[qlo + l] & 0xF) | ((int)((qh[qho + l] >> 0) & 3) << 4);
                        int q1 = (int)(ql[qlo + 32 + l] & 0xF) | ((int)((qh[qho + l] >> 2) & 3) << 4);
                        int q2 = (int)((ql[qlo + l] >> 4) & 0xF) | ((int)((qh[qho + l] >> 4) & 3) << 4);
                        int q3 = (int)((ql[qlo + 32 + l] >> 4) & 0xF) | ((int)((qh[qho + l] >> 6) & 3) << 4);
                        float s0 = d * (float)(int8_t)sc[sco + is + 0], s1 = d * (float)(int8_t)sc[sco + is + 2];
                        float s2 = d * (float)(int8_t)sc[sco + is + 4], s3 = d * (float)(int8_t)sc[sco + is + 6];
                        sum += s0 * (float)(q0 - 32) * in[io + l];
                        sum += s1 * (float)(q1 - 32) * in[io + 32 + l];
                        sum += s2 * (float)(q2 - 32) * in[io + 64 + l];
                        sum += s3 * (float)(q3 - 32) * in[io + 96 + l];
                    }
                }
            } break;
        default: break;
        }
        out[r] = sum;
    }
}

static int gguf_int_path(int t) { return t == KRO_Q4_K || t == KRO_Q8_0 || t == KRO_Q4_0; } /* gguf_kernels.rs:99 */

/* The same two rows with the host's AVX2 units (matvec_q4_k_avx2 / matvec_q8_0_avx2, gguf_kernels.rs:271-432): _mm256_madd_epi16 over
 * 16 widened weights x 16 INT16 activations leaves lane l = w[2l] a[2l] + w[2l+1] a[2l+1]; the two halves of a 32-value sub-block are added
 * as integers, converted, and enter the 8 f32 lane accumulators by one fused multiply-add with d * sc * a_scale; the min correction is a
 * scalar chain.  Bit-identical to q4k_row / q8_0_row above (tests/test_oracle_avx2.py) -- those are this code with the lanes spelled out. */
static void q4k_row_avx2(const uint8_t* row, int nblk, const int16_t* a, const float* a_s, const int32_t* a_sum, float* out) {
    __m256 acc = _mm256_setzero_ps();
    const __m128i m4 = _mm_set1_epi8(0x0F);
    float corr = 0.0f;
    for (int b = 0; b < nblk; b++) {
        const uint8_t* blk = row + (size_t)b * 144;
        float d = kro_f16_to_f32(rd16(blk)), dmin = kro_f16_to_f32(rd16(blk + 2));
        const uint8_t* sp = blk + 4; const uint8_t* quants = blk + 16;
        const int16_t* ab = a + (size_t)b * 256; int gb = b * 8;
        for (int j = 0; j < 4; j++) {
            uint8_t sc_lo, mn_lo, sc_hi, mn_hi;
            kro_get_scale_min_k4(2 * j, sp, &sc_lo, &mn_lo); kro_get_scale_min_k4(2 * j + 1, sp, &sc_hi, &mn_hi);
            const __m128i q0 = _mm_loadu_si128((const __m128i*)(quants + j * 32)), q1 = _mm_loadu_si128((const __m128i*)(quants + j * 32 + 16));
            int lo_g = gb + j * 2, hi_g = lo_g + 1;
            float as_lo = a_s[lo_g], as_hi = a_s[hi_g];
            const int16_t* al = ab + j * 64;
            __m256i ilo = _mm256_add_epi32(_mm256_madd_epi16(_mm256_cvtepu8_epi16(_mm_and_si128(q0, m4)), _mm256_loadu_si256((const __m256i*)al)),
                                           _mm256_madd_epi16(_mm256_cvtepu8_epi16(_mm_and_si128(q1, m4)), _mm256_loadu_si256((const __m256i*)(al + 16))));
            __m256i ihi = _mm256_add_epi32(_mm256_madd_epi16(_mm256_cvtepu8_epi16(_mm_and_si128(_mm_srli_epi16(q0, 4), m4)), _mm256_loadu_si256((const __m256i*)(al + 32))),
                                           _mm256_madd_epi16(_mm256_cvtepu8_epi16(_mm_and_si128(_mm_srli_epi16(q1, 4), m4)), _mm256_loadu_si256((const __m256i*)(al + 48))));
            float comb_lo = d * (float)sc_lo * as_lo;
            acc = _mm256_fmadd_ps(_mm256_cvtepi32_ps(ilo), _mm256_set1_ps(comb_lo), acc);
            corr += dmin * (float)mn_lo * as_lo * (float)a_sum[lo_g];
            float comb_hi = d * (float)sc_hi * as_hi;
            acc = _mm256_fmadd_ps(_mm256_cvtepi32_ps(ihi), _mm256_set1_ps(comb_hi), acc);
            corr += dmin * (float)mn_hi * as_hi * (float)a_sum[hi_g];
        }
    }
    float lanes[8]; _mm256_storeu_ps(lanes, acc);
    *out = hsum8(lanes) - corr;
}
static void q8_0_row_avx2(const uint8_t* row, int nblk, const int16_t* a, const float* a_s, float* out) {
    __m256 acc = _mm256_setzero_ps();
    for (int b = 0; b < nblk; b++) {
        const uint8_t* blk = row + (size_t)b * 34; float d = kro_f16_to_f32(rd16(blk));
        float comb = d * a_s[b];
        const int16_t* ab = a + (size_t)b * 32;
        __m256i v = _mm256_add_epi32(_mm256_madd_epi16(_mm256_cvtepi8_epi16(_mm_loadu_si128((const __m128i*)(blk + 2))), _mm256_loadu_si256((const __m256i*)ab)),
                                     _mm256_madd_epi16(_mm256_cvtepi8_epi16(_mm_loadu_si128((const __m128i*)(blk + 18))), _mm256_loadu_si256((const __m256i*)(ab + 16))));
        acc = _mm256_fmadd_ps(_mm256_cvtepi32_ps(v), _mm256_set1_ps(comb), acc);
    }
    float lanes[8]; _mm256_storeu_ps(lanes, acc);
    *out = hsum8(lanes);
}
static int g_gguf_avx2 = 0;   /* kro_gguf_set_avx2: rows through the AVX2 forms, output rows split over the OpenMP team (rayon in the reference) */
void kro_gguf_set_avx2(int on) { g_gguf_avx2 = on; }

void kro_gguf_matvec_int(int t, const uint8_t* w, const int16_t* a, const float* a_s, const int32_t* a_sum,
                         int n, int k, float* out) {
    size_t bs = kro_ggml_block_size(t), bb = kro_ggml_block_bytes(t);
    int nblk = (int)((size_t)k / bs); size_t row_bytes = (size_t)nblk * bb;
    if (!gguf_int_path(t)) { /* gguf_kernels.rs:226-235 */
        float* xf = (float*)malloc(sizeof(float) * (size_t)k);
        for (int g = 0; g < k / 32; g++) for (int i = 0; i < 32; i++) xf[g * 32 + i] = (float)a[g * 32 + i] * a_s[g];
        kro_gguf_matvec_f32(t, w, xf, n, k, out);
        free(xf); return;
    }
    if (g_gguf_avx2 && t != KRO_Q4_0) {
#pragma omp parallel for schedule(static) if (n >= 256)
        for (int r = 0; r < n; r++) {
            const uint8_t* row = w + (size_t)r * row_bytes;
            if (t == KRO_Q4_K) q4k_row_avx2(row, nblk, a, a_s, a_sum, out + r); else q8_0_row_avx2(row, nblk, a, a_s, out + r);
        }
        return;
    }
    for (int r = 0; r < n; r++) {
        const uint8_t* row = w + (size_t)r * row_bytes;
        if (t == KRO_Q4_K) q4k_row(row, nblk, a, a_s, a_sum, out + r);
        else if (t == KRO_Q8_0) q8_0_row(row, nblk, a, a_s, out + r);
        else q4_0_row(row, nblk, a, a_s, a_sum, out + r);
    }
}

/* ------------------------------------------------------------------ */
/* D/E: expert + MoE forward                                           */
/* ------------------------------------------------------------------ */

static void unified_matvec(const void* w, const uint16_t* s, int bits, const int16_t* a, const float* a_s,
                           int k, int n, int gs, float* out) {
    if (bits == 4) kro_matvec_int4_t((const uint32_t*)w, s, a, a_s, k, n, gs, out);
    else kro_matvec_int8_t((const int8_t*)w, s, a, a_s, k, n, gs, out);
}

void kro_expert_forward_unified(const kro_unified_expert* e, const int16_t* a, const float* a_s,
                                float swiglu_limit, float alpha, int sig_mode, float* out) {
    int k = e->hidden, n = e->inter, gs = e->gs;
    float* w13 = (float*)malloc(sizeof(float) * 2 * (size_t)n);
    int16_t* hq = (int16_t*)malloc(sizeof(int16_t) * (size_t)n);
    float* hs = (float*)malloc(sizeof(float) * (size_t)(n / gs + 1));
    unified_matvec(e->w13, e->w13_scales, e->num_bits, a, a_s, k, 2 * n, gs, w13);
    if (e->gate_bias) for (int i = 0; i < n; i++) w13[i] += e->gate_bias[i];
    if (e->up_bias) for (int i = 0; i < n; i++) w13[n + i] += e->up_bias[i];
    if (swiglu_limit > 0.0f) { /* moe.rs:268-287 */
        for (int i = 0; i < n; i++) {
            float gate = w13[i], up = w13[n + i];
            if (gate > swiglu_limit) gate = swiglu_limit;
            if (up > swiglu_limit) up = swiglu_limit;
            if (up < -swiglu_limit) up = -swiglu_limit;
            float glu = gate * kro_sigmoid(gate * alpha, KRO_SIG_POLY5_SCALAR);
            w13[i] = (up + 1.0f) * glu;
        }
        kro_quant_act_int16_f32(w13, n, gs, hq, hs);
    } else if (n % 8 == 0 && gs % 8 == 0) {
        kro_silu_quant_int16(w13, w13 + n, n, gs, sig_mode, NULL, hq, hs);
    } else {
        for (int i = 0; i < n; i++) { float g = w13[i]; w13[i] = g * kro_sigmoid(g, KRO_SIG_POLY5_SCALAR) * w13[n + i]; }
        kro_quant_act_int16_f32(w13, n, gs, hq, hs);
    }
    unified_matvec(e->w2, e->w2_scales, e->w2_bits, hq, hs, n, k, gs, out);
    if (e->down_bias) for (int j = 0; j < k; j++) out[j] += e->down_bias[j];
    free(w13); free(hq); free(hs);
}

void kro_expert_forward_gguf(const kro_gguf_expert* e, const uint16_t* act, float* out) {
    int k = e->hidden, n = e->inter;
    int int_gu = gguf_int_path(e->gate_up_type) && (k % 32 == 0);
    int int_dn = gguf_int_path(e->down_type) && (n % 32 == 0);
    float* gate = (float*)malloc(sizeof(float) * (size_t)n);
    float* up = (float*)malloc(sizeof(float) * (size_t)n);
    float* hid = (float*)malloc(sizeof(float) * (size_t)n);
    if (int_gu) {
        int16_t* q = (int16_t*)malloc(2 * (size_t)k); float* s = (float*)malloc(4 * (size_t)(k / 32)); int32_t* sm = (int32_t*)malloc(4 * (size_t)(k / 32));
        kro_gguf_quant_bf16(act, k, q, s, sm);
        kro_gguf_matvec_int(e->gate_up_type, e->gate, q, s, sm, n, k, gate);
        kro_gguf_matvec_int(e->gate_up_type, e->up, q, s, sm, n, k, up);
        free(q); free(s); free(sm);
    } else {
        float* xf = (float*)malloc(4 * (size_t)k);
        for (int i = 0; i < k; i++) xf[i] = kro_bf16_to_f32(act[i]);
        kro_gguf_matvec_f32(e->gate_up_type, e->gate, xf, n, k, gate);
        kro_gguf_matvec_f32(e->gate_up_type, e->up, xf, n, k, up);
        free(xf);
    }
    for (int i = 0; i < n; i++) { float g = gate[i]; float silu = g / (1.0f + expf(-g)); hid[i] = silu * up[i]; }
    if (int_dn) {
        int16_t* q = (int16_t*)malloc(2 * (size_t)n); float* s = (float*)malloc(4 * (size_t)(n / 32)); int32_t* sm = (int32_t*)malloc(4 * (size_t)(n / 32));
        kro_gguf_quant_f32(hid, n, q, s, sm);
        kro_gguf_matvec_int(e->down_type, e->down, q, s, sm, k, n, out);
        free(q); free(s); free(sm);
    } else {
        kro_gguf_matvec_f32(e->down_type, e->down, hid, k, n, out);
    }
    free(gate); free(up); free(hid);
}

void kro_moe_forward_unified(const kro_unified_expert* const* experts, const float* weights, int n_sel,
                             const kro_unified_expert* shared, float rsf,
                             const uint16_t* act, float swiglu_limit, float alpha, int sig_mode, float* out) {
    if (n_sel == 0 && !shared) return;
    const kro_unified_expert* e0 = n_sel > 0 ? experts[0] : shared;
    int hidden = e0->hidden, gs = e0->gs;
    int16_t* q = (int16_t*)malloc(2 * (size_t)hidden); float* s = (float*)malloc(4 * (size_t)(hidden / gs));
    float* eo = (float*)malloc(4 * (size_t)hidden);
    kro_quant_act_int16_bf16(act, hidden, gs, q, s);
    for (int j = 0; j < hidden; j++) out[j] = 0.0f;
    for (int i = 0; i < n_sel; i++) {
        kro_expert_forward_unified(experts[i], q, s, swiglu_limit, alpha, sig_mode, eo);
        float w = weights[i];
        for (int j = 0; j < hidden; j++) out[j] += w * eo[j]; /* moe.rs:661-667 */
    }
    if (shared) {
        kro_expert_forward_unified(shared, q, s, swiglu_limit, alpha, sig_mode, eo);
        for (int j = 0; j < hidden; j++) out[j] = rsf * out[j] + eo[j]; /* moe.rs:703-706 */
    }
    free(q); free(s); free(eo);
}

void kro_moe_forward_gguf(const kro_gguf_expert* const* experts, const float* weights, int n_sel,
                          const kro_gguf_expert* shared, float rsf, const uint16_t* act, float* out) {
    if (n_sel == 0 && !shared) return;
    int hidden = (n_sel > 0 ? experts[0] : shared)->hidden;
    float* eo = (float*)malloc(4 * (size_t)hidden);
    for (int j = 0; j < hidden; j++) out[j] = 0.0f;
    for (int i = 0; i < n_sel; i++) {
        kro_expert_forward_gguf(experts[i], act, eo);
        float w = weights[i];
        for (int j = 0; j < hidden; j++) out[j] += w * eo[j];
    }
    if (shared) {
        kro_expert_forward_gguf(shared, act, eo);
        for (int j = 0; j < hidden; j++) out[j] = rsf * out[j] + eo[j];
    }
    free(eo);
}

/* ------------------------------------------------------------------ */
/* F: routers                                                          */
/* ------------------------------------------------------------------ */

typedef struct { float v; int i; } hv;
static void stable_sort_asc(hv* h, int k) { /* any stable sort == Rust's sort_by (stable) */
    for (int i = 1; i < k; i++) { hv x = h[i]; int j = i - 1; while (j >= 0 && h[j].v > x.v) { h[j + 1] = h[j]; j--; } h[j + 1] = x; }
}
static void stable_sort_desc(hv* h, int k) {
    for (int i = 1; i < k; i++) { hv x = h[i]; int j = i - 1; while (j >= 0 && h[j].v < x.v) { h[j + 1] = h[j]; j--; } h[j + 1] = x; }
}

void kro_topk_indices(const float* values, int n, int k, int32_t* out) {
    hv* heap = (hv*)malloc(sizeof(hv) * (size_t)k);
    for (int i = 0; i < k; i++) { heap[i].v = values[i]; heap[i].i = i; }
    stable_sort_asc(heap, k);
    for (int i = k; i < n; i++) {
        if (values[i] > heap[0].v) {
            heap[0].v = values[i]; heap[0].i = i;
            int pos = 0;
            for (;;) {
                int left = 2 * pos + 1, right = 2 * pos + 2, smallest = pos;
                if (left < k && heap[left].v < heap[smallest].v) smallest = left;
                if (right < k && heap[right].v < heap[smallest].v) smallest = right;
                if (smallest == pos) break;
                hv t = heap[pos]; heap[pos] = heap[smallest]; heap[smallest] = t;
                pos = smallest;
            }
        }
    }
    stable_sort_desc(heap, k);
    for (int i = 0; i < k; i++) out[i] = heap[i].i;
    free(heap);
}

void kro_route_matmul(const float* gate, const float* hidden, int ne, int hd, float* logits) {
    int chunks = hd / 8, chunks2 = chunks / 2;
    for (int e = 0; e < ne; e++) {
        const float* row = gate + (size_t)e * hd;
        float a0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int i = 0;
        for (int c = 0; c < chunks2; c++) {
            for (int l = 0; l < 8; l++) a0[l] = fmaf(row[i + l], hidden[i + l], a0[l]);
            for (int l = 0; l < 8; l++) a1[l] = fmaf(row[i + 8 + l], hidden[i + 8 + l], a1[l]);
            i += 16;
        }
        if (chunks % 2) for (int l = 0; l < 8; l++) a0[l] = fmaf(row[i + l], hidden[i + l], a0[l]);
        float s8[8]; for (int l = 0; l < 8; l++) s8[l] = a0[l] + a1[l];
        logits[e] = hsum8(s8);
    }
}

static inline float sigmoid_poly4(float x) { /* decode.rs:4110-4131 */
    float neg_x = 0.0f - x;
    float t = neg_x * 1.4426950408889634f;
    float n = floorf(t); int32_t ni = (int32_t)lrintf(n); float f = t - n;
    float p = fmaf(fmaf(fmaf(fmaf(0.009518f, f, 0.0558011f), f, 0.2402265f), f, 0.6931472f), f, 1.0f);
    float pow2n = bits_f32((uint32_t)(ni + 127) << 23);
    float e = p * pow2n;
    return 1.0f / (1.0f + e);
}

void kro_route_score_topk(float* logits, int ne, const float* esc, int scoring, int norm_topk, int topk,
                          float* scores, int32_t* ids, float* w) {
    float* corrected = (float*)malloc(4 * (size_t)ne);
    if (scoring == 0 || scoring == 1) {
        if (scoring == 0) {
            int ne8 = ne / 8;
            for (int e = 0; e < ne8 * 8; e++) scores[e] = sigmoid_poly4(logits[e]);
            for (int e = ne8 * 8; e < ne; e++) scores[e] = 1.0f / (1.0f + expf(-logits[e]));
        } else {
            float mx = -INFINITY; for (int e = 0; e < ne; e++) mx = maxf_rust(mx, logits[e]);
            float se = 0.0f; for (int e = 0; e < ne; e++) { scores[e] = expf(logits[e] - mx); se += scores[e]; }
            float inv = 1.0f / se; for (int e = 0; e < ne; e++) scores[e] *= inv;
        }
        if (esc) { for (int e = 0; e < ne; e++) corrected[e] = scores[e] + esc[e]; kro_topk_indices(corrected, ne, topk, ids); }
        else kro_topk_indices(scores, ne, topk, ids);
        for (int i = 0; i < topk; i++) w[i] = scores[ids[i]];
        if (norm_topk) {
            float sum = 0.0f; for (int i = 0; i < topk; i++) sum += w[i]; /* iter().sum(): sequential from 0.0 */
            if (sum > 0.0f) for (int i = 0; i < topk; i++) w[i] /= sum;
        }
    } else if (scoring == 2) {
        kro_topk_indices(logits, ne, topk, ids);
        float mx = -INFINITY; for (int i = 0; i < topk; i++) mx = maxf_rust(mx, logits[ids[i]]);
        float se = 0.0f; for (int i = 0; i < topk; i++) { float v = expf(logits[ids[i]] - mx); w[i] = v; se += v; }
        float inv = 1.0f / se; for (int i = 0; i < topk; i++) w[i] *= inv;
    }
    free(corrected);
}

void kro_route_engine(const uint16_t* gate, const uint16_t* act, int ne, int hd, const float* bias,
                      int sigmoid, int norm_topk, int topk, float swiglu_limit, int32_t* ids, float* w) {
    float* logits = (float*)malloc(4 * (size_t)ne); float* scores = (float*)malloc(4 * (size_t)ne);
    float* sel = (float*)malloc(4 * (size_t)ne); char* used = (char*)calloc((size_t)ne, 1);
    for (int e = 0; e < ne; e++) {
        const uint16_t* row = gate + (size_t)e * hd; float sum = 0.0f;
        for (int j = 0; j < hd; j++) sum += kro_bf16_to_f32(act[j]) * kro_bf16_to_f32(row[j]);
        logits[e] = sum;
    }
    if (swiglu_limit > 0.0f) { /* moe.rs:3101-3141 */
        if (bias) for (int e = 0; e < ne; e++) logits[e] += bias[e];
        float mx = -INFINITY;
        for (int t = 0; t < topk; t++) {
            int best = 0; float bv = -INFINITY;
            for (int e = 0; e < ne; e++) if (!used[e] && logits[e] > bv) { bv = logits[e]; best = e; }
            used[best] = 1; ids[t] = best; w[t] = logits[best]; mx = maxf_rust(mx, logits[best]);
        }
        float se = 0.0f; for (int t = 0; t < topk; t++) { w[t] = expf(w[t] - mx); se += w[t]; }
        for (int t = 0; t < topk; t++) w[t] /= se;
    } else {
        if (sigmoid) for (int e = 0; e < ne; e++) scores[e] = 1.0f / (1.0f + expf(-logits[e]));
        else {
            float mx = -INFINITY; for (int e = 0; e < ne; e++) mx = maxf_rust(mx, logits[e]);
            float se = 0.0f; for (int e = 0; e < ne; e++) { scores[e] = expf(logits[e] - mx); se += scores[e]; }
            for (int e = 0; e < ne; e++) scores[e] /= se;
        }
        for (int e = 0; e < ne; e++) sel[e] = scores[e] + (bias ? bias[e] : 0.0f);
        if (!bias) memcpy(sel, scores, 4 * (size_t)ne);
        for (int t = 0; t < topk; t++) {
            int best = 0; float bs = -INFINITY;
            for (int e = 0; e < ne; e++) if (!used[e] && sel[e] > bs) { bs = sel[e]; best = e; }
            used[best] = 1; ids[t] = best; w[t] = scores[best];
        }
        if (norm_topk) {
            float sum = 0.0f; for (int t = 0; t < topk; t++) sum += w[t];
            if (sum > 0.0f) for (int t = 0; t < topk; t++) w[t] /= sum;
        }
    }
    free(logits); free(scores); free(sel); free(used);
}

/* ------------------------------------------------------------------ */
/* G: decode-graph ops                                                 */
/* ------------------------------------------------------------------ */

/* sum of squares with 8 fma lanes + hsum (decode.rs:1235-1252 and twins) */
static float sumsq_lanes(const float* x, int n) {
    int n8 = n / 8; float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < n8; b++) for (int j = 0; j < 8; j++) l[j] = fmaf(x[b * 8 + j], x[b * 8 + j], l[j]);
    float s = hsum8(l);
    for (int r = n8 * 8; r < n; r++) s += x[r] * x[r];
    return s;
}

void kro_fused_add_rmsnorm(float* hidden, float* residual, const float* w, int n, float eps, int first, int bias_one) {
    if (first) memcpy(residual, hidden, 4 * (size_t)n);
    else for (int i = 0; i < n; i++) residual[i] = hidden[i] + residual[i];
    float ss = sumsq_lanes(residual, n);
    float rms = 1.0f / sqrtf(ss / (float)n + eps); /* .sqrt().recip() */
    if (bias_one) for (int i = 0; i < n; i++) hidden[i] = (residual[i] * rms) * (w[i] + 1.0f);
    else for (int i = 0; i < n; i++) hidden[i] = (residual[i] * rms) * w[i];
    /* scalar tail uses residual*rms*(1.0+w): identical values (a+1 == 1+a) */
}

static void l2norm_expand(const float* src, float* dst, int src_off, int dk, int nv, int hr, float scale) { /* decode.rs:3909 */
    for (int vh = 0; vh < nv; vh++) {
        int kh = vh / hr; const float* s = src + src_off + kh * dk; float* d = dst + vh * dk;
        float ss = sumsq_lanes(s, dk);
        float inv_norm = ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f;
        float inv_v = inv_norm * scale;
        int dk8 = dk / 8;
        for (int i = 0; i < dk8 * 8; i++) d[i] = s[i] * inv_v;
        for (int i = dk8 * 8; i < dk; i++) d[i] = s[i] * inv_norm * scale;
    }
}

void kro_la_conv(const float* qkvz, const float* ba, float* conv_state, const float* conv_w, const float* a_log,
                 const float* dt_bias, float scale, float* q, float* k, float* v, float* z, float* g, float* beta,
                 int nk, int nv, int dk, int dv, int kd, int sig_mode) {
    int hr = nv / nk; int group_dim = 2 * dk + 2 * dv * hr; int key_dim = nk * dk; int conv_dim = 2 * key_dim + nv * dv;
    float* mixed = (float*)malloc(4 * (size_t)conv_dim); float* co = (float*)malloc(4 * (size_t)conv_dim);
    for (int h = 0; h < nk; h++) {
        int src = h * group_dim;
        memcpy(mixed + h * dk, qkvz + src, 4 * (size_t)dk);
        memcpy(mixed + key_dim + h * dk, qkvz + src + dk, 4 * (size_t)dk);
        for (int r = 0; r < hr; r++) {
            int vh = h * hr + r;
            memcpy(mixed + 2 * key_dim + vh * dv, qkvz + src + 2 * dk + r * dv, 4 * (size_t)dv);
            memcpy(z + vh * dv, qkvz + src + 2 * dk + hr * dv + r * dv, 4 * (size_t)dv);
        }
    }
    if (kd == 4) {
        for (int ch = 0; ch < conv_dim; ch++) {
            int b = ch * 4; float s1 = conv_state[b + 1], s2 = conv_state[b + 2], s3 = conv_state[b + 3], s4 = mixed[ch];
            conv_state[b] = s1; conv_state[b + 1] = s2; conv_state[b + 2] = s3; conv_state[b + 3] = s4;
            co[ch] = s1 * conv_w[b] + s2 * conv_w[b + 1] + s3 * conv_w[b + 2] + s4 * conv_w[b + 3];
        }
    } else {
        for (int ch = 0; ch < conv_dim; ch++) {
            int b = ch * kd;
            for (int t = 0; t < kd - 1; t++) conv_state[b + t] = conv_state[b + t + 1];
            conv_state[b + kd - 1] = mixed[ch];
            float dot = 0.0f; for (int t = 0; t < kd; t++) dot += conv_state[b + t] * conv_w[b + t];
            co[ch] = dot;
        }
    }
    { /* fast_silu_avx2, decode.rs:1639 */
        int n8 = conv_dim / 8;
        for (int i = 0; i < n8 * 8; i++) co[i] = co[i] * kro_sigmoid(co[i], sig_mode);
        for (int i = n8 * 8; i < conv_dim; i++) { float x = co[i]; co[i] = x * (1.0f / (1.0f + expf(-x))); }
    }
    l2norm_expand(co, q, 0, dk, nv, hr, scale);
    l2norm_expand(co, k, key_dim, dk, nv, hr, 1.0f);
    memcpy(v, co + 2 * key_dim, 4 * (size_t)(nv * dv));
    for (int h = 0; h < nk; h++) {
        int src = h * 2 * hr;
        for (int r = 0; r < hr; r++) {
            int vh = h * hr + r; float b_raw = ba[src + r], a_p = ba[src + hr + r];
            beta[vh] = 1.0f / (1.0f + expf(-b_raw));
            float ap_dt = a_p + dt_bias[vh];
            float softplus = ap_dt > 20.0f ? ap_dt : logf(1.0f + expf(ap_dt));
            g[vh] = -(expf(a_log[vh])) * softplus;
        }
    }
    free(mixed); free(co);
}

void kro_la_recurrent(float* state, const float* q, const float* k, const float* v, const float* g, const float* beta,
                      float* out, int nv, int dk, int dv) {
    float* kv = (float*)malloc(4 * (size_t)dv); float* delta = (float*)malloc(4 * (size_t)dv); float* ob = (float*)malloc(4 * (size_t)dv);
    for (int h = 0; h < nv; h++) {
        float g_exp = expf(g[h]), beta_h = beta[h];
        float* S = state + (size_t)h * dk * dv;
        for (int j = 0; j < dv; j++) { kv[j] = 0.0f; ob[j] = 0.0f; }
        for (int i = 0; i < dk; i++) {
            float kk = k[h * dk + i];
            for (int j = 0; j < dv; j++) { float sd = S[i * dv + j] * g_exp; S[i * dv + j] = sd; kv[j] = fmaf(sd, kk, kv[j]); }
        }
        for (int j = 0; j < dv; j++) delta[j] = (v[h * dv + j] - kv[j]) * beta_h;
        for (int i = 0; i < dk; i++) {
            float kk = k[h * dk + i], qq = q[h * dk + i];
            for (int j = 0; j < dv; j++) { float sn = fmaf(kk, delta[j], S[i * dv + j]); S[i * dv + j] = sn; ob[j] = fmaf(sn, qq, ob[j]); }
        }
        memcpy(out + h * dv, ob, 4 * (size_t)dv);
    }
    free(kv); free(delta); free(ob);
}

void kro_gated_rmsnorm_silu(const float* recur, const float* z, const float* w, float* out, int nv, int dv, float eps, int sig_mode) {
    int dv8 = dv / 8;
    for (int h = 0; h < nv; h++) {
        int base = h * dv;
        float ss = sumsq_lanes(recur + base, dv);
        float rms = 1.0f / sqrtf(ss / (float)dv + eps);
        for (int i = 0; i < dv8 * 8; i++) {
            int o = base + i; float normed = (recur[o] * rms) * w[o];
            float zz = z[o]; float silu = zz * kro_sigmoid(zz, sig_mode);
            out[o] = silu * normed;
        }
        for (int i = dv8 * 8; i < dv; i++) {
            int o = base + i; float normed = recur[o] * rms * w[o]; float zz = z[o];
            out[o] = (zz / (1.0f + expf(-zz))) * normed;
        }
    }
}

/* ---- the stand-alone CpuDecodeStore operators (src/decode.rs:473-890): plain scalar loops and libm exp, NOT the decode graph's AVX2 forms
 * above.  Restated operation for operation (the oracle is built with -ffp-contract=off: `a += b * c` is a multiply and an add, as in rustc). */
void kro_op_rmsnorm(const float* x, const float* w, float* out, int n, float eps, int bias_one) { /* decode.rs:473-507 */
    float ss = 0.0f; for (int i = 0; i < n; i++) ss += x[i] * x[i];
    float rms = 1.0f / sqrtf(ss / (float)n + eps);
    if (bias_one) for (int i = 0; i < n; i++) out[i] = x[i] * rms * (1.0f + w[i]);
    else for (int i = 0; i < n; i++) out[i] = x[i] * rms * w[i];
}
void kro_op_silu_mul(const float* gate, const float* up, float* out, int n) { /* decode.rs:511-538 */
    for (int i = 0; i < n; i++) { float x = gate[i]; float sg = 1.0f / (1.0f + expf(-x)); out[i] = x * sg * up[i]; }
}
void kro_op_gated_rmsnorm_silu(const float* x, const float* z, const float* w, float* out, float eps, int nv, int dv) { /* decode.rs:650-695 */
    for (int h = 0; h < nv; h++) {
        int base = h * dv; float ss = 0.0f;
        for (int j = 0; j < dv; j++) ss += x[base + j] * x[base + j];
        float rms = 1.0f / sqrtf(ss / (float)dv + eps);
        for (int j = 0; j < dv; j++) {
            float normed = x[base + j] * rms * w[base + j]; float zv = z[base + j];
            float silu_z = zv / (1.0f + expf(-zv));
            out[base + j] = silu_z * normed;
        }
    }
}
void kro_op_la_conv(const float* qkvz, const float* ba, float* conv_state, const float* conv_w, const float* a_log, const float* dt_bias, float scale,
                    float* q, float* k, float* v, float* z, float* g, float* beta, int nk, int nv, int dk, int dv, int hr, int kd) { /* decode.rs:713-890 */
    int conv_dim = nk * dk * 2 + nv * dv, group_dim = 2 * dk + 2 * dv * hr, key_dim = nk * dk;
    float* mixed = (float*)calloc((size_t)conv_dim, 4); float* co = (float*)calloc((size_t)conv_dim, 4);
    float* b_raw = (float*)calloc((size_t)nv, 4); float* a_param = (float*)calloc((size_t)nv, 4);
    for (int h = 0; h < nk; h++) {
        int src = h * group_dim;
        memcpy(mixed + h * dk, qkvz + src, 4 * (size_t)dk);
        memcpy(mixed + key_dim + h * dk, qkvz + src + dk, 4 * (size_t)dk);
        for (int r = 0; r < hr; r++) {
            int vh = h * hr + r;
            memcpy(mixed + 2 * key_dim + vh * dv, qkvz + src + 2 * dk + r * dv, 4 * (size_t)dv);
            memcpy(z + vh * dv, qkvz + src + 2 * dk + hr * dv + r * dv, 4 * (size_t)dv);
        }
    }
    for (int h = 0; h < nk; h++) for (int r = 0; r < hr; r++) { b_raw[h * hr + r] = ba[h * 2 * hr + r]; a_param[h * hr + r] = ba[h * 2 * hr + hr + r]; }
    for (int ch = 0; ch < conv_dim; ch++) {
        int base = ch * kd;
        for (int t = 0; t < kd - 1; t++) conv_state[base + t] = conv_state[base + t + 1];
        conv_state[base + kd - 1] = mixed[ch];
    }
    for (int ch = 0; ch < conv_dim; ch++) {
        float dot = 0.0f;
        for (int t = 0; t < kd; t++) dot += conv_state[ch * kd + t] * conv_w[ch * kd + t];
        float sg = 1.0f / (1.0f + expf(-dot));
        co[ch] = dot * sg;
    }
    for (int vh = 0; vh < nv; vh++) {
        int kh = vh / hr, sb = kh * dk, db = vh * dk; float ss = 0.0f;
        for (int i = 0; i < dk; i++) { float val = co[sb + i]; ss += val * val; }
        float inv = ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f;
        for (int i = 0; i < dk; i++) q[db + i] = co[sb + i] * inv * scale;
    }
    for (int vh = 0; vh < nv; vh++) {
        int kh = vh / hr, sb = key_dim + kh * dk, db = vh * dk; float ss = 0.0f;
        for (int i = 0; i < dk; i++) { float val = co[sb + i]; ss += val * val; }
        float inv = ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f;
        for (int i = 0; i < dk; i++) k[db + i] = co[sb + i] * inv;
    }
    memcpy(v, co + 2 * key_dim, 4 * (size_t)(nv * dv));
    for (int h = 0; h < nv; h++) {
        beta[h] = 1.0f / (1.0f + expf(-b_raw[h]));
        float ap_dt = a_param[h] + dt_bias[h];
        float softplus = ap_dt > 20.0f ? ap_dt : logf(1.0f + expf(ap_dt));
        g[h] = -(expf(a_log[h])) * softplus;
    }
    free(mixed); free(co); free(b_raw); free(a_param);
}

/* torch.float8_e4m3fn codecs (c10/util/Float8_e4m3fn.h: fp8e4m3fn_from_fp32_value / to_fp32): RNE, |x| >= 480 -> NaN (0x7F), no saturation;
 * the reference's GPU KV cache dtype (python/krasis/kv_cache.py:38-135) */
uint8_t kro_f32_to_e4m3(float f) {
    uint32_t b; memcpy(&b, &f, 4);
    uint32_t sign = (b >> 24) & 0x80u, a = b & 0x7FFFFFFFu;
    if (a >= 0x43F00000u) return (uint8_t)(sign | 0x7Fu);
    if (a < 0x3C800000u) { float x, m; uint32_t mb = 0x46800000u, tb; memcpy(&x, &a, 4); memcpy(&m, &mb, 4); float t = x + m; memcpy(&tb, &t, 4); return (uint8_t)(sign | ((tb - 0x46800000u) & 0xFFu)); }
    uint32_t r = a + 0x7FFFFu + ((a >> 20) & 1u);
    r = (r - 0x3C000000u) >> 20;
    return (uint8_t)(sign | (r & 0x7Fu));
}
float kro_e4m3_to_f32(uint8_t x) {
    uint32_t m = x & 0x7Fu; float v;
    if (m == 0x7Fu) { uint32_t nb = 0x7FC00000u | ((uint32_t)(x & 0x80u) << 24); memcpy(&v, &nb, 4); return v; }
    if ((m >> 3) == 0) v = ldexpf((float)(m & 7u), -9);                 /* subnormal: m * 2^-9 */
    else v = ldexpf(1.0f + (float)(m & 7u) / 8.0f, (int)(m >> 3) - 7);
    return (x & 0x80u) ? -v : v;
}

static int g_kv_fp8 = 0;   /* element type used by kro_gqa_step for the caches (uint16 slots hold the byte when FP8) */
void kro_set_kv_fp8(int on) { g_kv_fp8 = on; }
static inline float kv_ld(const uint16_t* c, size_t i) { return g_kv_fp8 ? kro_e4m3_to_f32((uint8_t)c[i]) : kro_f16_to_f32(c[i]); }
static inline uint16_t kv_st(float v) { return g_kv_fp8 ? (uint16_t)kro_f32_to_e4m3(v) : kro_f32_to_f16(v); }

static float dot_f32_f16_lanes(const float* q, const uint16_t* c, int dim) { /* decode.rs:4229-4242 (single accumulator) */
    int hd8 = dim / 8; float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < hd8; b++) for (int j = 0; j < 8; j++) l[j] = fmaf(q[b * 8 + j], kv_ld(c, (size_t)(b * 8 + j)), l[j]);
    return hsum8(l);
}

void kro_gqa_step(const float* q_in, float* k, float* v, const float* q_norm, int q_norm_len, const float* k_norm,
                  int k_norm_len, int gated, int nh, int nkv, int hd, float eps, const float* rope_cos,
                  const float* rope_sin, int d2, uint16_t* k_cache, uint16_t* v_cache, int max_seq, int position,
                  float sm_scale, float* attn_out) {
    (void)max_seq;
    float* q = (float*)malloc(4 * (size_t)(nh * hd)); float* gate = (float*)malloc(4 * (size_t)(nh * hd));
    if (gated) { /* decode.rs:2875-2886 */
        for (int h = 0; h < nh; h++) for (int d = 0; d < hd; d++) { gate[h * hd + d] = q_in[h * hd * 2 + hd + d]; q[h * hd + d] = q_in[h * hd * 2 + d]; }
    } else memcpy(q, q_in, 4 * (size_t)(nh * hd));
    if (q_norm) for (int h = 0; h < nh; h++) {
        float* b = q + h * hd; float ss = 0.0f; for (int d = 0; d < hd; d++) ss += b[d] * b[d];
        float rms = 1.0f / sqrtf(ss / (float)hd + eps); const float* wq = q_norm + (q_norm_len == nh * hd ? h * hd : 0);
        for (int d = 0; d < hd; d++) b[d] *= rms * wq[d];
    }
    if (k_norm) for (int h = 0; h < nkv; h++) {
        float* b = k + h * hd; float ss = 0.0f; for (int d = 0; d < hd; d++) ss += b[d] * b[d];
        float rms = 1.0f / sqrtf(ss / (float)hd + eps); const float* wk = k_norm + (k_norm_len == nkv * hd ? h * hd : 0);
        for (int d = 0; d < hd; d++) b[d] *= rms * wk[d];
    }
    const float* cs = rope_cos + (size_t)position * d2; const float* sn = rope_sin + (size_t)position * d2;
    for (int h = 0; h < nh; h++) { float* b = q + h * hd; for (int i = 0; i < d2; i++) { float x1 = b[i], x2 = b[d2 + i]; b[i] = x1 * cs[i] - x2 * sn[i]; b[d2 + i] = x2 * cs[i] + x1 * sn[i]; } }
    for (int h = 0; h < nkv; h++) { float* b = k + h * hd; for (int i = 0; i < d2; i++) { float x1 = b[i], x2 = b[d2 + i]; b[i] = x1 * cs[i] - x2 * sn[i]; b[d2 + i] = x2 * cs[i] + x1 * sn[i]; } }
    int kvs = nkv * hd;
    for (int i = 0; i < kvs; i++) { k_cache[(size_t)position * kvs + i] = kv_st(k[i]); v_cache[(size_t)position * kvs + i] = kv_st(v[i]); }
    int seq = position + 1; int groups = nh / nkv;
    float* sc = (float*)malloc(4 * (size_t)seq);
    for (int h = 0; h < nh; h++) {
        int kvh = h / groups;
        for (int s = 0; s < seq; s++) sc[s] = dot_f32_f16_lanes(q + h * hd, k_cache + (size_t)s * kvs + kvh * hd, hd) * sm_scale;   /* cache element type: see kv_ld */
        float mx = -INFINITY; for (int s = 0; s < seq; s++) mx = maxf_rust(mx, sc[s]);
        float se = 0.0f; for (int s = 0; s < seq; s++) { sc[s] = expf(sc[s] - mx); se += sc[s]; }
        float inv = 1.0f / se; for (int s = 0; s < seq; s++) sc[s] *= inv;
        float* o = attn_out + h * hd; for (int d = 0; d < hd; d++) o[d] = 0.0f;
        for (int s = 0; s < seq; s++) { const uint16_t* vv = v_cache + (size_t)s * kvs + kvh * hd; float w = sc[s]; for (int d = 0; d < hd; d++) o[d] = fmaf(w, kv_ld(vv, (size_t)d), o[d]); }
    }
    if (gated) for (int i = 0; i < nh * hd; i++) { float sg = 1.0f / (1.0f + expf(-gate[i])); attn_out[i] *= sg; }
    free(q); free(gate); free(sc);
}

/* ---- G5: MLA decode attention (decode.rs:2993-3252) ---- */
/* mla_attn_dot_fp16_avx2 (decode.rs:4286): two 8-lane fma accumulators over alternating 8-blocks, (acc0+acc1) then hsum, scalar tail */
static float mla_dot_f32_f16(const float* q, const uint16_t* c, int dim) {
    int n8 = dim / 8; float a0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int chunks = n8 / 2, i = 0;
    for (int ch = 0; ch < chunks; ch++) {
        for (int j = 0; j < 8; j++) a0[j] = fmaf(q[i * 8 + j], kv_ld(c, (size_t)(i * 8 + j)), a0[j]);
        for (int j = 0; j < 8; j++) a1[j] = fmaf(q[(i + 1) * 8 + j], kv_ld(c, (size_t)((i + 1) * 8 + j)), a1[j]);
        i += 2;
    }
    if (n8 % 2) for (int j = 0; j < 8; j++) a0[j] = fmaf(q[i * 8 + j], kv_ld(c, (size_t)(i * 8 + j)), a0[j]);
    float s8[8]; for (int j = 0; j < 8; j++) s8[j] = a0[j] + a1[j];
    float r = hsum8(s8);
    for (int t = n8 * 8; t < dim; t++) r += q[t] * kv_ld(c, (size_t)t);
    return r;
}

/* plain sequential RMSNorm used for kv_a_norm / q_a_norm (decode.rs:3023-3032, 3053-3062): scalar sum, x *= rms * w */
void kro_rmsnorm_seq(float* x, const float* w, int n, float eps) {
    float ss = 0.0f; for (int i = 0; i < n; i++) ss += x[i] * x[i];
    float rms = 1.0f / sqrtf(ss / (float)n + eps);
    for (int i = 0; i < n; i++) x[i] *= rms * w[i];
}

/* kv_out = kv_a_proj output [klr + rd]; q_full = q (or q_b) projection output [nh * (nd + rd)] (modified in place like the reference);
 * w_kc f32 [nh, nd, klr]; w_vc f32 [nh, vhd, klr]; caches FP16 (or, after kro_set_kv_fp8(1), E4M3 bytes in the u16 slots -- an extension of the
 * reference GPU cache dtype to the decode store, no counterpart in decode.rs) [max_seq, klr] / [max_seq, rd]; out v_projected [nh * vhd]. */
void kro_mla_step(float* kv_out, float* q_full, const float* kv_a_norm, const float* w_kc, const float* w_vc,
                  const float* rope_cos, const float* rope_sin, int nh, int klr, int nd, int rd, int vhd, float eps, float sm_scale,
                  uint16_t* ckv_cache, uint16_t* kpe_cache, int position, float* v_projected) {
    int hd = nd + rd, half = rd / 2, seq = position + 1;
    float* ckv = (float*)malloc(4 * (size_t)klr);
    memcpy(ckv, kv_out, 4 * (size_t)klr);
    kro_rmsnorm_seq(ckv, kv_a_norm, klr, eps);
    float tmp[256];
    /* de-interleave k_pe (decode.rs:3098-3107) */
    for (int i = 0; i < half; i++) { tmp[i] = kv_out[klr + 2 * i]; tmp[half + i] = kv_out[klr + 2 * i + 1]; }
    memcpy(kv_out + klr, tmp, 4 * (size_t)rd);
    const float* cs = rope_cos + (size_t)position * half; const float* sn = rope_sin + (size_t)position * half;
    for (int h = 0; h < nh; h++) { /* decode.rs:3113-3128 */
        float* b = q_full + (size_t)h * hd + nd;
        for (int i = 0; i < half; i++) { tmp[i] = b[2 * i]; tmp[half + i] = b[2 * i + 1]; }
        for (int i = 0; i < half; i++) { float x1 = tmp[i], x2 = tmp[half + i]; b[i] = x1 * cs[i] - x2 * sn[i]; b[half + i] = x2 * cs[i] + x1 * sn[i]; }
    }
    { float* kpe = kv_out + klr; /* decode.rs:3131-3140 */
      for (int i = 0; i < half; i++) { float x1 = kpe[i], x2 = kpe[half + i]; kpe[i] = x1 * cs[i] - x2 * sn[i]; kpe[half + i] = x2 * cs[i] + x1 * sn[i]; } }
    /* absorb (decode.rs:4508): out[h][j] = fma(q[h][i], w_kc[h][i][j], out) for i ascending */
    float* qabs = (float*)calloc((size_t)nh * klr, 4);
    int klr8 = klr / 8;
    for (int h = 0; h < nh; h++) for (int i = 0; i < nd; i++) {
        float qv = q_full[(size_t)h * hd + i]; const float* wr = w_kc + ((size_t)h * nd + i) * klr; float* o = qabs + (size_t)h * klr;
        for (int j = 0; j < klr8 * 8; j++) o[j] = fmaf(qv, wr[j], o[j]);
    }
    for (int i = 0; i < klr; i++) ckv_cache[(size_t)position * klr + i] = kv_st(ckv[i]);
    for (int i = 0; i < rd; i++) kpe_cache[(size_t)position * rd + i] = kv_st(kv_out[klr + i]);
    float* sc = (float*)malloc(4 * (size_t)seq); float* ao = (float*)malloc(4 * (size_t)klr);
    for (int h = 0; h < nh; h++) {
        float mx = -INFINITY;
        for (int t = 0; t < seq; t++) {
            float s = mla_dot_f32_f16(qabs + (size_t)h * klr, ckv_cache + (size_t)t * klr, klr);
            s += mla_dot_f32_f16(q_full + (size_t)h * hd + nd, kpe_cache + (size_t)t * rd, rd);
            s *= sm_scale; sc[t] = s; if (s > mx) mx = s;
        }
        float se = 0.0f; for (int t = 0; t < seq; t++) { float e = expf(sc[t] - mx); sc[t] = e; se += e; }
        float inv = 1.0f / se; for (int t = 0; t < seq; t++) sc[t] *= inv;
        for (int j = 0; j < klr; j++) ao[j] = 0.0f;   /* decode.rs:4326: only the klr8*8 prefix is touched; klr % 8 == 0 in every model */
        for (int t = 0; t < seq; t++) { float w = sc[t]; const uint16_t* c = ckv_cache + (size_t)t * klr; for (int j = 0; j < klr8 * 8; j++) ao[j] = fmaf(w, kv_ld(c, (size_t)j), ao[j]); }
        /* w_vc projection (decode.rs:4555): two accumulators, (acc0+acc1), hsum */
        for (int o = 0; o < vhd; o++) {
            const float* wr = w_vc + ((size_t)h * vhd + o) * klr; float a0[8] = {0,0,0,0,0,0,0,0}, a1[8] = {0,0,0,0,0,0,0,0}; int chunks = klr8 / 2, j = 0;
            for (int ch = 0; ch < chunks; ch++) {
                for (int l = 0; l < 8; l++) a0[l] = fmaf(wr[j * 8 + l], ao[j * 8 + l], a0[l]);
                for (int l = 0; l < 8; l++) a1[l] = fmaf(wr[(j + 1) * 8 + l], ao[(j + 1) * 8 + l], a1[l]);
                j += 2;
            }
            if (klr8 % 2) for (int l = 0; l < 8; l++) a0[l] = fmaf(wr[j * 8 + l], ao[j * 8 + l], a0[l]);
            float s8[8]; for (int l = 0; l < 8; l++) s8[l] = a0[l] + a1[l];
            v_projected[(size_t)h * vhd + o] = hsum8(s8);
        }
    }
    free(ckv); free(qabs); free(sc); free(ao);
}

int kro_sample_greedy(const float* logits, int n) { /* decode.rs:3718: first max wins */
    int best = 0; float bv = logits[0];
    for (int i = 1; i < n; i++) if (logits[i] > bv) { bv = logits[i]; best = i; }
    return best;
}

/* sample_from_logits (decode.rs:3718-3811).  logits are modified in place (temperature) like the reference.  The reference orders the
 * top-k with sort_unstable_by on the value only; equal logits are ordered here by ascending token id (one valid outcome of that sort). */
static const float* g_sort_logits;
static int cmp_logit_desc(const void* a, const void* b) {
    int ia = *(const int*)a, ib = *(const int*)b; float va = g_sort_logits[ia], vb = g_sort_logits[ib];
    if (va > vb) return -1; if (va < vb) return 1; return ia < ib ? -1 : (ia > ib ? 1 : 0);
}
uint64_t kro_xorshift64_next(uint64_t* st) { uint64_t x = *st; x ^= x << 13; x ^= x >> 7; x ^= x << 17; *st = x; return x; }
int kro_sample_from_logits(float* logits, int vocab, float temperature, int top_k, float top_p, uint64_t* rng_state) {
    if (temperature == 0.0f) return kro_sample_greedy(logits, vocab);
    float inv_temp = 1.0f / temperature;
    for (int i = 0; i < vocab; i++) logits[i] *= inv_temp;
    int k = (top_k > 0 && top_k < vocab) ? top_k : vocab;
    int* idx = (int*)malloc(sizeof(int) * (size_t)vocab);
    for (int i = 0; i < vocab; i++) idx[i] = i;
    g_sort_logits = logits; qsort(idx, (size_t)vocab, sizeof(int), cmp_logit_desc);
    float* probs = (float*)malloc(4 * (size_t)k);
    float mx = logits[idx[0]], sum = 0.0f;
    for (int i = 0; i < k; i++) { probs[i] = expf(logits[idx[i]] - mx); sum += probs[i]; }
    float inv_sum = 1.0f / sum;
    for (int i = 0; i < k; i++) probs[i] *= inv_sum;
    int cutoff = k;
    if (top_p < 1.0f) { float cum = 0.0f; for (int i = 0; i < k; i++) { cum += probs[i]; if (cum >= top_p) { cutoff = i + 1; break; } } }
    if (cutoff < k) { float ns = 0.0f; for (int i = 0; i < cutoff; i++) ns += probs[i]; float inv = 1.0f / ns; for (int i = 0; i < cutoff; i++) probs[i] *= inv; }
    float r = (float)((double)kro_xorshift64_next(rng_state) / 18446744073709551615.0);
    int pick = cutoff - 1; float cum = 0.0f;
    for (int i = 0; i < cutoff; i++) { cum += probs[i]; if (r < cum) { pick = i; break; } }
    int tok = idx[pick];
    free(idx); free(probs);
    return tok;
}

/* ------------------------------------------------------------------ */
/* H: GPU prefill semantics (sglang fused_marlin_moe dataflow)          */
/* ------------------------------------------------------------------ */

static void dequant_col_unified(const kro_unified_expert* e, int which /*0=w13,1=w2*/, int col, float* wcol /*[K]*/) {
    int k = which == 0 ? e->hidden : e->inter; int n = which == 0 ? 2 * e->inter : e->hidden; int gs = e->gs;
    int bits = which == 0 ? e->num_bits : e->w2_bits;
    const uint16_t* sc = which == 0 ? e->w13_scales : e->w2_scales;
    if (bits == 4) {
        const uint32_t* p = (const uint32_t*)(which == 0 ? e->w13 : e->w2);
        for (int kk = 0; kk < k; kk++) {
            uint32_t word = p[(size_t)(kk / 8) * n + col]; int qv = (int)((word >> ((kk & 7) * 4)) & 0xF) - 8;
            /* Marlin dequant: (q as bf16) * scale(bf16) -> bf16 product */
            float s = kro_bf16_to_f32(sc[(size_t)(kk / gs) * n + col]);
            wcol[kk] = kro_bf16_to_f32(kro_f32_to_bf16((float)qv * s));
        }
    } else {
        const int8_t* p = (const int8_t*)(which == 0 ? e->w13 : e->w2);
        for (int kk = 0; kk < k; kk++) {
            float s = kro_bf16_to_f32(sc[(size_t)(kk / gs) * n + col]);
            wcol[kk] = kro_bf16_to_f32(kro_f32_to_bf16((float)p[(size_t)kk * n + col] * s));
        }
    }
}

void kro_moe_prefill_bf16(const kro_unified_expert* const* experts, int n_experts,
                          const uint16_t* x, const int32_t* ids, const float* w, int m, int topk, float rsf, uint16_t* out) {
    if (n_experts <= 0) return;
    int H = experts[0]->hidden, N = experts[0]->inter;
    float* wcol = (float*)malloc(4 * (size_t)(H > N ? H : N));
    /* per (token, slot): c1[2N] bf16 -> act bf16[N] -> c3[H] bf16 (with topk weight) */
    uint16_t* c3 = (uint16_t*)calloc((size_t)m * topk * H, 2);
    uint16_t* c1 = (uint16_t*)malloc(2 * (size_t)m * topk * 2 * N);
    /* expert-major so each column is dequantized once per expert */
    for (int e = 0; e < n_experts; e++) {
        int any = 0; for (int i = 0; i < m * topk; i++) if (ids[i] == e) { any = 1; break; }
        if (!any) continue;
        const kro_unified_expert* ex = experts[e];
        for (int c = 0; c < 2 * N; c++) {
            dequant_col_unified(ex, 0, c, wcol);
            for (int i = 0; i < m * topk; i++) if (ids[i] == e) {
                const uint16_t* xr = x + (size_t)(i / topk) * H; float acc = 0.0f;
                for (int kk = 0; kk < H; kk++) acc += kro_bf16_to_f32(xr[kk]) * wcol[kk];
                c1[(size_t)i * 2 * N + c] = kro_f32_to_bf16(acc);
            }
        }
        for (int i = 0; i < m * topk; i++) if (ids[i] == e) {
            uint16_t* r = c1 + (size_t)i * 2 * N;
            for (int c = 0; c < N; c++) { /* silu_and_mul in f32, bf16 out */
                float g = kro_bf16_to_f32(r[c]), u = kro_bf16_to_f32(r[N + c]);
                float silu = g / (1.0f + expf(-g));
                r[c] = kro_f32_to_bf16(silu * u);
            }
        }
        for (int c = 0; c < H; c++) {
            dequant_col_unified(ex, 1, c, wcol);
            for (int i = 0; i < m * topk; i++) if (ids[i] == e) {
                const uint16_t* hr = c1 + (size_t)i * 2 * N; float acc = 0.0f;
                for (int kk = 0; kk < N; kk++) acc += kro_bf16_to_f32(hr[kk]) * wcol[kk];
                c3[(size_t)i * H + c] = kro_f32_to_bf16(acc * w[i]); /* mul_topk_weights=True */
            }
        }
    }
    for (int t = 0; t < m; t++) for (int c = 0; c < H; c++) { /* moe_sum_reduce */
        float acc = 0.0f; for (int s = 0; s < topk; s++) acc += kro_bf16_to_f32(c3[((size_t)t * topk + s) * H + c]);
        out[(size_t)t * H + c] = kro_f32_to_bf16(acc * rsf);
    }
    free(wcol); free(c3); free(c1);
}

/* ------------------------------------------------------------------ */
/* J                                                                   */
/* ------------------------------------------------------------------ */

void kro_reduce_sum_bf16(const uint16_t* const* in, int ni, size_t n, uint16_t* out) {
    if (ni <= 0) return;
    if (ni == 1) { memcpy(out, in[0], n * 2); return; }
    for (size_t i = 0; i < n; i++) { float s = 0.0f; for (int p = 0; p < ni; p++) s = s + kro_bf16_to_f32(in[p][i]); out[i] = kro_f32_to_bf16(s); }
}

/* ------------------------------------------------------------------ */
/* synthetic generator                                                 */
/* ------------------------------------------------------------------ */

uint64_t kro_xs_next(kro_xorshift64* r) { uint64_t x = r->state; x ^= x << 13; x ^= x >> 7; x ^= x << 17; r->state = x; return x; }
uint32_t kro_xs_next_u32(kro_xorshift64* r) { return (uint32_t)kro_xs_next(r); }
void kro_xs_fill_u32(kro_xorshift64* r, uint32_t* d, size_t n) { for (size_t i = 0; i < n; i++) d[i] = kro_xs_next_u32(r); }
void kro_xs_fill_bf16_scales(kro_xorshift64* r, uint16_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) { float f = 0.005f + ((float)kro_xs_next_u32(r) / (float)UINT32_MAX) * 0.045f; d[i] = (uint16_t)(f32_bits(f) >> 16); }
}
void kro_xs_fill_f32_uniform(kro_xorshift64* r, float* d, size_t n, float amp) {
    for (size_t i = 0; i < n; i++) { uint64_t b = kro_xs_next(r); d[i] = (float)((double)(int64_t)b / (double)INT64_MAX) * amp; }
}

/* ------------------------------------------------------------------ */
/* CPU baseline: AVX2 twin of avx2.rs:1066 on the tiled layout         */
/* ------------------------------------------------------------------ */

static void int4_tile_avx2(const uint32_t* packed, const uint16_t* wsc, const int16_t* a, const float* a_s,
                           float* out, int k, int n_stride, int n_out, int gs) {
    int ng = k / gs, ppg = gs / 8, nb = n_out / 8;
    const __m256i mask_0f = _mm256_set1_epi32(0xF), off8 = _mm256_set1_epi32(8), mffff = _mm256_set1_epi32(0xFFFF);
    __m256i scratch[32];
    for (int b = 0; b < nb; b++) _mm256_storeu_ps(out + b * 8, _mm256_setzero_ps());
    for (int g = 0; g < ng; g++) {
        for (int b = 0; b < nb; b++) scratch[b] = _mm256_setzero_si256();
        for (int p = 0; p < ppg; p++) {
            int kr = g * ppg + p, kb = kr * 8; __m256i ap[4];
            for (int t = 0; t < 4; t++) ap[t] = _mm256_set1_epi32((int)(((uint32_t)(uint16_t)a[kb + 2 * t]) | ((uint32_t)(uint16_t)a[kb + 2 * t + 1] << 16)));
            const uint32_t* rowp = packed + (size_t)kr * n_stride;
            for (int b = 0; b < nb; b++) {
                __m256i words = _mm256_loadu_si256((const __m256i*)(rowp + b * 8)); __m256i acc = scratch[b];
                for (int t = 0; t < 4; t++) {
                    __m256i lo = _mm256_sub_epi32(_mm256_and_si256(_mm256_srli_epi32(words, 8 * t), mask_0f), off8);
                    __m256i hi = _mm256_sub_epi32(_mm256_and_si256(_mm256_srli_epi32(words, 8 * t + 4), mask_0f), off8);
                    __m256i pr = _mm256_or_si256(_mm256_and_si256(lo, mffff), _mm256_slli_epi32(hi, 16));
                    acc = _mm256_add_epi32(acc, _mm256_madd_epi16(pr, ap[t]));
                }
                scratch[b] = acc;
            }
        }
        __m256 asv = _mm256_set1_ps(a_s[g]);
        for (int b = 0; b < nb; b++) {
            __m256 gf = _mm256_cvtepi32_ps(scratch[b]);
            __m128i sb = _mm_loadu_si128((const __m128i*)(wsc + (size_t)g * n_stride + b * 8));
            __m256 wsv = _mm256_castsi256_ps(_mm256_slli_epi32(_mm256_cvtepu16_epi32(sb), 16));
            __m256 comb = _mm256_mul_ps(wsv, asv);
            _mm256_storeu_ps(out + b * 8, _mm256_fmadd_ps(gf, comb, _mm256_loadu_ps(out + b * 8)));
        }
    }
}

void kro_matvec_int4_tiled_avx2(const uint32_t* pt, const uint16_t* st, const int16_t* a, const float* a_s,
                                int k, int n, int gs, float* out, int parallel) {
    int k_rows = k / 8, ng = k / gs; int nt = (n + KRO_TILE_N - 1) / KRO_TILE_N;
    (void)parallel;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (parallel && nt > 1)
#endif
    for (int t = 0; t < nt; t++) {
        int n0 = t * KRO_TILE_N; int tn = n - n0 < KRO_TILE_N ? n - n0 : KRO_TILE_N;
        int4_tile_avx2(pt + (size_t)t * k_rows * KRO_TILE_N, st + (size_t)t * ng * KRO_TILE_N, a, a_s, out + n0, k, KRO_TILE_N, tn & ~7, gs);
    }
}

/* CPU baseline for one MoE layer/token: the reference's moe_forward_unified on TILED weights with the AVX2 integer kernel,
 * parallelised as 3 flat phases over (expert, 256-column tile) work items (the reference's moe_forward_flattened shape,
 * moe.rs:727; results are bit-identical to the nested-rayon production path because every column is independent). */
void kro_moe_forward_unified_tiled_avx2(const kro_unified_expert* const* ex, const float* weights, int n_sel,
                                        const uint16_t* act, int sig_mode, float* out) {
    if (n_sel <= 0) return;
    const int H = ex[0]->hidden, I = ex[0]->inter, gs = ex[0]->gs;
    int16_t* q = (int16_t*)malloc(2 * (size_t)H); float* qs = (float*)malloc(4 * (size_t)(H / gs));
    float* gu = (float*)malloc(4 * (size_t)n_sel * 2 * I); float* eo = (float*)malloc(4 * (size_t)n_sel * H);
    int16_t* hq = (int16_t*)malloc(2 * (size_t)n_sel * I); float* hs = (float*)malloc(4 * (size_t)n_sel * (I / gs));
    kro_quant_act_int16_bf16(act, H, gs, q, qs);
    const int t13 = (2 * I + KRO_TILE_N - 1) / KRO_TILE_N, t2 = (H + KRO_TILE_N - 1) / KRO_TILE_N;
    const int kr13 = H / 8, ng13 = H / gs, kr2 = I / 8, ng2 = I / gs;
#pragma omp parallel
    {
#pragma omp for schedule(dynamic, 1)
        for (int it = 0; it < n_sel * t13; it++) {
            int e = it / t13, t = it % t13; int n0 = t * KRO_TILE_N; int tn = 2 * I - n0 < KRO_TILE_N ? 2 * I - n0 : KRO_TILE_N;
            int4_tile_avx2((const uint32_t*)ex[e]->w13 + (size_t)t * kr13 * KRO_TILE_N, ex[e]->w13_scales + (size_t)t * ng13 * KRO_TILE_N,
                           q, qs, gu + (size_t)e * 2 * I + n0, H, KRO_TILE_N, tn & ~7, gs);
        }
#pragma omp for schedule(dynamic, 1)
        for (int e = 0; e < n_sel; e++)
            kro_silu_quant_int16(gu + (size_t)e * 2 * I, gu + (size_t)e * 2 * I + I, I, gs, sig_mode, NULL, hq + (size_t)e * I, hs + (size_t)e * (I / gs));
#pragma omp for schedule(dynamic, 1)
        for (int it = 0; it < n_sel * t2; it++) {
            int e = it / t2, t = it % t2; int n0 = t * KRO_TILE_N; int tn = H - n0 < KRO_TILE_N ? H - n0 : KRO_TILE_N;
            int4_tile_avx2((const uint32_t*)ex[e]->w2 + (size_t)t * kr2 * KRO_TILE_N, ex[e]->w2_scales + (size_t)t * ng2 * KRO_TILE_N,
                           hq + (size_t)e * I, hs + (size_t)e * (I / gs), eo + (size_t)e * H + n0, I, KRO_TILE_N, tn & ~7, gs);
        }
    }
    for (int j = 0; j < H; j++) out[j] = 0.0f;
    for (int e = 0; e < n_sel; e++) { float w = weights[e]; for (int j = 0; j < H; j++) out[j] += w * eo[(size_t)e * H + j]; }
    free(q); free(qs); free(gu); free(eo); free(hq); free(hs);
}

void kro_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int kro_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
