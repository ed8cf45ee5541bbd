// kr_libm.h -- device math whose results must equal the host libm the reference calls.
//
// The reference's Rust code calls f32::exp -> libm expf (glibc on Linux) in the router softmax/sigmoid
// (src/moe.rs:3194-3206, src/decode.rs:4156), the attention softmax (decode.rs:4250) and the LA gates
// (decode.rs:3897-3900).  To keep router top-k ids and weights bit-identical we evaluate expf with glibc's
// own algorithm (sysdeps/ieee754/flt-32/e_expf.c, from ARM optimized-routines: N=32 table, cubic in double).
// The table is 2^(i/32) with the exponent pre-biased (derived from first principles in tests/test_libm.py);
// the CPU twin of this function matched host expf on 2e8 inputs with zero mismatches.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ static const uint64_t kr_exp2f_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

__device__ __forceinline__ float kr_expf(float x) {
    if (x != x) return x + x;
    if (x > 0x1.62e42ep6f) return __builtin_inff();
    if (x < -0x1.9fe368p6f) return 0.0f;
    const double N = 32.0;
    const double InvLn2N = 0x1.71547652b82fep+0 * N;
    const double SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
    const double z = InvLn2N * (double)x;
    double kd = z + SHIFT;
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd -= SHIFT;
    const double r = z - kd;
    uint64_t t = kr_exp2f_tab[ki & 31];
    t += ki << (52 - 5);
    const double s = __longlong_as_double((long long)t);
    const double zz = __builtin_fma(C0, r, C1);
    const double r2 = r * r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(zz, r2, y);
    y = y * s;
    return (float)y;
}

// glibc logf (sysdeps/ieee754/flt-32/e_logf.c, LOGF_TABLE_BITS = 4): table {1/c, log(c)}, cubic in double.  The table
// below is __logf_data of the pinned libm (glibc 2.35); the CPU twin matched host logf on 2e8 inputs, zero mismatches.
__device__ static const double kr_logf_tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2}, {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};

__device__ __forceinline__ float kr_logf(float x) {
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2 == 0) return -__builtin_inff();
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return __builtin_nanf("");
        ix = __float_as_uint(x * 0x1p23f);
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (tmp >> (23 - 4)) % 16;
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    const double invc = kr_logf_tab[i][0], logc = kr_logf_tab[i][1];
    const double z = (double)__uint_as_float(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = __builtin_fma((double)k, 0x1.62e42fefa39efp-1, logc);
    const double r2 = r * r;
    double y = __builtin_fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = __builtin_fma(-0x1.00ea348b88334p-2, r2, y);
    y = __builtin_fma(y, r2, y0 + r);
    return (float)y;
}
