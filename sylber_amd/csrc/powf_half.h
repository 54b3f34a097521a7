// powf(x, 0.5f) exactly as glibc 2.35 x86_64 computes it (sysdeps/ieee754/flt-32/e_powf.c, the
// FMA/AVX2 ifunc variant __powf_fma that every FMA-capable host selects), restated so that the
// device-side segmenter reproduces numpy's *scalar* `x ** .5`.
//
// Why: the reference's cossim (sylber/utils/segment_utils.py:68-69) evaluates
// `((x**2).sum(-1)+1e-8)**.5`; for 1-D inputs (segment_utils.py:97,114) the sum is a numpy float32
// SCALAR and `** .5` is libm powf, which is not correctly rounded (0.82 ULP) and differs from sqrtf
// on ~0.06 % of inputs.  Segment boundaries are decided by `>=` on these values, so bit-exact
// boundaries need bit-exact powf.  The algorithm (Szabolcs Nagy's optimized-routines powf) is:
//   log2(x)  : 16-entry table {1/c, log2 c} + degree-5 polynomial in r = z/c - 1, all in double
//   exp2(y)  : 32-entry table 2^(k/32) + degree-3 polynomial, double, result rounded to float once
// Tables/coefficients are data (public glibc constants `__powf_log2_data`, `__exp2f_data`); every
// a*b+c below is a fused multiply-add exactly where the FMA variant fuses it, plain ops elsewhere.
// tests/test_powf_replica.py checks this header against the host libm over 2^31 inputs (sampled by
// default, exhaustive with SYLBER_EXHAUSTIVE=1).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define PH_FN __host__ __device__ inline
#else
#define PH_FN static inline
#endif

#ifndef PH_FMA
#define PH_FMA(a, b, c) __builtin_fma((a), (b), (c))
#endif

PH_FN double ph_u2d(uint64_t u) { union { uint64_t u; double d; } v; v.u = u; return v.d; }
PH_FN uint64_t ph_d2u(double d) { union { uint64_t u; double d; } v; v.d = d; return v.u; }
PH_FN float ph_u2f(uint32_t u) { union { uint32_t u; float f; } v; v.u = u; return v.f; }
PH_FN uint32_t ph_f2u(float f) { union { uint32_t u; float f; } v; v.f = f; return v.u; }

PH_FN float powf_half_glibc(float x) {
    // {invc, logc} of __powf_log2_data.tab (POWF_SCALE = 1)
    const uint64_t LT[32] = {
        0x3ff661ec79f8f3beull, 0xbfdefec65b963019ull, 0x3ff571ed4aaf883dull, 0xbfdb0b6832d4fca4ull,
        0x3ff49539f0f010b0ull, 0xbfd7418b0a1fb77bull, 0x3ff3c995b0b80385ull, 0xbfd39de91a6dcf7bull,
        0x3ff30d190c8864a5ull, 0xbfd01d9bf3f2b631ull, 0x3ff25e227b0b8ea0ull, 0xbfc97c1d1b3b7af0ull,
        0x3ff1bb4a4a1a343full, 0xbfc2f9e393af3c9full, 0x3ff12358f08ae5baull, 0xbfb960cbbf788d5cull,
        0x3ff0953f419900a7ull, 0xbfaa6f9db6475fceull, 0x3ff0000000000000ull, 0x0000000000000000ull,
        0x3fee608cfd9a47acull, 0x3fb338ca9f24f53dull, 0x3feca4b31f026aa0ull, 0x3fc476a9543891baull,
        0x3feb2036576afce6ull, 0x3fce840b4ac4e4d2ull, 0x3fe9c2d163a1aa2dull, 0x3fd40645f0c6651cull,
        0x3fe886e6037841edull, 0x3fd88e9c2c1b9ff8ull, 0x3fe767dcf5534862ull, 0x3fdce0a44eb17bccull};
    // __exp2f_data.tab: 2^(i/32) with the exponent bits pre-subtracted
    const uint64_t ET[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
        0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
        0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2,
                 A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
    const double SHIFT = 0x1.8p+47;   // 0x1.8p52 / 32

    uint32_t ix = ph_f2u(x);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        // zero, subnormal, negative, inf, nan (never reached from cossim: argument >= 1e-8)
        if (ix * 2u == 0u) return 0.0f;                         // +-0 ** .5 = 0
        if (ix == 0x7f800000u) return x;                        // inf
        if (ix > 0x7f800000u) return ph_u2f(0x7fc00000u);       // nan or negative
        ix = ph_f2u(x * 0x1p23f);                               // subnormal: normalise
        ix &= 0x7fffffffu;
        ix -= 23u << 23;
    }
    // ---- log2_inline
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    const double invc = ph_u2d(LT[2 * i]), logc = ph_u2d(LT[2 * i + 1]);
    const double z = (double)ph_u2f(iz);
    const double r = PH_FMA(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double y = PH_FMA(A0, r, A1);
    const double p = PH_FMA(A2, r, A3);
    const double r4 = r2 * r2;
    double q = PH_FMA(A4, r, y0);
    q = PH_FMA(p, r2, q);
    y = PH_FMA(y, r4, q);
    const double ylogx = 0.5 * y;
    // ---- exp2_inline (sign_bias = 0; |ylogx| < 126 always for y = .5)
    double kd = ylogx + SHIFT;
    const uint64_t ki = ph_d2u(kd);
    kd -= SHIFT;
    const double rr = ylogx - kd;
    uint64_t t = ET[ki & 31u];
    t += ki << 47;
    const double s = ph_u2d(t);
    const double zz = PH_FMA(C0, rr, C1);
    const double rr2 = rr * rr;
    double yy = PH_FMA(C2, rr, 1.0);
    yy = PH_FMA(zz, rr2, yy);
    yy = yy * s;
    return (float)yy;
}
