// Shared device/host helpers for libsylber_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bf16 bits in HBM
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4_t __attribute__((ext_vector_type(4)));

#define SYL_HIDDEN 768
#define SYL_CONV 512
#define SYL_HEADS 12
#define SYL_HDIM 64
#define SYL_FFN 3072
#define SYL_POSK 128
#define SYL_POSG 16
#define SYL_POSC 48

// ---- bf16 <-> f32 (round to nearest even; NaN kept quiet) ---------------------------------------
__host__ __device__ __forceinline__ bf16_t f2bf(float f) {
    union { float f; uint32_t u; } v; v.f = f;
    uint32_t u = v.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__host__ __device__ __forceinline__ float bf2f(bf16_t h) {
    union { float f; uint32_t u; } v; v.u = ((uint32_t)h) << 16;
    return v.f;
}
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// device-side packing: the fptrunc lowers to v_cvt_pk_bf16_f32 (RNE) on gfx950
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    // an explicit <2 x float> -> <2 x bfloat> truncation: one v_cvt_pk_bf16_f32 without relying on the SLP vectorizer
    // (which is off, see build.py)
    typedef float f32pair_t __attribute__((ext_vector_type(2)));
    const f32pair_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf_dev(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

// ---- the 16-bit operand format of the MFMA path -------------------------------------------------------------------
// FMT_BF16 (default, BASELINE configs[1]): 8 exponent / 7 mantissa bits.  FMT_F16 (precision "fp16"): IEEE half, 10
// mantissa bits = 8x finer rounding of every activation / weight hand-over at the SAME matrix-pipe rate
// (v_mfma_f32_32x32x16_f16); conversions saturate at +-65504 instead of overflowing to infinity.  Buffers hold raw 16-bit
// words either way (bf16_t = unsigned short), so layouts, LDS images and LDS-DMA staging are identical.
// FMT_SPLIT (precision "split16"): every 16-bit operand is a PAIR of IEEE halves (hi = half(x), lo = half(x - hi):
// 22 significand bits) kept as two planes of the same buffer; a contraction runs three MFMA passes into one fp32
// accumulator, hi.hi + lo.hi + hi.lo (the lo.lo term is below 2^-22 relative), i.e. fp32-grade products at three
// times the fp16 MFMA work instead of the sixteen times of the f32 MFMA.
enum { FMT_BF16 = 0, FMT_F16 = 1, FMT_SPLIT = 2 };
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
// q leaves its projection pre-scaled by 64^-0.5 * log2(e): the attention scores q . k are then in log2 units, so the softmax is
// exp2(s - m) with no multiply per score (round 5; every 16-bit attention kernel and the q / k / v epilogues share this constant --
// the MXFP8 attention core keeps its own 2^-3 in EPI_QK8)
#define SYL_Q_SCALE 0.18033688011112042f
template <int FMT> struct H16 {
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
    static __device__ __forceinline__ uint32_t pack2_bounded(float lo, float hi) { return pack_bf16x2(lo, hi); }   // |x| known < 65504
    static __device__ __forceinline__ bf16_t cvt(float f) { return f2bf_dev(f); }
    static __device__ __forceinline__ float up(bf16_t h) { return bf2f(h); }
    static __device__ __forceinline__ f32x16_t mfma(bf16x8_t a, bf16x8_t b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct H16<FMT_F16> {
    static __device__ __forceinline__ float sat(float f) { return __builtin_amdgcn_fmed3f(f, -65504.0f, 65504.0f); }
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        typedef float f32pair_t __attribute__((ext_vector_type(2)));
        const f32pair_t v = {sat(lo), sat(hi)};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
    }
    static __device__ __forceinline__ uint32_t pack2_bounded(float lo, float hi) {
        typedef float f32pair_t __attribute__((ext_vector_type(2)));
        const f32pair_t v = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
    }
    static __device__ __forceinline__ bf16_t cvt(float f) { return __builtin_bit_cast(unsigned short, (_Float16)sat(f)); }
    static __device__ __forceinline__ float up(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
    static __device__ __forceinline__ f32x16_t mfma(bf16x8_t a, bf16x8_t b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct H16<FMT_SPLIT> : H16<FMT_F16> {
    // the lo halves of two values whose packed hi halves are `hi2`
    static __device__ __forceinline__ uint32_t pack2_lo(float lo, float hi, uint32_t hi2) {
        const f16x2_t h = __builtin_bit_cast(f16x2_t, hi2);
        return H16<FMT_F16>::pack2_bounded(lo - (float)h[0], hi - (float)h[1]);
    }
    static __device__ __forceinline__ bf16_t cvt_lo(float f, bf16_t hi) { return H16<FMT_F16>::cvt(f - H16<FMT_F16>::up(hi)); }
};
// host: float -> IEEE half, round to nearest even, saturating (weights packed at sylber_create)
__host__ __forceinline__ bf16_t f2h_host(float f) {
    union { float f; uint32_t u; } v; v.f = f;
    const uint32_t sign = (v.u >> 16) & 0x8000u;
    uint32_t a = v.u & 0x7fffffffu;
    if (a > 0x7f800000u) return (bf16_t)(sign | 0x7e00u);                 // NaN
    if (a >= 0x477ff000u) return (bf16_t)(sign | 0x7bffu);                // >= 65520 rounds beyond max: saturate at 65504
    if (a < 0x38800000u) {                                                // subnormal half (or zero)
        if (a < 0x33000000u) return (bf16_t)sign;
        const int e = (int)(a >> 23);
        const uint32_t m = (a & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e;                                        // 14 .. 24
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return (bf16_t)(sign | r);
    }
    a += 0xc8000000u;                                                     // rebias exponent 127 -> 15
    a += 0xfffu + ((a >> 13) & 1u);
    return (bf16_t)(sign | (a >> 13));
}


__host__ __forceinline__ float h2f_host(bf16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    union { float f; uint32_t u; } v;
    if (e == 0) {
        if (m == 0) { v.u = sign; return v.f; }
        int sh = 0;
        while (!(m & 0x400u)) { m <<= 1; ++sh; }
        m &= 0x3ffu; e = 1 - sh;
        v.u = sign | ((uint32_t)(e + 112) << 23) | (m << 13);
        return v.f;
    }
    if (e == 31) { v.u = sign | 0x7f800000u | (m << 13); return v.f; }
    v.u = sign | ((e + 112) << 23) | (m << 13);
    return v.f;
}

// ---- GELU --------------------------------------------------------------------------------------
// exact erf form (reference: transformers activations "gelu" = 0.5 x (1 + erf(x / sqrt 2)))
// written with an explicit fma: left to -ffp-contract the same expression compiled to different roundings in different
// kernels (direct vs staged epilogue, 4- vs 8-wave), which made the split16 mode's results depend on the tile shape the
// cost model picked, i.e. on the batch size (bf16 / fp16 outputs hide a last-bit difference, 22-bit planes do not)
__device__ __forceinline__ float gelu_erf(float x) {
    const float hx = 0.5f * x;
    return __builtin_fmaf(hx, erff(x * 0.70710678118654752f), hx);
}
// erf-form GELU for the split16 mode: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32-grade: erff's own
// error is 6e-8) in ~15 VALU + 2 transcendental instructions; ocml's erff costs several times that and made the GELU
// epilogues of the split16 GEMMs as long as their (three-pass) K loops.  Explicit fma everywhere: bit-identical in every
// kernel it is inlined into.
__device__ __forceinline__ float gelu_erf7(float x) {
    const float z = x * 0.70710678118654752f;
    const float az = __builtin_fabsf(z);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, az, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    p = p * t;
    const float e = __builtin_amdgcn_exp2f(az * az * -1.44269504088896341f);
    const float erf_abs = __builtin_fmaf(-p, e, 1.0f);
    const float erf = __builtin_copysignf(erf_abs, z);
    const float hx = 0.5f * x;
    return __builtin_fmaf(hx, erf, hx);
}
// 16-bit-path GELU without transcendentals.  gelu(x) = x Phi(x), Phi(x) - 1/2 = erf(x / sqrt 2) / 2 is ODD:
// Phi(x) = 1/2 + x Q(x^2) with a degree-8 polynomial Q on |x| <= 4.2 (Chebyshev fit of x^2 Q(x^2) to the even part of
// gelu: max |error| of gelu 6.4e-5 over the whole real line, gelu(0) = 0 exactly); beyond, x is clamped inside Phi only:
// Phi(+-4.2) = 1 - 2.7e-5 / 2.7e-5, i.e. gelu = 0.99997 x resp. 2.7e-5 x.  12 VALU instructions (clamp, square, 8 FMA,
// one FMA for Phi, one multiply), no compare / select (a v_cmp + v_cndmask form of the range split returned the PREVIOUS
// compare's result in lanes 48-63 when the wave shared its SIMD with MFMA waves of another kernel,
// profiles/r02_packed_f32_hazard.md), all scalar fp32 (packed fp32 is banned, build.py).  The GELU of a 256x256 tile
// costs about as much VALU issue as a K = 768 tile costs matrix-pipe time.  The result is rounded to a 16-bit operand
// (half-ulp 2e-3 resp. 2.4e-4 at |y| = 1) right after; the fp32 parity mode uses erff (gelu_erf).
__device__ __forceinline__ float gelu_fast(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.2f, 4.2f);
    const float u = xc * xc;
    float q = fmaf(6.949803233e-11f, u, -6.356798643e-09f);
    q = fmaf(q, u, 2.570604920e-07f);
    q = fmaf(q, u, -6.139445304e-06f);
    q = fmaf(q, u, 9.818511899e-05f);
    q = fmaf(q, u, -1.133762766e-03f);
    q = fmaf(q, u, 9.886963293e-03f);
    q = fmaf(q, u, -6.643489748e-02f);
    q = fmaf(q, u, 3.989362717e-01f);
    return x * fmaf(xc, q, 0.5f);
}

// A/B (round 5): sigmoid forms of the same function, gelu(x) = x / (1 + exp2(-x P(x^2))), two transcendentals (exp2, rcp) instead of the
// degree-8 polynomial.  SYL_GELU_VARIANT 1: P = p0 + p1 u, 7 instructions, max |error| 2.7e-4; 2: P = p0 + p1 u + p2 u^2 with u clamped at
// 36 (p2 < 0: unclamped, the argument changes sign beyond |x| = 11), 9 instructions, 2.5e-5 (fits: minimax over [-12, 12], fp32 evaluation)
#ifndef SYL_GELU_VARIANT
#define SYL_GELU_VARIANT 0
#endif
__device__ __forceinline__ float gelu_sig2(float x) {
    const float u = x * x;
    const float t = fmaf(0.1001257f, u, 2.30876518f) * x;
    const float e = __builtin_amdgcn_exp2f(-t);
    return x * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ float gelu_sig3(float x) {
    const float u = fminf(x * x, 36.0f);
    const float t = fmaf(fmaf(-1.01424778e-03f, u, 1.06775700e-01f), u, 2.30112128f) * x;
    const float e = __builtin_amdgcn_exp2f(-t);
    return x * __builtin_amdgcn_rcpf(e + 1.0f);
}
#if SYL_GELU_VARIANT == 1
#define gelu_fast gelu_sig2
#elif SYL_GELU_VARIANT == 2
#define gelu_fast gelu_sig3
#endif

// two values of one run.  (This used to go through the packed-fp32 VALU -- v_pk_fma_f32 carries two lanes' worth of
// work per issue -- and was measured neutral; packed fp32 is now banned from the library, see build.py.)
__device__ __forceinline__ void gelu_fast2(float& a, float& b) { a = gelu_fast(a); b = gelu_fast(b); }

// ---- OCP microscaling FP8 (MXFP8: e4m3 elements, one E8M0 power-of-two scale per 32 elements along K) ----------
// scale of a block with absolute maximum `amax`: the smallest 2^e with amax <= 448 * 2^e (448 = e4m3 max), so no
// element saturates; returned biased (e + 127), clamped to [0, 254]; an all-zero block gets 1.0
__device__ __forceinline__ unsigned mx_e8m0(float amax) {
    const unsigned u = __float_as_uint(amax);
    if (u == 0u) return 127u;
    const int b = (int)(u >> 23) - 8 + ((u & 0x7fffffu) > 0x600000u ? 1 : 0);   // 448 = 1.75 * 2^8
    return (unsigned)(b < 0 ? 0 : (b > 254 ? 254 : b));
}
// scale storage: "K-pair-major" [K/64][rows][2] — the two scales a row contributes to one 64-wide MFMA K slice are
// adjacent, and the scales of 8 consecutive rows for one slice are one 16-byte run (one lane of an LDS-DMA instruction)
__host__ __device__ __forceinline__ size_t mx_scale_index(long row, int block, long rows) {
    return (size_t)(block >> 1) * (size_t)rows * 2 + (size_t)row * 2 + (size_t)(block & 1);
}
__device__ __forceinline__ float mx_inv_scale(unsigned e8m0) { return __uint_as_float((254u - e8m0) << 23); }
// four floats -> four e4m3 bytes (v_cvt_pk_fp8_f32: round to nearest even, OCP format on gfx950)
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}

// ---- K order of the 3-tap stride-2 conv layers (GemmArgs::kpat) --------------------------------------------------------------
// The operand row of output frame m is ONE run of 3 x 512 channels-last values starting at input frame 2 m; its third tap is the
// first tap of frame m + 1.  Walked tap by tap, a tile re-reads those shared rows a 512-channel sweep later, when the XCD's L2 has
// long dropped them (profiles/r04_conv_fetch_account.md: 1.5x over-fetch).  Walked chunk-major -- for each 64-channel chunk: tap 0,
// tap 2, tap 1 -- the re-read follows one K step later.  Byte position o of that walk (128 bytes per chunk-tap) -> byte offset
// inside the operand row.  Every kernel that runs these layers uses this one order (results must not depend on the tile shape).
__host__ __device__ __forceinline__ int tap3_offset(int o) {
    const int q = o >> 7, c = q / 3, i = q - 3 * c;
    return (c << 7) + (i == 0 ? 0 : (i == 1 ? 2048 : 1024)) + (o & 127);
}

// ---- wave64 reductions ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// bijective XCD-aware remap of a linear workgroup id: consecutive ids land on consecutive XCDs
// (observed dispatch: block b -> XCD b % 8); give each XCD a contiguous chunk of the tile space so
// neighbouring tiles (sharing an operand panel) share one L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) { syl_set_error(#expr, hipGetErrorString(_e)); return 1; }      \
    } while (0)

void syl_set_error(const char* what, const char* detail);

// "has this kernel's dynamic-LDS attribute been raised on the CURRENT device yet" — function attributes are per device,
// and a process may hold handles on several GPUs; one instance per launcher template instantiation
struct PerDeviceOnce {
    bool done[64] = {};
    bool need() { int d = 0; (void)hipGetDevice(&d); d &= 63; if (done[d]) return false; done[d] = true; return true; }
};
