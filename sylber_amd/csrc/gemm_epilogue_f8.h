// The quantising (MXFP8) output epilogue of the fp8 GEMMs, shared by gemm_mxfp8.hip and gemm_asm_f8.hip.
#pragma once
#include "kernels.h"
#include "gemm_epilogue.h"

// ---- MXFP8 output epilogue (FFN1): act(acc + bias) -> e4m3 + one E8M0 scale per (token, 32 features) ----------
// A 32x32 fragment IS one scale block per token: a lane holds 16 of its token's 32 values, lane^32 the other 16.
// Rows are transposed through a private LDS region and leave as 16-byte chunks over whole lines.
template <int FN>
struct StagedF8 {
    static constexpr int ROWB = 32 * FN;            // payload bytes per row
    static constexpr int RS = ROWB + 16;
    static constexpr int CH = ROWB / 16;
    static constexpr int BYTES = 32 * RS;
};

template <int FN, int ACT>
__device__ __forceinline__ void epilogue_mxfp8_rows32(const GemmF8Args& a, const f32x16_t (&acc)[FN], const float4 (&bias)[FN][4], int mrow0,
                                                      int ncol0, char* lds, int lane) {
    using S = StagedF8<FN>;
    const int ml = lane & 31, h = lane >> 5;
    const int m = mrow0 + ml;
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        float v[16];
        float amax = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bb = bias[fn][g];
            v[4 * g + 0] = acc[fn][4 * g + 0] + bb.x; v[4 * g + 1] = acc[fn][4 * g + 1] + bb.y;
            v[4 * g + 2] = acc[fn][4 * g + 2] + bb.z; v[4 * g + 3] = acc[fn][4 * g + 3] + bb.w;
            apply_act4<ACT>(v[4 * g + 0], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fabsf(v[4 * g + j]));
        }
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        const unsigned e = mx_e8m0(amax);
        const float inv = mx_inv_scale(e);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(unsigned*)(lds + ml * S::RS + 32 * fn + 8 * g + 4 * h) =
                pack_fp8x4(v[4 * g + 0] * inv, v[4 * g + 1] * inv, v[4 * g + 2] * inv, v[4 * g + 3] * inv);
        if (h == 0 && m < a.g.M && ncol0 + 32 * fn < a.g.N) a.out_scale[mx_scale_index(m, (ncol0 + 32 * fn) >> 5, a.os_rows)] = (uint8_t)e;
    }
#pragma unroll
    for (int it = 0; it < (S::CH + 1) / 2; ++it) {
        const int idx = it * 64 + lane;
        const int r = idx / S::CH, c = idx - r * S::CH;
        const int mo = mrow0 + r;
        const int n = ncol0 + c * 16;
        if (r >= 32 || mo >= a.g.M || n >= a.g.N) continue;
        *(uint4*)((uint8_t*)a.g.out0 + (size_t)mo * a.g.ld0 + n) = *(const uint4*)(lds + r * S::RS + c * 16);
    }
}


// ---- EPI_QK8: the fused q / k / v projection with MXFP8 outputs for the fp8 attention core (attention.hip attention_f8_kernel) ----
// q / k thirds: like the MXFP8 epilogue above (a 32x32 fragment = one scale block per token: 32 features of one head), rows leave
// head-major.  V third: V^T wants one scale per (feature, 32 KEYS): the wave's 32 tokens x 32 FN features are transposed through its
// LDS region as fp32 ([feature][token], 144-byte rows), then lane pair (2 f, 2 f + 1) owns feature f: 16 tokens each, block maximum
// by one exchange, 16 e4m3 bytes = one 16-byte store per lane.  A 32-token block never straddles utterances (Tp % 32 == 0).
template <int FN>
struct StagedQK8 {
    static constexpr int RS = 32 * 4 + 16;          // V path: 32 tokens of fp32 per feature row, padded
    static constexpr int BYTES = 32 * FN * RS;      // (the q / k path needs 32 x (32 FN + 16))
};

template <int FN>
__device__ __forceinline__ void epilogue_qk8_rows32(const GemmF8Args& a, const f32x16_t (&acc)[FN], const float4 (&bias)[FN][4], int mrow0,
                                                    int ncol0, char* lds, int lane) {
    const int ml = lane & 31, h = lane >> 5;
    // rows beyond the batch (M padded up to whole 256-row tiles so that the fp8 attention core does not depend on the batch shape:
    // ADVICE r4): computed from whatever follows the operand, never stored.  Wave-uniform (M_store % 32 == 0 as Tp % 32 == 0)
    if (a.g.M_store > 0 && mrow0 >= a.g.M_store) return;
    const int b = mrow0 / a.g.Tp, t0 = mrow0 - b * a.g.Tp;
    if (ncol0 >= 2 * SYL_HIDDEN) {
        // ---- V third
        using S = StagedQK8<FN>;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bb = bias[fn][g];
                const int f = 32 * fn + 8 * g + 4 * h;
                *(float*)(lds + (f + 0) * S::RS + ml * 4) = acc[fn][4 * g + 0] + bb.x;
                *(float*)(lds + (f + 1) * S::RS + ml * 4) = acc[fn][4 * g + 1] + bb.y;
                *(float*)(lds + (f + 2) * S::RS + ml * 4) = acc[fn][4 * g + 2] + bb.z;
                *(float*)(lds + (f + 3) * S::RS + ml * 4) = acc[fn][4 * g + 3] + bb.w;
            }
        const int vsp = a.g.Tpv / 32;
#pragma unroll
        for (int it = 0; it < FN; ++it) {
            const int idx = it * 64 + lane;
            const int f = idx >> 1, half = idx & 1;
            const float4* src = (const float4*)(lds + f * S::RS + half * 64);
            const float4 x0 = src[0], x1 = src[1], x2 = src[2], x3 = src[3];
            float amax = fmaxf(fmaxf(fmaxf(fabsf(x0.x), fabsf(x0.y)), fmaxf(fabsf(x0.z), fabsf(x0.w))),
                               fmaxf(fmaxf(fabsf(x1.x), fabsf(x1.y)), fmaxf(fabsf(x1.z), fabsf(x1.w))));
            amax = fmaxf(amax, fmaxf(fmaxf(fmaxf(fabsf(x2.x), fabsf(x2.y)), fmaxf(fabsf(x2.z), fabsf(x2.w))),
                                     fmaxf(fmaxf(fabsf(x3.x), fabsf(x3.y)), fmaxf(fabsf(x3.z), fabsf(x3.w)))));
            amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
            const unsigned e = mx_e8m0(amax);
            const float inv = mx_inv_scale(e);
            const int n = ncol0 + f - 2 * SYL_HIDDEN;                  // feature of v: head * 64 + d
            const size_t row = (size_t)b * SYL_HIDDEN + n;             // = (b * 12 + head) * 64 + d
            uint8_t* dst = (uint8_t*)a.g.out2 + row * a.g.Tpv + t0 + 16 * half;
            *(uint4*)dst = make_uint4(pack_fp8x4(x0.x * inv, x0.y * inv, x0.z * inv, x0.w * inv), pack_fp8x4(x1.x * inv, x1.y * inv, x1.z * inv, x1.w * inv),
                                      pack_fp8x4(x2.x * inv, x2.y * inv, x2.z * inv, x2.w * inv), pack_fp8x4(x3.x * inv, x3.y * inv, x3.z * inv, x3.w * inv));
            if (half == 0) a.vs[row * vsp + (t0 >> 5)] = (uint8_t)e;
            // the key tail [Tp, Tpv) (0 or 32 keys) is read by the attention kernel's last tile with P = 0: zero values, scale 1
            // (the region is shared with the FFN intermediate, so it is rewritten in every layer)
            if (t0 + 32 == a.g.Tp && a.g.Tpv > a.g.Tp) {
                *(uint4*)(dst + 32) = make_uint4(0u, 0u, 0u, 0u);
                if (half == 0) a.vs[row * vsp + (t0 >> 5) + 1] = (uint8_t)127;
            }
        }
        return;
    }
    // ---- q / k thirds
    constexpr int ROWB = 32 * FN, RS = ROWB + 16, CH = ROWB / 16;
    const int m = mrow0 + ml;
    const bool isk = ncol0 >= SYL_HIDDEN;
    const float qs = isk ? 1.0f : 0.125f;                              // q pre-scaled by 64^-0.5 (exact)
    uint8_t* sc = isk ? a.ks : a.qs;
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        float v[16];
        float amax = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bb = bias[fn][g];
            v[4 * g + 0] = (acc[fn][4 * g + 0] + bb.x) * qs; v[4 * g + 1] = (acc[fn][4 * g + 1] + bb.y) * qs;
            v[4 * g + 2] = (acc[fn][4 * g + 2] + bb.z) * qs; v[4 * g + 3] = (acc[fn][4 * g + 3] + bb.w) * qs;
#pragma unroll
            for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fabsf(v[4 * g + j]));
        }
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        const unsigned e = mx_e8m0(amax);
        const float inv = mx_inv_scale(e);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(unsigned*)(lds + ml * RS + 32 * fn + 8 * g + 4 * h) =
                pack_fp8x4(v[4 * g + 0] * inv, v[4 * g + 1] * inv, v[4 * g + 2] * inv, v[4 * g + 3] * inv);
        if (h == 0) {
            const int nn = ncol0 + 32 * fn - (isk ? SYL_HIDDEN : 0);
            sc[(((size_t)b * SYL_HEADS + (nn >> 6)) * a.g.Tp + t0 + ml) * 2 + ((nn & 63) >> 5)] = (uint8_t)e;
        }
    }
    (void)m;
#pragma unroll
    for (int it = 0; it < (CH + 1) / 2; ++it) {
        const int idx = it * 64 + lane;
        const int r = idx / CH, c = idx - r * CH;
        if (r >= 32) continue;
        const int nn = ncol0 + c * 16 - (isk ? SYL_HIDDEN : 0);
        uint8_t* dst = (uint8_t*)(isk ? a.g.out1 : a.g.out0) + (((size_t)b * SYL_HEADS + (nn >> 6)) * a.g.Tp + t0 + r) * 64 + (nn & 63);
        *(uint4*)dst = *(const uint4*)(lds + r * RS + c * 16);
    }
}
