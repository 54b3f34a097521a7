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

