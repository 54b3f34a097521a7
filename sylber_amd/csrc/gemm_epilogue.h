// Epilogue pieces shared by the bf16 and the MXFP8 GEMM kernels (swapped orientation: lane = token).
#pragma once
#include "kernels.h"

template <int ACT>
__device__ __forceinline__ float apply_act(float v) {
    if constexpr (ACT == 1) return gelu_fast(v);
    if constexpr (ACT == 2) return gelu_erf(v);
    return v;
}
// four values of one run: the fast GELU goes through the packed-fp32 pipe two at a time
template <int ACT>
__device__ __forceinline__ void apply_act4(float& v0, float& v1, float& v2, float& v3) {
    if constexpr (ACT == 1) { gelu_fast2(v0, v1); gelu_fast2(v2, v3); }
    else { v0 = apply_act<ACT>(v0); v1 = apply_act<ACT>(v1); v2 = apply_act<ACT>(v2); v3 = apply_act<ACT>(v3); }
}

// Epilogue for one 32x32 fragment in SWAPPED orientation: lane owns token m (column l&31) and 16
// output features n = nb + (r&3) + 8*(r>>2) + 4*(l>>5): four runs of 4 consecutive n.
template <int EPI, int ACT>
__device__ __forceinline__ void epilogue_swapped(const GemmArgs& a, const f32x16_t& acc, int m, int nb, int lane) {
    if (m >= a.M) return;
    const int h = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = nb + 8 * g + 4 * h;
        if (n >= a.N) continue;
        float v0 = acc[4 * g + 0], v1 = acc[4 * g + 1], v2 = acc[4 * g + 2], v3 = acc[4 * g + 3];
        if (a.bias) {
            const float4 bb = *(const float4*)(a.bias + n);
            v0 += bb.x; v1 += bb.y; v2 += bb.z; v3 += bb.w;
        }
        if constexpr (EPI == EPI_BF16) {
            apply_act4<ACT>(v0, v1, v2, v3);
            uint2 pk; pk.x = pack_bf16x2(v0, v1); pk.y = pack_bf16x2(v2, v3);
            *(uint2*)((bf16_t*)a.out0 + (size_t)m * a.ld0 + n) = pk;
        } else if constexpr (EPI == EPI_F32) {
            apply_act4<ACT>(v0, v1, v2, v3);
            *(float4*)((float*)a.out0 + (size_t)m * a.ld0 + n) = make_float4(v0, v1, v2, v3);
        } else if constexpr (EPI == EPI_F32_RES) {
            const float4 rr = *(const float4*)(a.res + (size_t)m * a.ldres + n);
            *(float4*)((float*)a.out0 + (size_t)m * a.ld0 + n) = make_float4(v0 + rr.x, v1 + rr.y, v2 + rr.z, v3 + rr.w);
        } else if constexpr (EPI == EPI_F32_RESLN) {
            // residual = LayerNorm(res row) re-applied here with the row statistics the LayerNorm kernel left
            // behind (same expression as layernorm_kernel -> bitwise the value it would have stored)
            const float4 rr = *(const float4*)(a.res + (size_t)m * a.ldres + n);
            const float2 st = *(const float2*)(a.ln_stats + (size_t)m * 2);
            const float4 gg = *(const float4*)(a.ln_gamma + n);
            const float4 be = *(const float4*)(a.ln_beta + n);
            const float h0 = fmaf((rr.x - st.x) * st.y, gg.x, be.x), h1 = fmaf((rr.y - st.x) * st.y, gg.y, be.y);
            const float h2 = fmaf((rr.z - st.x) * st.y, gg.z, be.z), h3 = fmaf((rr.w - st.x) * st.y, gg.w, be.w);
            *(float4*)((float*)a.out0 + (size_t)m * a.ld0 + n) = make_float4(v0 + h0, v1 + h1, v2 + h2, v3 + h3);
        } else if constexpr (EPI == EPI_QK) {
            // n < 1536 here (q and k thirds); head-major [B,H,Tp,64]
            const int which = n >= SYL_HIDDEN;            // 0 = q, 1 = k
            const int nn = n - which * SYL_HIDDEN;
            const int head = nn >> 6, d = nn & 63;
            const int b = m / a.Tp, t = m - b * a.Tp;
            if (!which) { v0 *= 0.125f; v1 *= 0.125f; v2 *= 0.125f; v3 *= 0.125f; }
            bf16_t* dst = (bf16_t*)(which ? a.out1 : a.out0) + (((size_t)b * SYL_HEADS + head) * a.Tp + t) * 64 + d;
            uint2 pk; pk.x = pack_bf16x2(v0, v1); pk.y = pack_bf16x2(v2, v3);
            *(uint2*)dst = pk;
        } else if constexpr (EPI == EPI_PROJ) {
            const int b = m / a.Tp, t = m - b * a.Tp;
            const int nv = a.valid[b] < a.T ? a.valid[b] : a.T;
            if (t >= nv) { v0 = v1 = v2 = v3 = 0.f; }   // TP:428-431 zero padded frames (and rows beyond T)
            *(float4*)((float*)a.out0 + (size_t)m * a.ld0 + n) = make_float4(v0, v1, v2, v3);
            uint2 pk; pk.x = pack_bf16x2(v0, v1); pk.y = pack_bf16x2(v2, v3);
            *(uint2*)((bf16_t*)a.out1 + ((size_t)b * a.xpad_rows + 64 + t) * SYL_HIDDEN + n) = pk;
        }
    }
}

