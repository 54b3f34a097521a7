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

// ---------------------------------------------------------------------------------------------------
// Coalesced epilogue through LDS (swapped orientation).  Straight from the MFMA layout a lane owns ONE
// output row and runs of 4 columns, so a store instruction would touch 32 rows with 16-32 bytes each
// (measured: 20-48 % of the kernel on the hot-path shapes).  Instead every wave transposes one 32-row block
// of its tile through a private LDS region (row stride padded by 16 B: 2-way worst case on ds_write_b64)
// and reads it back as 16-byte chunks with consecutive lanes on consecutive chunks of a row, so global
// loads (fp32 residual) and stores run over whole 128-byte lines.  Bias / activation / q-scaling / padded
// frame zeroing are applied on the way in; the residual add on the way out.
template <int FN, int EPI>
struct StagedEpi {
    static constexpr bool F32OUT = (EPI == EPI_F32 || EPI == EPI_F32_RES || EPI == EPI_F32_RESLN || EPI == EPI_PROJ);
    static constexpr int ES = F32OUT ? 4 : 2;
    static constexpr int ROWB = 32 * FN * ES;       // payload bytes per row
    static constexpr int RS = ROWB + 16;            // padded row stride
    static constexpr int CH = ROWB / 16;            // 16-byte chunks per row
    static constexpr int BYTES = 32 * RS;           // private LDS bytes per wave
};

// V third of the fused q/k/v projection (EPI_QK, columns n >= 1536), swapped orientation like q and k: the lane
// owns token mrow0 + (lane & 31) and 16 features per fragment, but V^T wants [b][feature][key] with the key axis
// contiguous (and bits 2/3 of the key index swapped, see attention.hip).  The wave transposes its 32 tokens x 32 FN
// features through its private LDS region (ds_write_b16 at [feature][pos(token)], 64-byte rows) and stores 16-byte
// chunks = 8 keys of one feature row; a 32-token block never straddles utterances because Tp % 32 == 0.
template <int FN>
__device__ __forceinline__ void epilogue_vt_rows32(const GemmArgs& a, const f32x16_t (&acc)[FN], int mrow0, int ncol0, char* lds,
                                                   int lane) {
    const int ml = lane & 31, h = lane >> 5;
    const int pos = (ml & ~12) | ((ml & 4) << 1) | ((ml & 8) >> 1);
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = 32 * fn + 8 * g + 4 * h;
            const int n = ncol0 + nl;
            float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias && n < a.N) bb = *(const float4*)(a.bias + n);
            *(bf16_t*)(lds + (nl + 0) * 64 + pos * 2) = f2bf_dev(acc[fn][4 * g + 0] + bb.x);
            *(bf16_t*)(lds + (nl + 1) * 64 + pos * 2) = f2bf_dev(acc[fn][4 * g + 1] + bb.y);
            *(bf16_t*)(lds + (nl + 2) * 64 + pos * 2) = f2bf_dev(acc[fn][4 * g + 2] + bb.z);
            *(bf16_t*)(lds + (nl + 3) * 64 + pos * 2) = f2bf_dev(acc[fn][4 * g + 3] + bb.w);
        }
    if (mrow0 >= a.M) return;
    const int b = mrow0 / a.Tp, t0 = mrow0 - b * a.Tp;
#pragma unroll
    for (int it = 0; it < FN * 2; ++it) {
        const int idx = it * 64 + lane;
        const int f = idx >> 2, c = idx & 3;
        const int n = ncol0 + f;
        if (n >= a.N) continue;
        const uint4 raw = *(const uint4*)(lds + f * 64 + c * 16);
        *(uint4*)((bf16_t*)a.out2 + ((size_t)b * SYL_HIDDEN + (n - 2 * SYL_HIDDEN)) * a.Tpv + t0 + 8 * c) = raw;
    }
}

template <int FN, int EPI, int ACT>
__device__ __forceinline__ void epilogue_rows32(const GemmArgs& a, const f32x16_t (&acc)[FN], int mrow0, int ncol0, char* lds,
                                                int lane) {
    using S = StagedEpi<FN, EPI>;
    if constexpr (EPI == EPI_QK) {
        // one launch for q, k and v (N = 2304): the V third leaves through the transposing epilogue (wave-uniform:
        // every wave's column range lies inside one third, 768 being a multiple of every wave width in use)
        static_assert(FN * 32 * 64 <= S::BYTES, "V^T staging must fit the wave's region");
        if (ncol0 >= 2 * SYL_HIDDEN) { epilogue_vt_rows32<FN>(a, acc, mrow0, ncol0, lds, lane); return; }
    }
    const int ml = lane & 31, h = lane >> 5;
    const int m = mrow0 + ml;
    // ---- in: MFMA layout -> row-major LDS
    bool zero_row = false;
    if constexpr (EPI == EPI_PROJ) {
        const int mm = m < a.M ? m : a.M - 1;
        const int b = mm / a.Tp, t = mm - b * a.Tp;
        const int nv = a.valid[b] < a.T ? a.valid[b] : a.T;
        zero_row = t >= nv;                              // TP:428-431 zero padded frames (and rows beyond T)
    }
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = 32 * fn + 8 * g + 4 * h;
            const int n = ncol0 + nl;
            float v0 = acc[fn][4 * g + 0], v1 = acc[fn][4 * g + 1], v2 = acc[fn][4 * g + 2], v3 = acc[fn][4 * g + 3];
            if (a.bias && n < a.N) {
                const float4 bb = *(const float4*)(a.bias + n);
                v0 += bb.x; v1 += bb.y; v2 += bb.z; v3 += bb.w;
            }
            if constexpr (EPI == EPI_BF16 || EPI == EPI_F32) {
                apply_act4<ACT>(v0, v1, v2, v3);
            }
            if constexpr (EPI == EPI_QK) {
                if (n < SYL_HIDDEN) { v0 *= 0.125f; v1 *= 0.125f; v2 *= 0.125f; v3 *= 0.125f; }
            }
            if constexpr (EPI == EPI_PROJ) { if (zero_row) { v0 = v1 = v2 = v3 = 0.f; } }
            if constexpr (S::F32OUT) {
                *(float4*)(lds + ml * S::RS + nl * 4) = make_float4(v0, v1, v2, v3);
            } else {
                uint2 pk; pk.x = pack_bf16x2(v0, v1); pk.y = pack_bf16x2(v2, v3);
                *(uint2*)(lds + ml * S::RS + nl * 2) = pk;
            }
        }
    // ---- out: 16-byte chunks, consecutive lanes on consecutive chunks of a row
#pragma unroll
    for (int it = 0; it < S::CH / 2; ++it) {
        const int idx = it * 64 + lane;
        const int r = idx / S::CH, c = idx - r * S::CH;
        const int mo = mrow0 + r;
        const int n = ncol0 + c * (16 / S::ES);
        if (mo >= a.M || n >= a.N) continue;
        const uint4 raw = *(const uint4*)(lds + r * S::RS + c * 16);
        if constexpr (EPI == EPI_BF16) {
            *(uint4*)((bf16_t*)a.out0 + (size_t)mo * a.ld0 + n) = raw;
        } else if constexpr (EPI == EPI_F32) {
            *(uint4*)((float*)a.out0 + (size_t)mo * a.ld0 + n) = raw;
        } else if constexpr (EPI == EPI_F32_RES) {
            const float4 rr = *(const float4*)(a.res + (size_t)mo * a.ldres + n);
            const float4 v = __builtin_bit_cast(float4, raw);
            *(float4*)((float*)a.out0 + (size_t)mo * a.ld0 + n) = make_float4(v.x + rr.x, v.y + rr.y, v.z + rr.z, v.w + rr.w);
        } else if constexpr (EPI == EPI_QK) {
            const int which = n >= SYL_HIDDEN;
            const int nn = n - which * SYL_HIDDEN;
            const int head = nn >> 6, d = nn & 63;
            const int b = mo / a.Tp, t = mo - b * a.Tp;
            *(uint4*)((bf16_t*)(which ? a.out1 : a.out0) + (((size_t)b * SYL_HEADS + head) * a.Tp + t) * 64 + d) = raw;
        } else if constexpr (EPI == EPI_PROJ) {
            const float4 v = __builtin_bit_cast(float4, raw);
            *(float4*)((float*)a.out0 + (size_t)mo * a.ld0 + n) = v;
            const int b = mo / a.Tp, t = mo - b * a.Tp;
            uint2 pk; pk.x = pack_bf16x2(v.x, v.y); pk.y = pack_bf16x2(v.z, v.w);
            *(uint2*)((bf16_t*)a.out1 + ((size_t)b * a.xpad_rows + 64 + t) * SYL_HIDDEN + n) = pk;
        }
    }
}

