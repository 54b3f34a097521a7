// Epilogue pieces shared by the bf16 and the MXFP8 GEMM kernels (swapped orientation: lane = token).
//
// Rule that shapes all of them (measured, profiles/r02_epilogue_ablation.md): an epilogue whose loads sit behind
// per-element runtime conditions compiles to a CHAIN of "branch, load, s_waitcnt vmcnt(0), compute, store" blocks --
// the first version of the residual + LayerNorm epilogue took 48 serialized memory round trips per wave (38 k of
// the out-projection's 67 k cycles).  So: every load of a wave's epilogue is issued unconditionally from CLAMPED
// addresses, in batches (column vectors once per wave, residual rows once per fragment column), and only the stores
// are predicated.
#pragma once
#include "kernels.h"

template <int ACT>
__device__ __forceinline__ float apply_act(float v) {
    if constexpr (ACT == ACT_GELU_FAST) return gelu_fast(v);
    if constexpr (ACT == ACT_GELU_ERF) return gelu_erf(v);
    if constexpr (ACT == ACT_GELU_ERF7) return gelu_erf7(v);
    return v;
}
// four values of one run: the fast GELU goes through the packed-fp32 pipe two at a time
template <int ACT>
__device__ __forceinline__ void apply_act4(float& v0, float& v1, float& v2, float& v3) {
    if constexpr (ACT == ACT_GELU_FAST) { gelu_fast2(v0, v1); gelu_fast2(v2, v3); }
    else { v0 = apply_act<ACT>(v0); v1 = apply_act<ACT>(v1); v2 = apply_act<ACT>(v2); v3 = apply_act<ACT>(v3); }
}

// A wave's FN 32x32 fragments in swapped orientation: lane (ml = lane & 31, h = lane >> 5) owns token row ml and,
// per fragment fn, the 16 columns ncol0 + 32 fn + 8 g + 4 h + e (g, e < 4): four runs of 4 consecutive columns.
// Column vector p[n .. n+3] for every (fn, g) of this lane; columns beyond N are clamped (their stores are predicated
// off); p == nullptr gives zeros.  N % 4 == 0 and N >= 4.
template <int FN>
__device__ __forceinline__ void load_colvec(const float* __restrict__ p, int ncol0, int h, int N, float4 (&v)[FN][4]) {
    if (p == nullptr) {
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
            for (int g = 0; g < 4; ++g) v[fn][g] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int n = ncol0 + 32 * fn + 8 * g + 4 * h;
            n = n < N - 4 ? n : N - 4;
            v[fn][g] = *(const float4*)(p + n);
        }
}

// Direct (register -> global) epilogue of a whole wave tile, FM x FN fragments: EPI_BF16 / EPI_F32 / EPI_F32_RES /
// EPI_F32_RESLN.  A store instruction covers 32 rows x 32 bytes; loads of the residual likewise.
template <int FM, int FN, int EPI, int ACT, int FMT = FMT_BF16>
__device__ __forceinline__ void epilogue_direct(const GemmArgs& a, const f32x16_t (&acc)[FM][FN], int mrow0, int ncol0, int lane) {
    static_assert(EPI == EPI_BF16 || EPI == EPI_F32 || EPI == EPI_F32_RES || EPI == EPI_F32_RESLN, "direct epilogue");
    const int ml = lane & 31, h = lane >> 5;
    int mrow[FM];                                    // clamped row for loads
    bool mok[FM];
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int m = mrow0 + 32 * fm + ml;
        mok[fm] = m < a.M;
        mrow[fm] = mok[fm] ? m : a.M - 1;
    }
    float2 st[FM];
    if constexpr (EPI == EPI_F32_RESLN) {
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) st[fm] = *(const float2*)(a.ln_stats + (size_t)mrow[fm] * 2);
    }
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        int ncl[4]; bool nok[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = ncol0 + 32 * fn + 8 * g + 4 * h;
            nok[g] = n < a.N;
            ncl[g] = n < a.N - 4 ? n : a.N - 4;
        }
        // this fragment column's loads, all issued before the first use: bias / gamma / beta runs and the residual rows
        float4 bias[1][4], gg[4], be[4], rr[FM][4];
        load_colvec<1>(a.bias, ncol0 + 32 * fn, h, a.N, bias);
        if constexpr (EPI == EPI_F32_RESLN) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { gg[g] = *(const float4*)(a.ln_gamma + ncl[g]); be[g] = *(const float4*)(a.ln_beta + ncl[g]); }
        }
        if constexpr (EPI == EPI_F32_RES || EPI == EPI_F32_RESLN) {
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                for (int g = 0; g < 4; ++g) rr[fm][g] = *(const float4*)(a.res + (size_t)mrow[fm] * a.ldres + ncl[g]);
        }
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = acc[fm][fn][4 * g + 0] + bias[0][g].x, v1 = acc[fm][fn][4 * g + 1] + bias[0][g].y;
                float v2 = acc[fm][fn][4 * g + 2] + bias[0][g].z, v3 = acc[fm][fn][4 * g + 3] + bias[0][g].w;
                const bool ok = mok[fm] && nok[g];
                const size_t o = (size_t)mrow[fm] * a.ld0 + ncl[g];
                if constexpr (EPI == EPI_BF16) {
                    apply_act4<ACT>(v0, v1, v2, v3);
                    uint2 pk; pk.x = H16<FMT>::pack2(v0, v1); pk.y = H16<FMT>::pack2(v2, v3);
                    if (ok) *(uint2*)((bf16_t*)a.out0 + o) = pk;
                    if constexpr (FMT == FMT_SPLIT) {
                        uint2 lo; lo.x = H16<FMT>::pack2_lo(v0, v1, pk.x); lo.y = H16<FMT>::pack2_lo(v2, v3, pk.y);
                        if (ok) *(uint2*)((bf16_t*)a.out0 + a.out_lo + o) = lo;
                    }
                } else if constexpr (EPI == EPI_F32) {
                    apply_act4<ACT>(v0, v1, v2, v3);
                    if (ok) *(float4*)((float*)a.out0 + o) = make_float4(v0, v1, v2, v3);
                } else if constexpr (EPI == EPI_F32_RES) {
                    const float4 r = rr[fm][g];
                    if (ok) *(float4*)((float*)a.out0 + o) = make_float4(v0 + r.x, v1 + r.y, v2 + r.z, v3 + r.w);
                } else {
                    // residual = LayerNorm(res row) re-applied here with the row statistics the LayerNorm kernel left
                    // behind (same expression as layernorm_kernel -> bitwise the value it would have stored)
                    const float4 r = rr[fm][g];
                    const float mean = st[fm].x, rstd = st[fm].y;
                    const float h0 = fmaf((r.x - mean) * rstd, gg[g].x, be[g].x), h1 = fmaf((r.y - mean) * rstd, gg[g].y, be[g].y);
                    const float h2 = fmaf((r.z - mean) * rstd, gg[g].z, be[g].z), h3 = fmaf((r.w - mean) * rstd, gg[g].w, be[g].w);
                    if (ok) *(float4*)((float*)a.out0 + o) = make_float4(v0 + h0, v1 + h1, v2 + h2, v3 + h3);
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------
// Coalesced epilogue through LDS (swapped orientation).  Straight from the MFMA layout a lane owns ONE
// output row and runs of 4 columns, so a store instruction would touch 32 rows with 16-32 bytes each.
// Instead every wave transposes one 32-row block of its tile through a private LDS region (row stride padded
// by 16 B: 2-way worst case on ds_write_b64) and reads it back as 16-byte chunks with consecutive lanes on
// consecutive chunks of a row, so global loads (fp32 residual) and stores run over whole 128-byte lines.
// Bias / activation / q-scaling / padded frame zeroing are applied on the way in; the residual add on the way out.
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
// PLANE = -1: acc + bias, one plane.  FMT_SPLIT: acc already holds the final values (bias is zero); PLANE 0 stores their hi
// halves, PLANE 1 the lo halves into the plane a.out_lo elements further on.
template <int FN, int FMT = FMT_BF16, int PLANE = -1>
__device__ __forceinline__ void epilogue_vt_rows32(const GemmArgs& a, const f32x16_t (&acc)[FN], const float4 (&bias)[FN][4], int mrow0,
                                                   int ncol0, char* lds, int lane) {
    const int ml = lane & 31, h = lane >> 5;
    const int pos = (ml & ~12) | ((ml & 4) << 1) | ((ml & 8) >> 1);
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = 32 * fn + 8 * g + 4 * h;
            const float4 bb = bias[fn][g];
            if constexpr (PLANE != 1) {
                *(bf16_t*)(lds + (nl + 0) * 64 + pos * 2) = H16<FMT>::cvt(acc[fn][4 * g + 0] + bb.x);
                *(bf16_t*)(lds + (nl + 1) * 64 + pos * 2) = H16<FMT>::cvt(acc[fn][4 * g + 1] + bb.y);
                *(bf16_t*)(lds + (nl + 2) * 64 + pos * 2) = H16<FMT>::cvt(acc[fn][4 * g + 2] + bb.z);
                *(bf16_t*)(lds + (nl + 3) * 64 + pos * 2) = H16<FMT>::cvt(acc[fn][4 * g + 3] + bb.w);
            } else {
                const float v[4] = {acc[fn][4 * g + 0] + bb.x, acc[fn][4 * g + 1] + bb.y, acc[fn][4 * g + 2] + bb.z, acc[fn][4 * g + 3] + bb.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) *(bf16_t*)(lds + (nl + e) * 64 + pos * 2) = H16<FMT_SPLIT>::cvt_lo(v[e], H16<FMT>::cvt(v[e]));
            }
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
        bf16_t* dst = (bf16_t*)a.out2 + (PLANE == 1 ? a.out2_lo : 0L) + ((size_t)b * SYL_HIDDEN + (n - 2 * SYL_HIDDEN)) * a.Tpv + t0 + 8 * c;
        *(uint4*)dst = raw;
        // the key tail [Tp, Tpv) (0 or 32 keys: Tp % 32 == 0, Tpv % 64 == 0) is read by the attention kernel's last
        // tile with P = 0; it must stay finite although the region is shared with the FFN intermediate
        if (t0 + 32 == a.Tp && a.Tpv > a.Tp) *(uint4*)(dst + 32) = make_uint4(0u, 0u, 0u, 0u);
    }
}

// one 32-row block of a wave's tile; `bias` = load_colvec(a.bias, ncol0, ...) of the wave (loaded once per tile)
template <int FN, int EPI, int ACT, int FMT = FMT_BF16, int PLANE = -1>
__device__ __forceinline__ void epilogue_rows32(const GemmArgs& a, const f32x16_t (&acc)[FN], const float4 (&bias)[FN][4], int mrow0,
                                                int ncol0, char* lds, int lane) {
    using S = StagedEpi<FN, EPI>;
    const long plane_off = PLANE == 1 ? a.out_lo : 0L;
    if constexpr (EPI == EPI_QK) {
        // one launch for q, k and v (N = 2304): the V third leaves through the transposing epilogue (wave-uniform:
        // every wave's column range lies inside one third, 768 being a multiple of every wave width in use)
        static_assert(FN * 32 * 64 <= S::BYTES, "V^T staging must fit the wave's region");
        if (ncol0 >= 2 * SYL_HIDDEN) { epilogue_vt_rows32<FN, FMT, PLANE>(a, acc, bias, mrow0, ncol0, lds, lane); return; }
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
    const float qs = (EPI == EPI_QK && ncol0 < SYL_HIDDEN && PLANE < 0) ? SYL_Q_SCALE : 1.0f;     // q third pre-scaled by 64^-0.5 log2(e)
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = 32 * fn + 8 * g + 4 * h;
            const float4 bb = bias[fn][g];
            float v0 = acc[fn][4 * g + 0] + bb.x, v1 = acc[fn][4 * g + 1] + bb.y, v2 = acc[fn][4 * g + 2] + bb.z, v3 = acc[fn][4 * g + 3] + bb.w;
            if constexpr ((EPI == EPI_BF16 || EPI == EPI_F32) && PLANE < 0) {
                apply_act4<ACT>(v0, v1, v2, v3);
            }
            if constexpr (EPI == EPI_QK) { v0 *= qs; v1 *= qs; v2 *= qs; v3 *= qs; }
            if constexpr (EPI == EPI_PROJ) { if (zero_row) { v0 = v1 = v2 = v3 = 0.f; } }
            if constexpr (S::F32OUT) {
                *(float4*)(lds + ml * S::RS + nl * 4) = make_float4(v0, v1, v2, v3);
            } else {
                uint2 pk; pk.x = H16<FMT>::pack2(v0, v1); pk.y = H16<FMT>::pack2(v2, v3);
                if constexpr (PLANE == 1) { pk.x = H16<FMT_SPLIT>::pack2_lo(v0, v1, pk.x); pk.y = H16<FMT_SPLIT>::pack2_lo(v2, v3, pk.y); }
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
            *(uint4*)((bf16_t*)a.out0 + plane_off + (size_t)mo * a.ld0 + n) = raw;
        } else if constexpr (EPI == EPI_F32) {
            *(uint4*)((float*)a.out0 + (size_t)mo * a.ld0 + n) = raw;
        } else if constexpr (EPI == EPI_QK) {
            const int which = n >= SYL_HIDDEN;
            const int nn = n - which * SYL_HIDDEN;
            const int head = nn >> 6, d = nn & 63;
            const int b = mo / a.Tp, t = mo - b * a.Tp;
            *(uint4*)((bf16_t*)(which ? a.out1 : a.out0) + plane_off + (((size_t)b * SYL_HEADS + head) * a.Tp + t) * 64 + d) = raw;
        } else if constexpr (EPI == EPI_PROJ) {
            const float4 v = __builtin_bit_cast(float4, raw);
            *(float4*)((float*)a.out0 + (size_t)mo * a.ld0 + n) = v;
            const int b = mo / a.Tp, t = mo - b * a.Tp;
            uint2 pk; pk.x = H16<FMT>::pack2(v.x, v.y); pk.y = H16<FMT>::pack2(v.z, v.w);
            *(uint2*)((bf16_t*)a.out1 + ((size_t)b * a.xpad_rows + 64 + t) * SYL_HIDDEN + n) = pk;
            if constexpr (FMT == FMT_SPLIT) {
                uint2 lo; lo.x = H16<FMT>::pack2_lo(v.x, v.y, pk.x); lo.y = H16<FMT>::pack2_lo(v.z, v.w, pk.y);
                *(uint2*)((bf16_t*)a.out1 + a.out_lo + ((size_t)b * a.xpad_rows + 64 + t) * SYL_HIDDEN + n) = lo;
            }
        }
    }
}

// all FM 32-row blocks of a wave's tile through the staged epilogue (bias column vectors loaded once)
template <int FM, int FN, int EPI, int ACT, int FMT = FMT_BF16>
__device__ __forceinline__ void epilogue_staged(const GemmArgs& a, const f32x16_t (&acc)[FM][FN], int mrow0, int ncol0, char* lds, int lane) {
    float4 bias[FN][4];
    load_colvec<FN>(a.bias, ncol0, lane >> 5, a.N, bias);
    if constexpr (FMT == FMT_SPLIT && !StagedEpi<FN, EPI>::F32OUT) {
        // two planes: the final fp32 values (bias, activation, q scaling) are formed ONCE, in place of the accumulators,
        // then their hi halves and their lo halves go through the wave's staging region one after the other (LDS
        // operations of one wave execute in order, so pass 1 may overwrite what pass 0 has read)
        f32x16_t v[FM][FN];
        const float qs = (EPI == EPI_QK && ncol0 < SYL_HIDDEN) ? SYL_Q_SCALE : 1.0f;
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v0 = acc[fm][fn][4 * g + 0] + bias[fn][g].x, v1 = acc[fm][fn][4 * g + 1] + bias[fn][g].y;
                    float v2 = acc[fm][fn][4 * g + 2] + bias[fn][g].z, v3 = acc[fm][fn][4 * g + 3] + bias[fn][g].w;
                    if constexpr (EPI == EPI_BF16) apply_act4<ACT>(v0, v1, v2, v3);
                    v[fm][fn][4 * g + 0] = v0 * qs; v[fm][fn][4 * g + 1] = v1 * qs; v[fm][fn][4 * g + 2] = v2 * qs; v[fm][fn][4 * g + 3] = v3 * qs;
                }
        float4 zero[FN][4];
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
            for (int g = 0; g < 4; ++g) zero[fn][g] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) epilogue_rows32<FN, EPI, ACT, FMT, 0>(a, v[fm], zero, mrow0 + fm * 32, ncol0, lds, lane);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) epilogue_rows32<FN, EPI, ACT, FMT, 1>(a, v[fm], zero, mrow0 + fm * 32, ncol0, lds, lane);
    } else {
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) epilogue_rows32<FN, EPI, ACT, FMT>(a, acc[fm], bias, mrow0 + fm * 32, ncol0, lds, lane);
    }
}
