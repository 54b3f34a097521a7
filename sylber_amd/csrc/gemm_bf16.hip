// bf16 MFMA GEMM for gfx950:  out[m][n] = sum_k X[m][k] * W[n][k]  with fused epilogues.
//
// Covers the dense contractions of the Segmenter hot path (reference call site
// sylber/model/sylber.py:122 -> transformers HubertModel): the six 512->512 strided Conv1d layers
// (TP:112-124, as implicit GEMM on channels-last activations: ldx = stride*512, K = taps*512), the
// feature projection (TP:225-231), q/k/v/out projections (TP:318-342) and the FFN (TP:361-368).
//
// Structure: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave =
// 2x2 v_mfma_f32_32x32x16_bf16 fragments, 64 fp32 accumulators/lane), K step 64.  Both operands are
// staged HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip) into a double buffer; the LDS
// image is lane-linear per wave instruction (8 rows x 128 B), so the bank-conflict swizzle
// (16-B chunk ^= (row>>1)&7, conflict-free for ds_read_b128's 16-lane groups on 128-B rows) is
// applied to the per-lane SOURCE address and again on the fragment read.  One barrier per K tile:
// the loads of tile t+1 are issued before the MFMAs of tile t and land under them.
// Workgroup ids are remapped so that each XCD owns a contiguous run of tiles (shared X panel in L2).
#include "kernels.h"

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (128 * BK * 2)          // 16 KiB per operand tile
#define STAGE_BYTES (2 * TILE_BYTES)       // X tile + W tile
#define GEMM_LDS (2 * STAGE_BYTES)         // double buffered: 64 KiB

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((glb_vptr)g, (lds_vptr)l, 16, 0, 0);
}

template <int ACT>
__device__ __forceinline__ float apply_act(float v) {
    if constexpr (ACT == 1) return gelu_fast(v);
    if constexpr (ACT == 2) return gelu_erf(v);
    return v;
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
            v0 = apply_act<ACT>(v0); v1 = apply_act<ACT>(v1);
            v2 = apply_act<ACT>(v2); v3 = apply_act<ACT>(v3);
            uint2 pk; pk.x = pack_bf16x2(v0, v1); pk.y = pack_bf16x2(v2, v3);
            *(uint2*)((bf16_t*)a.out0 + (size_t)m * a.ld0 + n) = pk;
        } else if constexpr (EPI == EPI_F32) {
            v0 = apply_act<ACT>(v0); v1 = apply_act<ACT>(v1);
            v2 = apply_act<ACT>(v2); v3 = apply_act<ACT>(v3);
            *(float4*)((float*)a.out0 + (size_t)m * a.ld0 + n) = make_float4(v0, v1, v2, v3);
        } else if constexpr (EPI == EPI_F32_RES) {
            const float4 rr = *(const float4*)(a.res + (size_t)m * a.ldres + n);
            *(float4*)((float*)a.out0 + (size_t)m * a.ld0 + n) = make_float4(v0 + rr.x, v1 + rr.y, v2 + rr.z, v3 + rr.w);
        } else if constexpr (EPI == EPI_QKV) {
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

// V third of the fused QKV projection, NATURAL orientation: lane owns feature n (column l&31) and 16
// tokens m = mb + (r&3) + 8*(r>>2) + 4*(l>>5): four runs of 4 consecutive tokens -> 8-byte stores
// into the key-contiguous Vt[b][head][d][t] image the attention kernel reads as MFMA A-operand.
__device__ __forceinline__ void epilogue_v_natural(const GemmArgs& a, const f32x16_t& acc, int mb, int n, int lane) {
    if (n >= a.N) return;
    const int h = lane >> 5;
    const float bias = a.bias ? a.bias[n] : 0.f;
    const int nn = n - 2 * SYL_HIDDEN;
    const int head = nn >> 6, d = nn & 63;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int m = mb + 8 * g + 4 * h;       // multiple of 4; Tp % 4 == 0 so the run stays in one utterance
        if (m >= a.M) continue;
        const int b = m / a.Tp, t = m - b * a.Tp;
        // key axis stored with bits 2 and 3 swapped: a 16-B chunk then holds exactly the 8 keys one
        // half-wave contributes to a 16-key P.V MFMA (see attention.hip)
        const int pos = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1);
        uint2 pk;
        pk.x = pack_bf16x2(acc[4 * g + 0] + bias, acc[4 * g + 1] + bias);
        pk.y = pack_bf16x2(acc[4 * g + 2] + bias, acc[4 * g + 3] + bias);
        *(uint2*)((bf16_t*)a.out2 + (((size_t)b * SYL_HEADS + head) * 64 + d) * a.Tpv + pos) = pk;
    }
}

template <int EPI, int ACT>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = (a.N + BN - 1) / BN;
    const int tiles_m = (a.M + BM - 1) / BM;
    const int wg = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (wg / tiles_n) * BM;
    const int n0 = (wg % tiles_n) * BN;

    // ---- staging addresses: wave w fills rows [32w, 32w+32) of both tiles, 8 rows per instruction
    const int srow = lane >> 3;                 // row within the 8-row piece
    const int spos = lane & 7;                  // 16-B position within the 128-B LDS row
    const bf16_t* gx[4];
    const bf16_t* gw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wave * 32 + i * 8 + srow;
        const int c = spos ^ ((r >> 1) & 7);    // source chunk that must land at LDS position spos
        int xm = m0 + r; xm = xm < a.M ? xm : a.M - 1;
        int wr = n0 + r; wr = wr < a.N ? wr : a.N - 1;
        gx[i] = a.X + (size_t)xm * a.ldx + c * 8;
        gw[i] = a.W + (size_t)wr * a.K + c * 8;
    }
    const int lds_piece = (wave * 32) * 128;    // byte offset of this wave's first piece in a tile

    // ---- fragment read addresses (bytes within a tile)
    const int frow = lane & 31;
    const int swz = (lane >> 1) & 7;            // == ((row >> 1) & 7) for row = 32*j + (lane & 31)
    const int fhalf = lane >> 5;
    int koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = (((2 * kk + fhalf) ^ swz) << 4);
    const int xrow_off = (wm * 64 + frow) * 128;
    const int wrow_off = (wn * 64 + frow) * 128;

    // V third of the QKV GEMM runs in natural orientation (block-uniform)
    const bool natural = (EPI == EPI_QKV) && (n0 >= 2 * SYL_HIDDEN);

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = a.K / BK;
    // prologue: stage tile 0 into buffer 0
    {
        char* xb = smem;
        char* wb = smem + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(gx[i], xb + lds_piece + i * 1024);
            glds16(gw[i], wb + lds_piece + i * 1024);
        }
    }
    for (int t = 0; t < nt; ++t) {
        __syncthreads();   // tile t landed (vmcnt(0) precedes the barrier); everyone is done with the other buffer
        if (t + 1 < nt) {
            char* xb = smem + ((t + 1) & 1) * STAGE_BYTES;
            char* wb = xb + TILE_BYTES;
            const int ko = (t + 1) * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                glds16(gx[i] + ko, xb + lds_piece + i * 1024);
                glds16(gw[i] + ko, wb + lds_piece + i * 1024);
            }
        }
        const char* xb = smem + (t & 1) * STAGE_BYTES;
        const char* wb = xb + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8_t xf[2], wf[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                xf[f] = *(const bf16x8_t*)(xb + xrow_off + f * 32 * 128 + koff[kk]);
                wf[f] = *(const bf16x8_t*)(wb + wrow_off + f * 32 * 128 + koff[kk]);
            }
            if (natural) {
#pragma unroll
                for (int fm = 0; fm < 2; ++fm)
#pragma unroll
                    for (int fn = 0; fn < 2; ++fn)
                        acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[fm], wf[fn], acc[fm][fn], 0, 0, 0);
            } else {
#pragma unroll
                for (int fm = 0; fm < 2; ++fm)
#pragma unroll
                    for (int fn = 0; fn < 2; ++fn)
                        acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[fn], xf[fm], acc[fm][fn], 0, 0, 0);
            }
        }
    }

    // ---- epilogue
#pragma unroll
    for (int fm = 0; fm < 2; ++fm)
#pragma unroll
        for (int fn = 0; fn < 2; ++fn) {
            const int mb = m0 + wm * 64 + fm * 32;
            const int nb = n0 + wn * 64 + fn * 32;
            if (natural) {
                if constexpr (EPI == EPI_QKV) epilogue_v_natural(a, acc[fm][fn], mb, nb + frow, lane);
            } else {
                epilogue_swapped<EPI, ACT>(a, acc[fm][fn], mb + frow, nb, lane);
            }
        }
}

template <int EPI, int ACT>
static int launch_t(const GemmArgs& a, hipStream_t s) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    static bool attr_set = false;
    auto kern = gemm_bf16_kernel<EPI, ACT>;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), GEMM_LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_gemm_bf16(int epi, const GemmArgs& a, hipStream_t s) {
    if (a.K % BK != 0 || a.K <= 0 || a.M <= 0 || a.N <= 0) { syl_set_error("launch_gemm_bf16", "K must be a positive multiple of 64"); return 1; }
    if (a.N % 4 != 0) { syl_set_error("launch_gemm_bf16", "N must be a multiple of 4"); return 1; }
    switch (epi) {
        case EPI_BF16:
            if (a.act == 1) return launch_t<EPI_BF16, 1>(a, s);
            if (a.act == 2) return launch_t<EPI_BF16, 2>(a, s);
            return launch_t<EPI_BF16, 0>(a, s);
        case EPI_F32:
            if (a.act == 1) return launch_t<EPI_F32, 1>(a, s);
            if (a.act == 2) return launch_t<EPI_F32, 2>(a, s);
            return launch_t<EPI_F32, 0>(a, s);
        case EPI_F32_RES: return launch_t<EPI_F32_RES, 0>(a, s);
        case EPI_QKV: return launch_t<EPI_QKV, 0>(a, s);
        case EPI_PROJ: return launch_t<EPI_PROJ, 0>(a, s);
    }
    syl_set_error("launch_gemm_bf16", "unknown epilogue");
    return 1;
}
