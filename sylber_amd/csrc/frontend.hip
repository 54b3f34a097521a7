// HBM-bound kernels of the conv frontend and the normalisations (gfx950).
//
//  * conv layer 0 of HubertFeatureEncoder (TP:160-175, reached from sylber/model/sylber.py:122):
//    Conv1d(1->512, k=10, s=5, no bias) -> GroupNorm(512 groups: per (b,c) statistics over TIME,
//    zero padding included) -> GELU.  GroupNorm needs the full-time mean/variance before the first
//    output can be written.  Because conv0 is linear in the waveform, its per-channel moments follow
//    from the 10 strided sums S_j = sum_l x[5l+j] and the 10x10 lag products R_jj' = sum_l x[5l+j]x[5l+j']:
//        mean_c = w_c . S / L,   E[v_c^2] = w_c^T R w_c / L.
//    So the statistics pass reads only the waveform (0.64 MB per 10 s clip) in fp64, and the
//    16.4 M-element conv0 output is produced exactly once, already normalised + activated, as
//    channels-last bf16 rows of 1 KiB (one wave stores one row: fully coalesced).
//  * LayerNorm(512/768) rows (TP:225-231, 441, 392-397): one wave per row, values held in registers,
//    two-pass mean/variance, fp32 and/or bf16 outputs (the bf16 copy feeds the next MFMA GEMM).
#include <stdlib.h>

#include "kernels.h"

#define NSTAT 65  // 10 sums + 55 upper-triangular lag products
#define C0_ROWS 256  // conv0 output rows per workgroup

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv0_stats_kernel(const float* __restrict__ wav, int Lmax, int L0, int chunk,
                                                          double* __restrict__ partials, int nchunk) {
    const int b = blockIdx.y, ck = blockIdx.x;
    const float* x = wav + (size_t)b * Lmax;
    const int l0 = ck * chunk;
    const int l1 = min(L0, l0 + chunk);
    double acc[NSTAT];
#pragma unroll
    for (int i = 0; i < NSTAT; ++i) acc[i] = 0.0;
    for (int l = l0 + threadIdx.x; l < l1; l += 256) {
        float v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = x[5 * l + j];
        int q = 10;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            acc[j] += (double)v[j];
#pragma unroll
            for (int k = j; k < 10; ++k) acc[q++] += (double)v[j] * (double)v[k];
        }
    }
    __shared__ double red[4][NSTAT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NSTAT; ++i) {
        double v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NSTAT)
        partials[((size_t)b * nchunk + ck) * NSTAT + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// one block per utterance, one thread per channel: a_c = gamma/sqrt(var+eps), b_c = beta - mean*a_c
__global__ __launch_bounds__(512) void conv0_finalize_kernel(const double* __restrict__ partials, int nchunk,
                                                             const float* __restrict__ w0, const float* __restrict__ gn_w,
                                                             const float* __restrict__ gn_b, int L0,
                                                             float* __restrict__ scale_shift) {
    __shared__ double st[NSTAT];
    const int b = blockIdx.x, c = threadIdx.x;
    if (c < NSTAT) {
        double s = 0.0;
        for (int k = 0; k < nchunk; ++k) s += partials[((size_t)b * nchunk + k) * NSTAT + c];
        st[c] = s;
    }
    __syncthreads();
    double w[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) w[j] = (double)w0[c * 10 + j];
    double mean = 0.0, ex2 = 0.0;
    int q = 10;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        mean += w[j] * st[j];
#pragma unroll
        for (int k = j; k < 10; ++k) {
            const double t = w[j] * w[k] * st[q++];
            ex2 += (k == j) ? t : 2.0 * t;
        }
    }
    mean /= (double)L0;
    ex2 /= (double)L0;
    double var = ex2 - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double a = (double)gn_w[c] / sqrt(var + 1e-5);
    scale_shift[((size_t)b * SYL_CONV + c) * 2 + 0] = (float)a;
    scale_shift[((size_t)b * SYL_CONV + c) * 2 + 1] = (float)((double)gn_b[c] - mean * a);
}

// grid (ceil(R0/256), B); wave w of a block produces rows l = 256*blockIdx.x + 64*w + i; lane owns 8 channels
// (80 tap weights + scale/shift stay in registers across 64 rows).  The 1285 waveform samples a block needs are staged in LDS once (coalesced), then every row reads its 10
// taps as LDS broadcasts instead of 10 wave-uniform global loads.
template <bool OUT_F32, bool ERF, int FMT>
__global__ __launch_bounds__(256) void conv0_gn_gelu_kernel(const float* __restrict__ wav, int Lmax, int L0, int R0,
                                                            const float* __restrict__ w0,
                                                            const float* __restrict__ scale_shift, void* __restrict__ out, long out_lo) {
    __shared__ float xs[C0_ROWS * 5 + 16];
    const int b = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int c0 = lane * 8;
    const float* x = wav + (size_t)b * Lmax;
    const int l_blk = blockIdx.x * C0_ROWS;
    for (int i = threadIdx.x; i < C0_ROWS * 5 + 5; i += 256) {
        const int idx = 5 * l_blk + i;
        xs[i] = idx < Lmax ? x[idx] : 0.f;
    }
    float w[8][10], sa[8], sb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 10; ++j) w[i][j] = w0[(c0 + i) * 10 + j];
        sa[i] = scale_shift[((size_t)b * SYL_CONV + c0 + i) * 2 + 0];
        sb[i] = scale_shift[((size_t)b * SYL_CONV + c0 + i) * 2 + 1];
        if constexpr (!OUT_F32 && FMT != FMT_SPLIT) {
            // 16-bit modes: the GroupNorm scale goes into the tap weights once per (utterance, channel) -- the affine
            // then costs nothing per value (the accumulation starts from the shift); the fp32 parity instantiation keeps the
            // reference's order (conv, then scale and shift)
#pragma unroll
            for (int j = 0; j < 10; ++j) w[i][j] *= sa[i];
        }
    }
    __syncthreads();
    const int lbase = l_blk + wave * (C0_ROWS / 4);
#pragma unroll 2
    for (int r = 0; r < C0_ROWS / 4; ++r) {
        const int l = lbase + r;
        if (l >= R0) break;
        float y[8];
        if (l < L0) {
            float xv[10];
            const float* xr = xs + 5 * (wave * (C0_ROWS / 4) + r);
#pragma unroll
            for (int j = 0; j < 10; ++j) xv[j] = xr[j];          // LDS broadcast
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                constexpr bool REF_ORDER = OUT_F32 || FMT == FMT_SPLIT;   // conv, then scale and shift (as the fp32 parity mode)
                float v = REF_ORDER ? 0.f : sb[i];
#pragma unroll
                for (int j = 0; j < 10; ++j) v = fmaf(w[i][j], xv[j], v);
                if constexpr (REF_ORDER) v = fmaf(v, sa[i], sb[i]);
                y[i] = ERF ? (FMT == FMT_SPLIT ? gelu_erf7(v) : gelu_erf(v)) : gelu_fast(v);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = 0.f;
        }
        const size_t o = ((size_t)b * R0 + l) * SYL_CONV + c0;
        if constexpr (OUT_F32) {
            float* op = (float*)out + o;
            *(float4*)op = make_float4(y[0], y[1], y[2], y[3]);
            *(float4*)(op + 4) = make_float4(y[4], y[5], y[6], y[7]);
        } else {
            // 1 GiB per 32 clips, consumed exactly once by conv1: stream it past the caches
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
            u32x4_t pk;
            pk[0] = H16<FMT>::pack2(y[0], y[1]); pk[1] = H16<FMT>::pack2(y[2], y[3]);
            pk[2] = H16<FMT>::pack2(y[4], y[5]); pk[3] = H16<FMT>::pack2(y[6], y[7]);
            __builtin_nontemporal_store(pk, (u32x4_t*)((bf16_t*)out + o));
            if constexpr (FMT == FMT_SPLIT) {
                u32x4_t lo;
                lo[0] = H16<FMT>::pack2_lo(y[0], y[1], pk[0]); lo[1] = H16<FMT>::pack2_lo(y[2], y[3], pk[1]);
                lo[2] = H16<FMT>::pack2_lo(y[4], y[5], pk[2]); lo[3] = H16<FMT>::pack2_lo(y[6], y[7], pk[3]);
                __builtin_nontemporal_store(lo, (u32x4_t*)((bf16_t*)out + out_lo + o));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same layer with the ten taps on the MATRIX pipe (16-bit modes).  The VALU kernel above spends
// 10 FMA + ~12 GELU instructions per value and is VALU-bound at half the store bandwidth; here the taps are one
// [32 channels x 16] . [16 x 32 rows] MFMA block (K = 10 padded to 16, the pad rows of the weight operand are zero), so
// only the GELU stays on the VALU.  Accuracy: the waveform window AND the (GroupNorm-scaled) tap weights enter as IEEE-half
// hi / lo pairs, three v_mfma_f32_32x32x16_f16 per block (hi.hi + lo.hi + hi.lo, fp32 accumulate): 2^-22-grade products,
// i.e. the fp32 FMA chain of the VALU kernel up to ~1e-6 relative -- far inside the 16-bit output rounding.  The
// GroupNorm shift is the accumulator's initial value, the scale is folded into the weights once per (utterance, channel).
// Orientation "lane = output row": A = weights (rows = channels), B = waveform windows (columns = rows l), so a lane's 16
// results per block are runs of 4 consecutive channels of its row; four blocks (128 channels) are transposed through a
// per-wave LDS region and leave as 16-byte chunks, consecutive lanes on consecutive chunks of a row (whole 128-byte lines).
#define C0M_WFR (16 * 64 * 32)                    // weight fragments: [16 blocks][64 lanes][hi 16 B | lo 16 B] = 32 KiB
#define C0M_STG (32 * (256 + 16))                 // per-wave staging: 32 rows x 128 channels (+ 16 B pad): 8704 B
#define C0M_LDS (C0M_WFR + 512 * 4 + (C0_ROWS * 5 + 16) * 4 + 4 * C0M_STG)
// NOSTORE (experiments build, SYLBER_OPT_CONV0_VALU = 2): the rows are produced into the LDS staging region and never leave it --
// the cost of conv0 as a PRODUCER inside another kernel (profiles/r05_conv0_fusion.md); results wrong by construction
template <int FMT, bool NOSTORE = false>
__global__ __launch_bounds__(256, 2) void conv0_mfma_kernel(const float* __restrict__ wav, int Lmax, int L0, int R0,
                                                            const float* __restrict__ w0, const float* __restrict__ scale_shift,
                                                            bf16_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    char* wfr = smem;
    float* shf = (float*)(smem + C0M_WFR);
    float* xs = shf + 512;
    const int b = blockIdx.y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    char* stg = smem + C0M_WFR + 512 * 4 + (C0_ROWS * 5 + 16) * 4 + wave * C0M_STG;
    const float* x = wav + (size_t)b * Lmax;
    // weight fragments of this utterance: entry (blk, ln): channel c = 32 blk + (ln & 31), taps k = 8 (ln >> 5) .. + 7
    for (int e = tid; e < 16 * 64; e += 256) {
        const int blk = e >> 6, ln = e & 63;
        const int c = 32 * blk + (ln & 31), k0 = 8 * (ln >> 5);
        const float sa = scale_shift[((size_t)b * SYL_CONV + c) * 2 + 0];
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ka = k0 + 2 * j, kb = ka + 1;
            const float wa = ka < 10 ? w0[c * 10 + ka] * sa : 0.f, wb = kb < 10 ? w0[c * 10 + kb] * sa : 0.f;
            hi[j] = H16<FMT_SPLIT>::pack2(wa, wb);
            lo[j] = H16<FMT_SPLIT>::pack2_lo(wa, wb, hi[j]);
        }
        *(uint4*)(wfr + e * 32) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *(uint4*)(wfr + e * 32 + 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
    for (int c = tid; c < 512; c += 256) shf[c] = scale_shift[((size_t)b * SYL_CONV + c) * 2 + 1];
    const int ml = lane & 31, h = lane >> 5;
    // a workgroup walks row blocks blockIdx.x, blockIdx.x + gridDim.x, ... of ITS utterance: the weight fragments are built
    // once, and the launch holds ~2 workgroups per CU instead of 16 waves of short-lived ones (stores from few, long-lived
    // workgroups reach a higher write bandwidth on this part: tools/ubench/hbm_bw.hip; measured in launch_conv0_gn_gelu)
    const int nblk = (R0 + C0_ROWS - 1) / C0_ROWS;
#pragma unroll 1
    for (int bx = blockIdx.x; bx < nblk; bx += gridDim.x) {
    const int l_blk = bx * C0_ROWS;
    __syncthreads();                                   // the previous block's window has been consumed
    for (int i = tid; i < C0_ROWS * 5 + 16; i += 256) {
        const int idx = 5 * l_blk + i;
        xs[i] = idx < Lmax ? x[idx] : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int rb = 0; rb < 2; ++rb) {
        const int lrow0 = wave * 64 + rb * 32;           // first row of this 32-row block inside the workgroup's 256
        const int l0 = l_blk + lrow0;
        if (l0 >= R0) break;
        // B operand: lane (row ml, half h) holds the window samples x[5 l + 8 h .. + 7] as half hi / lo pairs
        f16x8_t xh, xl;
        {
            const float* xr = xs + 5 * (lrow0 + ml) + 8 * h;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = xr[j];
                const _Float16 vh = (_Float16)v;
                xh[j] = vh; xl[j] = (_Float16)(v - (float)vh);
            }
        }
        const bool live = l0 + ml < L0;                  // rows in [L0, R0) are written as zeros
#pragma unroll 1
        for (int q4 = 0; q4 < 4; ++q4) {                 // 128 channels = 4 blocks per staging round
#pragma unroll
            for (int bi = 0; bi < 4; ++bi) {
                const int blk = q4 * 4 + bi;
                const f16x8_t whi = *(const f16x8_t*)(wfr + (blk * 64 + lane) * 32);
                const f16x8_t wlo = *(const f16x8_t*)(wfr + (blk * 64 + lane) * 32 + 16);
                f32x16_t acc;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 sv = *(const float4*)(shf + 32 * blk + 8 * g + 4 * h);
                    acc[4 * g + 0] = sv.x; acc[4 * g + 1] = sv.y; acc[4 * g + 2] = sv.z; acc[4 * g + 3] = sv.w;
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, xh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, xl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, xh, acc, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float y0 = gelu_fast(acc[4 * g + 0]), y1 = gelu_fast(acc[4 * g + 1]), y2 = gelu_fast(acc[4 * g + 2]), y3 = gelu_fast(acc[4 * g + 3]);
                    if (!live) { y0 = y1 = y2 = y3 = 0.f; }
                    uint2 pk; pk.x = H16<FMT>::pack2(y0, y1); pk.y = H16<FMT>::pack2(y2, y3);
                    *(uint2*)(stg + ml * (256 + 16) + (32 * bi + 8 * g + 4 * h) * 2) = pk;
                }
            }
            // out: 32 rows x 256 B; 16 chunks of 16 B per row, consecutive lanes on consecutive chunks
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int idx = it * 64 + lane;
                const int r = idx >> 4, ch = idx & 15;
                typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
                const u32x4_t v = *(const u32x4_t*)(stg + r * (256 + 16) + ch * 16);
                const int l = l0 + r;
                if constexpr (NOSTORE) { if (v.x == 0x7fc1dead && l < 0) out[0] = (bf16_t)v.y; }      // (keeps the staging reads alive)
                else if (l < R0) __builtin_nontemporal_store(v, (u32x4_t*)(out + ((size_t)b * R0 + l) * SYL_CONV + q4 * 128 + ch * 8));
            }
        }
    }
    }
}

int launch_conv0_stats(const float* wav, int B, int Lmax, int L0, double* partials, int nchunk, hipStream_t s) {
    const int chunk = (L0 + nchunk - 1) / nchunk;
    hipLaunchKernelGGL(conv0_stats_kernel, dim3(nchunk, B), dim3(256), 0, s, wav, Lmax, L0, chunk, partials, nchunk);
    HIP_TRY(hipGetLastError());
    return 0;
}
int launch_conv0_finalize(const double* partials, int nchunk, const float* w0, const float* gn_w, const float* gn_b, int B,
                          int L0, float* scale_shift, hipStream_t s) {
    hipLaunchKernelGGL(conv0_finalize_kernel, dim3(B), dim3(512), 0, s, partials, nchunk, w0, gn_w, gn_b, L0, scale_shift);
    HIP_TRY(hipGetLastError());
    return 0;
}
int launch_conv0_gn_gelu(const float* wav, int B, int Lmax, int L0, int R0, const float* w0, const float* scale_shift,
                         void* out, int out_f32, hipStream_t s, int fmt, long out_lo, int valu16) {
    dim3 grid((R0 + C0_ROWS - 1) / C0_ROWS, B);
    if (out_f32)
        hipLaunchKernelGGL((conv0_gn_gelu_kernel<true, true, FMT_BF16>), grid, dim3(256), 0, s, wav, Lmax, L0, R0, w0, scale_shift, out, 0L);
    else if (fmt == FMT_SPLIT)      // erf GELU in the reference's order (conv, then scale and shift), two half planes out
        hipLaunchKernelGGL((conv0_gn_gelu_kernel<false, true, FMT_SPLIT>), grid, dim3(256), 0, s, wav, Lmax, L0, R0, w0, scale_shift, out, out_lo);
    else if ((fmt == FMT_F16 || fmt == FMT_BF16) && valu16 == 1) {
        if (fmt == FMT_F16) hipLaunchKernelGGL((conv0_gn_gelu_kernel<false, false, FMT_F16>), grid, dim3(256), 0, s, wav, Lmax, L0, R0, w0, scale_shift, out, 0L);
        else hipLaunchKernelGGL((conv0_gn_gelu_kernel<false, false, FMT_BF16>), grid, dim3(256), 0, s, wav, Lmax, L0, R0, w0, scale_shift, out, 0L);
    } else if (fmt == FMT_F16 || fmt == FMT_BF16) {
        // 16-bit modes: the taps on the matrix pipe (conv0_mfma_kernel).  Same-box A/B (32 x 10 s): VALU kernel 0.260 ms,
        // this one 0.254 (4096 short-lived workgroups) / 0.244 (1024 row-block walkers) / 0.254 (512) / 0.395 (256): with the
        // taps off the VALU the kernel sits at 4.4 TB/s of stores, the write ceiling of this part (tools/ubench/hbm_bw.hip)
        static PerDeviceOnce once;
        if (once.need()) {
            HIP_TRY(hipFuncSetAttribute((const void*)conv0_mfma_kernel<FMT_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, C0M_LDS));
            HIP_TRY(hipFuncSetAttribute((const void*)conv0_mfma_kernel<FMT_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, C0M_LDS));
        }
        int gx = (1024 + B - 1) / B;                  // ~1024 workgroups however long the batch: gx row-block walkers per utterance
        gx = gx < 1 ? 1 : (gx > (int)grid.x ? (int)grid.x : gx);
        const dim3 pgrid(gx, B);
#ifdef SYLBER_GEMM_ASM_EXPERIMENTS
        if (valu16 == 2) {
            static PerDeviceOnce once2;
            if (once2.need()) HIP_TRY(hipFuncSetAttribute((const void*)conv0_mfma_kernel<FMT_BF16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C0M_LDS));
            hipLaunchKernelGGL((conv0_mfma_kernel<FMT_BF16, true>), pgrid, dim3(256), C0M_LDS, s, wav, Lmax, L0, R0, w0, scale_shift, (bf16_t*)out);
        } else
#endif
        if (fmt == FMT_F16) hipLaunchKernelGGL((conv0_mfma_kernel<FMT_F16>), pgrid, dim3(256), C0M_LDS, s, wav, Lmax, L0, R0, w0, scale_shift, (bf16_t*)out);
        else hipLaunchKernelGGL((conv0_mfma_kernel<FMT_BF16>), pgrid, dim3(256), C0M_LDS, s, wav, Lmax, L0, R0, w0, scale_shift, (bf16_t*)out);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, D/64 values per lane in registers.
#ifndef LN_ROWS
#define LN_ROWS 1
#endif
template <int D, bool IN_BF16, int FMT, int FMT_IN = FMT>
__global__ __launch_bounds__(256) void layernorm_kernel(const LnArgs a) {
    constexpr int V = D / 256;   // float4 groups per lane (2 for 512, 3 for 768)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // XCD-aware row order: the rows an XCD normalises are the rows its GEMM tiles produced / will consume.
    // A workgroup handles 4 * LN_ROWS consecutive rows (a wave LN_ROWS of them, interleaved with its siblings).
    // LN_ROWS = 4 (1024 workgroups, the grid size at which tools/ubench/hbm_bw.hip's copy peaks) measured the same
    // 0.33 ms per forward as LN_ROWS = 1: the kernel is at what this read + write mix sustains.
    const int m_first = xcd_remap(blockIdx.x, gridDim.x) * (4 * LN_ROWS) + wave;
#pragma unroll 1
    for (int rr = 0; rr < LN_ROWS; ++rr) {
    const int m = m_first + 4 * rr;
    if (m >= a.M) break;
    int orow = m;
    if (a.Tp > 0) {
        const int b = m / a.Tp, t = m - b * a.Tp;
        if (t >= a.T) continue;
        orow = b * a.T + t;
    }
    float x[V][4];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = i * 256 + lane * 4;
        if constexpr (IN_BF16) {
            const uint2 raw = *(const uint2*)((const bf16_t*)a.in + (size_t)m * a.ld_in + c);
            x[i][0] = H16<FMT_IN>::up((bf16_t)(raw.x & 0xffff)); x[i][1] = H16<FMT_IN>::up((bf16_t)(raw.x >> 16));
            x[i][2] = H16<FMT_IN>::up((bf16_t)(raw.y & 0xffff)); x[i][3] = H16<FMT_IN>::up((bf16_t)(raw.y >> 16));
            if constexpr (FMT_IN == FMT_SPLIT) {
                const uint2 lo = *(const uint2*)((const bf16_t*)a.in + a.in_lo + (size_t)m * a.ld_in + c);
                x[i][0] += H16<FMT_IN>::up((bf16_t)(lo.x & 0xffff)); x[i][1] += H16<FMT_IN>::up((bf16_t)(lo.x >> 16));
                x[i][2] += H16<FMT_IN>::up((bf16_t)(lo.y & 0xffff)); x[i][3] += H16<FMT_IN>::up((bf16_t)(lo.y >> 16));
            }
        } else {
            const float4 v = *(const float4*)((const float*)a.in + (size_t)m * a.ld_in + c);
            x[i][0] = v.x; x[i][1] = v.y; x[i][2] = v.z; x[i][3] = v.w;
        }
        if (a.res) {
            const float4 r = *(const float4*)(a.res + (size_t)m * a.ld_res + c);
            x[i][0] += r.x; x[i][1] += r.y; x[i][2] += r.z; x[i][3] += r.w;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) s += (x[i][0] + x[i][1]) + (x[i][2] + x[i][3]);
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = x[i][j] - mean; q = fmaf(d, d, q); }
    const float var = wave_sum(q) * (1.0f / D);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    if (a.out_stats && lane == 0) *(float2*)(a.out_stats + (size_t)m * 2) = make_float2(mean, rstd);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = i * 256 + lane * 4;
        const float4 g = *(const float4*)(a.gamma + c);
        const float4 be = *(const float4*)(a.beta + c);
        const float y0 = fmaf((x[i][0] - mean) * rstd, g.x, be.x);
        const float y1 = fmaf((x[i][1] - mean) * rstd, g.y, be.y);
        const float y2 = fmaf((x[i][2] - mean) * rstd, g.z, be.z);
        const float y3 = fmaf((x[i][3] - mean) * rstd, g.w, be.w);
        if (a.out_f32) *(float4*)(a.out_f32 + (size_t)orow * a.ld_f32 + c) = make_float4(y0, y1, y2, y3);
        if (a.out_bf16) {
            uint2 pk; pk.x = H16<FMT>::pack2(y0, y1); pk.y = H16<FMT>::pack2(y2, y3);
            *(uint2*)(a.out_bf16 + (size_t)m * a.ld_bf16 + c) = pk;
            if constexpr (FMT == FMT_SPLIT) {
                uint2 lo; lo.x = H16<FMT>::pack2_lo(y0, y1, pk.x); lo.y = H16<FMT>::pack2_lo(y2, y3, pk.y);
                *(uint2*)(a.out_bf16 + a.out_lo + (size_t)m * a.ld_bf16 + c) = lo;
            }
        }
        if (a.out_fp8) {
            // a 32-feature scale block = the 4 values of 8 consecutive lanes
            float amax = fmaxf(fmaxf(fabsf(y0), fabsf(y1)), fmaxf(fabsf(y2), fabsf(y3)));
            amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
            amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
            amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
            const unsigned e = mx_e8m0(amax);
            const float inv = mx_inv_scale(e);
            *(unsigned*)(a.out_fp8 + (size_t)m * a.ld_fp8 + c) = pack_fp8x4(y0 * inv, y1 * inv, y2 * inv, y3 * inv);
            if ((lane & 7) == 0) a.out_scale[mx_scale_index(m, c >> 5, a.scale_rows)] = (uint8_t)e;
        }
    }
    }
}

template <int FMT>
static bool launch_ln_fmt(const LnArgs& a, dim3 grid, hipStream_t s) {
    if (a.D == 768 && !a.in_bf16) hipLaunchKernelGGL((layernorm_kernel<768, false, FMT>), grid, dim3(256), 0, s, a);
    else if (a.D == 768 && a.in_bf16) hipLaunchKernelGGL((layernorm_kernel<768, true, FMT>), grid, dim3(256), 0, s, a);
    else if (a.D == 512 && !a.in_bf16) hipLaunchKernelGGL((layernorm_kernel<512, false, FMT>), grid, dim3(256), 0, s, a);
    else if (a.D == 512 && a.in_bf16) hipLaunchKernelGGL((layernorm_kernel<512, true, FMT>), grid, dim3(256), 0, s, a);
    else return false;
    return true;
}

int launch_layernorm(const LnArgs& a, hipStream_t s) {
    dim3 grid((a.M + 4 * LN_ROWS - 1) / (4 * LN_ROWS));
    if (a.in_bf16 && a.fmt_in != a.fmt) {
        // SYLBER_MIXED16: the one hand-over between the fp16 conv stack and the bf16 encoder (feature-projection LayerNorm)
        if (a.D != 512 || a.fmt_in != FMT_F16 || a.fmt != FMT_BF16) { syl_set_error("launch_layernorm", "unsupported format pair"); return 1; }
        hipLaunchKernelGGL((layernorm_kernel<512, true, FMT_BF16, FMT_F16>), grid, dim3(256), 0, s, a);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (a.fmt == FMT_SPLIT) {
        if (a.D == 768 && !a.in_bf16) hipLaunchKernelGGL((layernorm_kernel<768, false, FMT_SPLIT>), grid, dim3(256), 0, s, a);
        else if (a.D == 512 && a.in_bf16) hipLaunchKernelGGL((layernorm_kernel<512, true, FMT_SPLIT>), grid, dim3(256), 0, s, a);
        else { syl_set_error("launch_layernorm", "split16: only LN(768) of fp32 rows and LN(512) of split rows"); return 1; }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (a.fmt == FMT_F16 ? launch_ln_fmt<FMT_F16>(a, grid, s) : launch_ln_fmt<FMT_BF16>(a, grid, s)) {}
    else { syl_set_error("launch_layernorm", "D must be 512 or 768"); return 1; }
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = *(const float4*)(in + i * 4);
        uint2 pk; pk.x = pack_bf16x2(v.x, v.y); pk.y = pack_bf16x2(v.z, v.w);
        *(uint2*)(out + i * 4) = pk;
    }
}
int launch_f32_to_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s) {
    if (n % 4) { syl_set_error("launch_f32_to_bf16", "n must be a multiple of 4"); return 1; }
    const size_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid ? grid : 1), dim3(256), 0, s, in, out, n4);
    HIP_TRY(hipGetLastError());
    return 0;
}

// f32 -> split16 planes: hi = half(x) at out, lo = half(x - hi) at out + lo_off (element offset)
__global__ __launch_bounds__(256) void f32_to_split16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long lo_off, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = *(const float4*)(in + i * 4);
        uint2 hi; hi.x = H16<FMT_SPLIT>::pack2(v.x, v.y); hi.y = H16<FMT_SPLIT>::pack2(v.z, v.w);
        uint2 lo; lo.x = H16<FMT_SPLIT>::pack2_lo(v.x, v.y, hi.x); lo.y = H16<FMT_SPLIT>::pack2_lo(v.z, v.w, hi.y);
        *(uint2*)(out + i * 4) = hi;
        *(uint2*)(out + lo_off + i * 4) = lo;
    }
}
int launch_f32_to_split16(const float* in, bf16_t* out, long lo_off, size_t n, hipStream_t s) {
    if (n % 4 || lo_off % 4) { syl_set_error("launch_f32_to_split16", "n and the plane offset must be multiples of 4"); return 1; }
    const size_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(f32_to_split16_kernel, dim3(grid ? grid : 1), dim3(256), 0, s, in, out, lo_off, n4);
    HIP_TRY(hipGetLastError());
    return 0;
}

// compacting copy bf16 [B][Tp][D] (row stride ld_in) -> f32 [B][T][D]
__global__ __launch_bounds__(256) void bf16_rows_to_f32_kernel(const bf16_t* __restrict__ in, long ld_in, float* __restrict__ out,
                                                               int Tp, int T, int D, int fmt, long in_lo) {
    const int b = blockIdx.y, t = blockIdx.x;
    const bf16_t* src = in + ((size_t)b * Tp + t) * ld_in;
    float* dst = out + ((size_t)b * T + t) * D;
    for (int c = threadIdx.x; c < D; c += 256)
        dst[c] = fmt == FMT_SPLIT ? H16<FMT_F16>::up(src[c]) + H16<FMT_F16>::up(src[in_lo + c]) : (fmt == FMT_F16 ? H16<FMT_F16>::up(src[c]) : bf2f(src[c]));
}
int launch_bf16_to_f32_rows(const bf16_t* in, long ld_in, float* out, int B, int Tp, int T, int D, hipStream_t s, int fmt, long in_lo) {
    hipLaunchKernelGGL(bf16_rows_to_f32_kernel, dim3(T, B), dim3(256), 0, s, in, ld_in, out, Tp, T, D, fmt, in_lo);
    HIP_TRY(hipGetLastError());
    return 0;
}
