// Residual GEMM with the LayerNorm that follows it in the SAME kernel: one workgroup owns FULL rows.
//
//   x[m][:]   = X[m][:] . W^T + bias + LN_prev(res[m][:])          (the residual is the previous LayerNorm's output, re-derived
//                                                                    from the pre-LN sum `res` and its row statistics: EPI_F32_RESLN)
//   pre[m][:] = x                      (fp32, in place over res)    -> the next residual GEMM re-derives LN(x) from it
//   stats[m]  = (mean, rstd) of x over its 768 columns
//   hbf[m][:] = LN(x) = (x - mean) rstd gamma + beta               (16-bit copy: the A operand of the next GEMM)
//
// Reference semantics: transformers HubertEncoderLayer (post-LN), TP:403-412: hidden = layer_norm(attn_residual + out_proj(ctx)),
// reached from sylber/model/sylber.py:122.  Replaces one EPI_F32_RESLN GEMM launch + one layernorm_kernel launch: the
// LayerNorm no longer re-reads the 50 MB of pre-LN rows the GEMM epilogue has just written.
//
// Tile 64 rows x 768 columns (N = 768 is the whole row), 8 waves as 1 x 8 (each 64 x 96: 2 x 3 fragments of 32 x 32, 96
// accumulator registers), K step 32 in a 3-slot LDS ring (3 x 52 KiB = 156 of the 160 KiB), one barrier per step, software-
// pipelined inside the wave like gemm8u (fragments of the next k-substep and the LDS-DMA of step s+2 between the MFMAs).
// 16384 rows = 256 tiles = one workgroup per CU.  The tile is operand-hungry -- (64 + 768) x 64 B per 12 MFMAs per wave --
// so its K loop is DMA-bound (52 one-KiB pieces per step against 768 MFMA cycles per SIMD), which is why it is used only
// where the fused LayerNorm pays for it: K = 768 (out-proj).  Measured same-box against GEMM + LayerNorm: DESIGN.md §6.
#include "kernels.h"

typedef __attribute__((address_space(3))) void* lds_vptr_r;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_r(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0xffffffffu, 0x00020000);
}
__device__ __forceinline__ void glds16r(__amdgpu_buffer_rsrc_t rs, int voff, int soff, void* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr_r)l, 16, voff, soff, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt_r() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#define RL_FENCE() __builtin_amdgcn_sched_barrier(0)

#define RL_BM 64
#define RL_N 768
#define RL_WT (RL_N * 64)                 // W rows of one K step: 49152 B
#define RL_STAGE (RL_WT + RL_BM * 64)     // + X rows: 53248 B
#define RL_LDS (3 * RL_STAGE)             // 159744 B

template <int FMT>
__global__ __launch_bounds__(512, 2) void gemm_rowln_kernel(const GemmArgs a) {
    constexpr int FM = 2, FN = 3, RB = 64;
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int ntiles = (a.M + RL_BM - 1) / RL_BM;
    const int m0 = xcd_remap(blockIdx.x, ntiles) * RL_BM;

    // ---- staging.  Pieces (16 rows x 64 B) 0..47 are W rows, 48..51 the X rows: piece wave + 8 i is a W piece for every
    // wave when i < 6; i == 6 is the X piece of waves 0..3 (waves 4..7 have none)
    const int srow = lane >> 2, spos = lane & 3;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc_r(a.X + (size_t)m0 * a.ldx), rw = make_rsrc_r(a.W);
    int voff[7], lds_off[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int p = wave + 8 * i;
        const bool isx = i == 6;
        const int r = (isx ? (p - 48 < 4 ? p - 48 : 3) : p) * 16 + srow;
        const int c = spos ^ ((r >> 2) & 3);
        if (isx) { int xm = m0 + r; xm = xm < a.M ? xm : a.M - 1; voff[i] = (int)(((long)(xm - m0) * a.ldx + c * 8) * 2); }
        else voff[i] = (r * a.K + c * 8) * 2;
        lds_off[i] = isx ? RL_WT + (p - 48 < 4 ? p - 48 : 3) * 1024 : p * 1024;
    }
    const bool has_x = wave < 4;
    auto dma1 = [&](int ks, int i, char* base) {
        if (i == 6) { if (has_x) glds16r(rx, voff[6], ks * 64, base + lds_off[6]); }
        else glds16r(rw, voff[i], ks * 64, base + lds_off[i]);
    };
    auto stage = [&](int ks, int slot) {
#pragma unroll
        for (int i = 0; i < 7; ++i) dma1(ks, i, smem + slot * RL_STAGE);
    };
    // my pieces per step: 7 (waves 0..3) or 6; "first half" = pieces 0..2, issued before the step's barrier
    auto wait_keep3 = [&]() { wait_vmcnt_r<3>(); };

    const int frow = lane & 31, swz = (lane >> 2) & 3, fhalf = lane >> 5;
    const int koff0 = (((0 + fhalf) ^ swz) << 4), koff1 = (((2 + fhalf) ^ swz) << 4);
    const int xrow_off = RL_WT + frow * RB;
    const int wrow_off = (wave * 96 + frow) * RB;
    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8_t xf[2][FM], wf[2][FN];
    auto read_frags = [&](const char* sb, int koff, int buf) {
#pragma unroll
        for (int f = 0; f < FM; ++f) xf[buf][f] = *(const bf16x8_t*)(sb + xrow_off + f * 32 * RB + koff);
#pragma unroll
        for (int f = 0; f < FN; ++f) wf[buf][f] = *(const bf16x8_t*)(sb + wrow_off + f * 32 * RB + koff);
    };
    // 6 MFMAs of one k-substep with DMA pieces [p0, p1) of step `ks` between them (one after each MFMA from the second on)
    auto mfmas_dma = [&](int buf, int ks, char* dbase, int p0, int p1, bool on) {
#pragma unroll
        for (int i = 0; i < FM * FN; ++i) {
            const int fm = i / FN, fn = i % FN;
            acc[fm][fn] = H16<FMT>::mfma(wf[buf][fn], xf[buf][fm], acc[fm][fn]);
            const int q = p0 + i - 1;
            if (i >= 1 && q < p1) {
                RL_FENCE();
                if (on) dma1(ks, q, dbase);
                RL_FENCE();
            }
        }
    };
    const int nt = a.K / 32;
    stage(0, 0);
    if (nt > 1) stage(1, 1);
    // step 0 landed; younger: step 1 (7 or 6 pieces)
    if (nt > 1) { if (has_x) wait_vmcnt_r<7>(); else wait_vmcnt_r<6>(); } else wait_vmcnt_r<0>();
    __builtin_amdgcn_s_barrier();
    read_frags(smem, koff0, 0);
    int slot = 0;
    for (int s = 0; s < nt; ++s) {
        const char* sb = smem + slot * RL_STAGE;
        const int nslot = slot == 2 ? 0 : slot + 1;
        char* dbase = smem + (slot == 0 ? 2 : slot - 1) * RL_STAGE;     // slot of step s-1 = slot of step s+2
        const bool dma = s + 2 < nt;
        RL_FENCE();
        read_frags(sb, koff1, 1);
        RL_FENCE();
        mfmas_dma(0, s + 2, dbase, 0, 3, dma);
        RL_FENCE();
        if (s + 1 < nt) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // my pieces of step s+1 have landed; younger: pieces 0..2 of step s+2
            if (dma) wait_keep3(); else wait_vmcnt_r<0>();
            __builtin_amdgcn_s_barrier();
            RL_FENCE();
            read_frags(smem + nslot * RL_STAGE, koff0, 0);
            RL_FENCE();
        }
        mfmas_dma(1, s + 2, dbase, 3, 7, dma);
        slot = nslot;
    }
    __builtin_amdgcn_s_barrier();                     // every wave is done with the operand ring: it becomes scratch

    // ---- epilogue.  The accumulators leave through LDS (the ring is free: 32 rows x 768 fp32 = 96 KiB per half), and the
    // rows are then finished ONE WAVE PER ROW, exactly as the unfused pair does it: x = (acc + bias) + LN_prev(res) with the
    // expression of the EPI_F32_RESLN epilogue, statistics and LayerNorm with the expressions and the summation order of
    // layernorm_kernel (lane owns columns 256 i + 4 lane .. + 3) -- so pre, stats and the 16-bit copy are bit for bit what
    // the GEMM launch + the LayerNorm launch produce, and every global access is a run of whole 128-byte lines.
    constexpr int XS = RL_N * 4 + 16;                 // LDS row stride: 772 dwords -> rows 4 banks apart, float4 writes conflict-free
    const int ml = lane & 31, h = lane >> 5;
    // column vectors of this lane's 12 columns (the same for every row): loaded once, in flight under the staging
    float4 cb[3], cg[3], ce[3], cg2[3], ce2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = i * 256 + lane * 4;
        cb[i] = *(const float4*)(a.bias + c); cg[i] = *(const float4*)(a.ln_gamma + c); ce[i] = *(const float4*)(a.ln_beta + c);
        cg2[i] = *(const float4*)(a.ln_gamma_out + c); ce2[i] = *(const float4*)(a.ln_beta_out + c);
    }
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        // the four rows this wave finishes in this half: their residual rows and old statistics are requested BEFORE the
        // staging barrier, all at once (memory-level parallelism: a wave has only 8 rows to hide HBM latency with)
        float4 rres[4][3];
        float2 so[4];
        int mr[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int m = m0 + 32 * fm + wave * 4 + rr;
            mr[rr] = m < a.M ? m : a.M - 1;
            so[rr] = *(const float2*)(a.ln_stats + (size_t)mr[rr] * 2);
#pragma unroll
            for (int i = 0; i < 3; ++i) rres[rr][i] = *(const float4*)(a.res + (size_t)mr[rr] * a.ldres + i * 256 + lane * 4);
        }
        if (fm > 0) __syncthreads();                  // the rows of the previous half have been read
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = wave * 96 + 32 * fn + 8 * g + 4 * h;
                *(float4*)(smem + ml * XS + n * 4) = make_float4(acc[fm][fn][4 * g + 0], acc[fm][fn][4 * g + 1], acc[fm][fn][4 * g + 2], acc[fm][fn][4 * g + 3]);
            }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = wave * 4 + rr;            // 32 rows over 8 waves
            const int m = m0 + 32 * fm + row;
            float x[3][4];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int c = i * 256 + lane * 4;
                const float4 ac = *(const float4*)(smem + row * XS + c * 4);
                const float4 r = rres[rr][i];
                x[i][0] = (ac.x + cb[i].x) + fmaf((r.x - so[rr].x) * so[rr].y, cg[i].x, ce[i].x);
                x[i][1] = (ac.y + cb[i].y) + fmaf((r.y - so[rr].x) * so[rr].y, cg[i].y, ce[i].y);
                x[i][2] = (ac.z + cb[i].z) + fmaf((r.z - so[rr].x) * so[rr].y, cg[i].z, ce[i].z);
                x[i][3] = (ac.w + cb[i].w) + fmaf((r.w - so[rr].x) * so[rr].y, cg[i].w, ce[i].w);
            }
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) sm += (x[i][0] + x[i][1]) + (x[i][2] + x[i][3]);
            const float mean = wave_sum(sm) * (1.0f / RL_N);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = x[i][j] - mean; q = fmaf(d, d, q); }
            const float var = wave_sum(q) * (1.0f / RL_N);
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
            if (m < a.M) {
                if (lane == 0) *(float2*)(a.ln_stats_out + (size_t)m * 2) = make_float2(mean, rstd);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int c = i * 256 + lane * 4;
                    *(float4*)((float*)a.out0 + (size_t)m * a.ld0 + c) = make_float4(x[i][0], x[i][1], x[i][2], x[i][3]);
                    const float y0 = fmaf((x[i][0] - mean) * rstd, cg2[i].x, ce2[i].x), y1 = fmaf((x[i][1] - mean) * rstd, cg2[i].y, ce2[i].y);
                    const float y2 = fmaf((x[i][2] - mean) * rstd, cg2[i].z, ce2[i].z), y3 = fmaf((x[i][3] - mean) * rstd, cg2[i].w, ce2[i].w);
                    uint2 pk; pk.x = H16<FMT>::pack2(y0, y1); pk.y = H16<FMT>::pack2(y2, y3);
                    *(uint2*)((bf16_t*)a.out1 + (size_t)m * RL_N + c) = pk;
                }
            }
        }
    }
}

// true when the fused kernel is the better launch for this shape: whole rows of 768, and a tile count that fills the chip
bool gemm_rowln_applicable(const GemmArgs& a) {
    if (a.N != RL_N || a.K % 32 != 0 || a.K < 64 || (a.fmt != FMT_BF16 && a.fmt != FMT_F16)) return false;
    const long tiles = (a.M + RL_BM - 1) / RL_BM;
    const long rounds = (tiles + 255) / 256;
    return tiles * 4 >= rounds * 256 * 3;             // at least 75 % of the workgroup slots of its rounds are used
}

// a: X [M][K] (ldx), W [768][K], bias, res / ldres + ln_stats / ln_gamma / ln_beta (the previous LayerNorm, EPI_F32_RESLN),
// out0 = pre-LN sum fp32 [M][ld0] (may alias res), out1 = 16-bit LayerNorm output [M][768], ln_stats_out [M][2],
// ln_gamma_out / ln_beta_out = this LayerNorm's affine
int launch_gemm_rowln(const GemmArgs& a, hipStream_t s) {
    if (!gemm_rowln_applicable(a)) { syl_set_error("launch_gemm_rowln", "shape not supported by the fused row kernel"); return 1; }
    const int tiles = (a.M + RL_BM - 1) / RL_BM;
    static PerDeviceOnce once;
    if (once.need()) {
        HIP_TRY(hipFuncSetAttribute((const void*)gemm_rowln_kernel<FMT_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, RL_LDS));
        HIP_TRY(hipFuncSetAttribute((const void*)gemm_rowln_kernel<FMT_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, RL_LDS));
    }
    if (a.fmt == FMT_F16) hipLaunchKernelGGL(gemm_rowln_kernel<FMT_F16>, dim3(tiles), dim3(512), RL_LDS, s, a);
    else hipLaunchKernelGGL(gemm_rowln_kernel<FMT_BF16>, dim3(tiles), dim3(512), RL_LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}
