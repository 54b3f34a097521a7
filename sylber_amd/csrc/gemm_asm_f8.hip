// MXFP8 GEMMs with the hand-scheduled X3 K loop (BASELINE.json configs[4]; tile ids 85 / 91 of the fp8 launcher).
//
// Same contract as gemm_mxfp8.hip (out = epilogue(sum_k X8 2^XS . W8 2^WS), e4m3 operands with one E8M0 scale per 32 k; reference:
// the encoder's nn.Linear layers, transformers modeling_hubert.py TP:234-368 reached from sylber/model/sylber.py:122) and the same
// epilogues; what differs is the K loop: the generated inline-asm loop of gemm_asm.hip's X3 kernels carried over to the block-scaled
// fp8 MFMA (tools/gen_gemm_asm.py emit_f8 -> gemm_asm_f8*.inc).  The byte geometry is the bf16 kernel's -- 256-row x 128-byte
// operand tiles, three X slots + two W slots = the whole 160 KiB, whole-line LDS-DMA through buffer descriptors, one barrier per K
// step, four waves of 128 x 32 FN -- but a 128-byte row holds 128 k, so a step feeds TWICE the FLOPs through the same LDS-DMA and
// fragment-read traffic (v_mfma_scale_f32_32x32x64_f8f6f4: 64 cycles per 32x32x64).  Round 1-3's fp8 kernels (gemm_mxfp8.hip: 64-byte
// rows, 4-slot ring, hipcc's schedule) stay as the fallback for ragged tiles and as the bitwise reference of this one.
#include "kernels.h"
#include "gemm_epilogue.h"
#include "gemm_epilogue_f8.h"

typedef __attribute__((address_space(3))) void* lds_vptr_f;
typedef int i32x4f_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16_f(__amdgpu_buffer_rsrc_t rs, int voff, int soff, void* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr_f)l, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ i32x4f_t rsrc_words_f(const void* base) {
    const unsigned long long p = (unsigned long long)base;
    i32x4f_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(p >> 32) & 0xffffu));
    r.z = (int)0xffffffffu;
    r.w = 0x00020000;
    return r;
}

// FN = 4: 256x256 tile, FN = 3: 256x192.  Whole tiles only (M % 256 == 0, N % (64 FN) == 0; the launcher checks), K % 256 == 0, K >= 512.
// SCL (256x192 only): the block scales reach the lanes through LDS (one 1-KiB LDS-DMA piece per wave and step + ds_read_u8) instead
// of one byte load per lane, fragment and slice: 1 instead of 14 requests per wave and step on the texture path (same box: FFN2 52.9
// -> 48.1 us, out-proj 30.9 -> 29.9, q,k,v 57.9 -> 57.0).  The 256x256 tile has no LDS to spare and keeps the byte loads
template <int EPI, int ACT, int FN, bool SCL = false>
__global__ __launch_bounds__(256, 1) void gemmf8_kernel(const GemmF8Args a) {
    static_assert(!SCL || FN == 3, "LDS-staged scales: the 256x192 tile (16 KiB of LDS to spare)");
    constexpr int FM = 4, BM = 256, BN = 64 * FN, RB = 128;
    constexpr int XT = BM * RB, WS_ = BN * RB;
    constexpr int XRING = 3 * XT;
    constexpr int NPW = (BM + BN) / 8 / 4, NXW = 8;
    constexpr bool STAGED = (EPI == EPI_QK || EPI == EPI_MXFP8 || EPI == EPI_QK8);
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int M = a.g.M, N = a.g.N, K = a.g.K;
    const int tiles_n = N / BN, tiles_m = M / BM;
    const int ntiles = tiles_m * tiles_n;

    const int srow = lane >> 3, spos = lane & 7;
    struct Tile { int m0, n0; int voff[NPW]; };
    auto setup = [&](int tile_id, Tile& t) {
        const int wg = xcd_remap(tile_id, ntiles);
        t.m0 = (wg / tiles_n) * BM;
        t.n0 = (wg % tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int p = wave + 4 * i;
            const bool isx = i < NXW;
            const int r = (isx ? p : p - 32) * 8 + srow;
            const int c = spos ^ ((r >> 1) & 7);
            t.voff[i] = isx ? (int)((long)r * a.ldx8 + c * 16) : (r * K + c * 16);
        }
    };
    auto stage_x = [&](const Tile& t, int s, int slot) {
        const __amdgpu_buffer_rsrc_t bx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X8 + (size_t)t.m0 * a.ldx8), 0, (int)0xffffffffu, 0x00020000);
#pragma unroll
        for (int i = 0; i < NXW; ++i) glds16_f(bx, t.voff[i], s * 128, smem + slot * XT + (wave + 4 * i) * 1024);
    };
    auto stage_w = [&](const Tile& t, int s, int slot) {
        const __amdgpu_buffer_rsrc_t bw = __builtin_amdgcn_make_buffer_rsrc((void*)(a.W8 + (size_t)t.n0 * K), 0, (int)0xffffffffu, 0x00020000);
#pragma unroll
        for (int i = NXW; i < NPW; ++i) glds16_f(bw, t.voff[i], s * 128, smem + XRING + slot * WS_ + (wave + 4 * (i - NXW)) * 1024);
    };
    const int lds0 = (int)(unsigned)(unsigned long long)(lds_vptr_f)smem;
    const int lbase = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);

    // fragment addresses: the bf16 loop's (slice kk of a row = 16-byte chunks 2 kk + half, XORed with the row's swizzle); an fp8
    // K slice s of 64 is the pair kk = 2 s, 2 s + 1
    const int frow = lane & 31, swz = (lane >> 1) & 7, fhalf = lane >> 5;
    int ax[4], aw[4], awh[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int ko = ((2 * kk + fhalf) ^ swz) << 4;
        ax[kk] = lds0 + (wm * 128 + frow) * RB + ko;
        aw[kk] = lds0 + XRING + (wn * 32 * FN + frow) * RB + ko;
        awh[kk] = aw[kk] + WS_;
    }
    // block scales: [K / 64][rows][2] bytes; lane (r, h) reads ITS byte (K block h of the slice) per fragment and slice
    const int vsx = (wm * 128 + frow) * 2 + fhalf, vsw = (wn * 32 * FN + frow) * 2 + fhalf;
    const int xstep = __builtin_amdgcn_readfirstlane((int)(a.xs_rows * 2)), wstep = __builtin_amdgcn_readfirstlane((int)(a.ws_rows * 2));

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    Tile cur, nxt;
    setup(tile, cur);
    // scales of step 0 (requested first: the oldest loads, so the wait for step 0's pieces covers them)
    int xs0[2][4], ws0[2][FN], xs1[2][4], ws1[2][FN];
    auto scales0 = [&](const Tile& t) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
#pragma unroll
            for (int f = 0; f < 4; ++f) xs0[sl][f] = (int)a.XS[(size_t)sl * a.xs_rows * 2 + (size_t)(t.m0 + wm * 128 + f * 32 + frow) * 2 + fhalf];
#pragma unroll
            for (int f = 0; f < FN; ++f) ws0[sl][f] = (int)a.WS[(size_t)sl * a.ws_rows * 2 + (size_t)(t.n0 + wn * 32 * FN + f * 32 + frow) * 2 + fhalf];
        }
    };
    scales0(cur);
    stage_x(cur, 0, 0); stage_w(cur, 0, 0);
    stage_x(cur, 1, 1); stage_w(cur, 1, 1);
    stage_x(cur, 2, 2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW + NXW) : "memory");
    for (;;) {
        __builtin_amdgcn_s_barrier();
        f32x16_t acc[FM][FN];
        int koff = 2 * 128;
        int nloop = (K / 128 - 4) / 2;
        {
            const int (&voff)[NPW] = cur.voff;
            i32x4f_t rx = rsrc_words_f(a.X8 + (size_t)cur.m0 * a.ldx8), rw = rsrc_words_f(a.W8 + (size_t)cur.n0 * K);
            i32x4f_t rxs = rsrc_words_f(a.XS + (size_t)cur.m0 * 2), rws = rsrc_words_f(a.WS + (size_t)cur.n0 * 2);
            int axc[4] = {ax[0], ax[1], ax[2], ax[3]};
            int xr = 0, xwl, dlt, kofx;
            const int cneg = -2 * XT;
            if constexpr (SCL) {
                // scales through LDS (the 256x192 tile leaves 16 KiB free): one 1-KiB LDS-DMA piece per wave and step -- waves 0 / 2
                // the X scales of the next step, waves 1 / 3 the W scales (lanes 0-31: the step's first 64-wide slice, lanes 32-63 its
                // second; 16 bytes = 8 rows per lane, rows beyond the tile clamped) -- into a double buffer behind the W ring
                const bool wsc = wave & 1;
                const int rows8 = wsc ? BN / 8 : BM / 8;
                const int l31 = (lane & 31) < rows8 ? (lane & 31) : rows8 - 1;
                const int vsc = (int)((long)(lane >> 5) * (wsc ? a.ws_rows : a.xs_rows) * 2 + l31 * 16);
                const i32x4f_t rsc = wsc ? rws : rxs;
                int ksc = 2 * (wsc ? wstep : xstep);            // step 1's first slice
                const int scstep = 2 * (wsc ? wstep : xstep);
                const int lsc = __builtin_amdgcn_readfirstlane(lds0 + XRING + 2 * WS_ + (wsc ? 1024 : 0));
                const int axs = lds0 + XRING + 2 * WS_ + (wm * 128 + frow) * 2 + fhalf;
                const int aws = lds0 + XRING + 2 * WS_ + (wn * 32 * FN + frow) * 2 + fhalf;
#include "gemm_asm_f8_n3s.inc"
            } else {
                int ksx1, ksw1;
                int ksx = 2 * xstep, ksw = 2 * wstep;           // scalar offsets of step 1's first slice
                if constexpr (FN == 3) {
#include "gemm_asm_f8_n3.inc"
                } else {
#include "gemm_asm_f8.inc"
                }
                (void)ksx1; (void)ksw1;
            }
            (void)xwl; (void)dlt; (void)kofx;
        }
        const int next = tile + (int)gridDim.x;
        const bool more = next < ntiles;
        const int mw = cur.m0 + wm * 128, nw = cur.n0 + wn * 32 * FN;
        if constexpr (STAGED) {
            float4 bias[FN][4];
            load_colvec<FN>(a.g.bias, nw, lane >> 5, N, bias);
            if (more) { setup(next, nxt); scales0(nxt); stage_x(nxt, 0, 0); stage_w(nxt, 0, 0); }
            constexpr int MYB = EPI == EPI_MXFP8 ? StagedF8<FN>::BYTES : (EPI == EPI_QK8 ? StagedQK8<FN>::BYTES : StagedEpi<FN, EPI == EPI_QK8 ? EPI_QK : EPI>::BYTES);
            static_assert(4 * MYB <= 2 * XT, "epilogue staging must fit X slots 1 and 2");
            char* my = smem + XT + wave * MYB;   // X slots 1 and 2
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                f32x16_t blk[FN];
#pragma unroll
                for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v;
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[fm][fn][r]));
                        blk[fn][r] = v;
                    }
                if constexpr (EPI == EPI_MXFP8) epilogue_mxfp8_rows32<FN, ACT>(a, blk, bias, mw + fm * 32, nw, my, lane);
                else if constexpr (EPI == EPI_QK8) epilogue_qk8_rows32<FN>(a, blk, bias, mw + fm * 32, nw, my, lane);
                else epilogue_rows32<FN, EPI, ACT, FMT_BF16>(a.g, blk, bias, mw + fm * 32, nw, my, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            epilogue_direct<FM, FN, EPI, ACT>(a.g, acc, mw, nw, lane);
            if (more) { setup(next, nxt); scales0(nxt); stage_x(nxt, 0, 0); stage_w(nxt, 0, 0); }
        }
        if (!more) break;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // the staging area (X slots 1, 2) is free again
        stage_x(nxt, 1, 1); stage_w(nxt, 1, 1);
        stage_x(nxt, 2, 2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW + NXW) : "memory");
        cur = nxt;
        tile = next;
    }
}

template <int EPI, int ACT, int FN, bool SCL = false>
static int launch_f8a(const GemmF8Args& a, hipStream_t s) {
    constexpr int BN = 64 * FN;
    constexpr int LDS = (3 * 256 + 2 * BN) * 128 + (SCL ? 4096 : 0);      // (256x192: + the double-buffered scale image)
    const int tiles = (a.g.M / 256) * (a.g.N / BN);
    static PerDeviceOnce attr_once;
    auto kern = gemmf8_kernel<EPI, ACT, FN, SCL>;
    if (attr_once.need()) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    int grid = tiles;
    if (a.g.tune_persist >= 0 && tiles > 256) grid = 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// which tile (0 = none) the asm loop has for this launch: whole tiles, K a multiple of 256 and >= 512, scale pitches covering the tile
int gemm_asm_f8_tile(int epi, const GemmF8Args& a) {
    if (a.g.K % 256 != 0 || a.g.K < 512 || a.g.M % 256 != 0 || (a.ldx8 & 15)) return 0;
    if (a.xs_rows < a.g.M || a.ws_rows < a.g.N) return 0;
    if (epi != EPI_MXFP8 && epi != EPI_F32 && epi != EPI_F32_RESLN && epi != EPI_QK && epi != EPI_QK8) return 0;
    if (epi == EPI_QK8) return a.g.N == 3 * SYL_HIDDEN && a.g.Tp % 32 == 0 ? 91 : 0;       // the 256x192 tile only
    const bool n256 = a.g.N % 256 == 0, n192 = a.g.N % 192 == 0;
    if (!n256 && !n192) return 0;
    if (n256 && n192) {                                       // fewer rounds over 256 CUs wins (ties: the larger tile)
        const long t2 = (long)(a.g.M / 256) * (a.g.N / 256), t3 = (long)(a.g.M / 256) * (a.g.N / 192);
        const double c2 = (double)((t2 + 255) / 256) * 256 * 256, c3 = (double)((t3 + 255) / 256) * 256 * 192;
        return c2 <= c3 ? 85 : 91;
    }
    return n256 ? 85 : 91;
}

int launch_gemm_asm_f8(int epi, const GemmF8Args& a, hipStream_t s, int tile) {
    // 91 = 92 = the 256x192 tile with the scales through LDS; 93 = the same tile with per-lane scale loads (A/B: never faster)
    const bool scl = tile == 92 || tile == 91;
    if (tile == 92 || tile == 93) tile = 91;
    if (tile != 85 && tile != 91) { syl_set_error("launch_gemm_asm_f8", "tile must be 85 (256x256) or 91 (256x192)"); return 1; }
    if (tile == 85) switch (epi) {
        case EPI_MXFP8: return a.g.act == 1 ? launch_f8a<EPI_MXFP8, 1, 4>(a, s) : launch_f8a<EPI_MXFP8, 0, 4>(a, s);
        case EPI_F32: return a.g.act == 1 ? launch_f8a<EPI_F32, 1, 4>(a, s) : launch_f8a<EPI_F32, 0, 4>(a, s);
        case EPI_F32_RESLN: return launch_f8a<EPI_F32_RESLN, 0, 4>(a, s);
        case EPI_QK: return launch_f8a<EPI_QK, 0, 4>(a, s);
    }
    else if (scl) switch (epi) {                             // 256x192, scales through LDS
        case EPI_MXFP8: return a.g.act == 1 ? launch_f8a<EPI_MXFP8, 1, 3, true>(a, s) : launch_f8a<EPI_MXFP8, 0, 3, true>(a, s);
        case EPI_F32: return a.g.act == 1 ? launch_f8a<EPI_F32, 1, 3, true>(a, s) : launch_f8a<EPI_F32, 0, 3, true>(a, s);
        case EPI_F32_RESLN: return launch_f8a<EPI_F32_RESLN, 0, 3, true>(a, s);
        case EPI_QK: return launch_f8a<EPI_QK, 0, 3, true>(a, s);
        case EPI_QK8: return launch_f8a<EPI_QK8, 0, 3, true>(a, s);
    }
    else switch (epi) {
        case EPI_MXFP8: return a.g.act == 1 ? launch_f8a<EPI_MXFP8, 1, 3>(a, s) : launch_f8a<EPI_MXFP8, 0, 3>(a, s);
        case EPI_F32: return a.g.act == 1 ? launch_f8a<EPI_F32, 1, 3>(a, s) : launch_f8a<EPI_F32, 0, 3>(a, s);
        case EPI_F32_RESLN: return launch_f8a<EPI_F32_RESLN, 0, 3>(a, s);
        case EPI_QK: return launch_f8a<EPI_QK, 0, 3>(a, s);
    }
    syl_set_error("launch_gemm_asm_f8", "unsupported epilogue");
    return 1;
}
