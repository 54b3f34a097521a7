// Tile 47: tile 97's geometry (256x256 on eight waves, X3 ring, persistent) with the K loop on v_mfma_f32_16x16x32 (round 6).
//
// Same contract and same operands as launch_gemm_asm's tile 97 (y = act(x W^T + b) as 16-bit rows: the GELU GEMMs of the path -- conv1-5
// TP:154-213, FFN1 TP:347-368 reached from sylber.py:122), same LDS image (128-byte rows, source-side chunk swizzle), same LDS-DMA pieces,
// same ring and tile walk as gemmb_bf16_kernel<.., 2, 8, true>.  What differs is the MFMA shape of the loop (tools/gen_gemm_asm.py "Y3"):
// the vendor library's kernels on this part issue 16x16x32, and profiles/r06_mfma_shape.md measured that shape 8.5 % cheaper per FLOP
// under the package power cap as a register-only stream.  This kernel is the same question asked of the real loop.
//
// An output element's fp32 chain adds 32-k blocks here and 16-k blocks in every other 16-bit GEMM kernel of the library, so results agree
// with them to fp32 rounding, NOT bit for bit: the tile is forced-only (SYLBER_OPT_GEMM_TILE = 47, sylber_op_linear cfg 47), the cost
// model never picks it, and tests hold it to the torch reference (tests/test_gpu_ops.py::test_tile47_mfma16).
//
// C layout of v_mfma_f32_16x16x32 with A = W fragment, B = X fragment: lane (t = lane & 15, q = lane >> 4) owns token t of the X fragment and
// output columns 4 q + e (e < 4) of the W fragment.  A wave's 128 x 64 tile = 8 X fragments x 4 W fragments = 32 accumulator quads.
#include "kernels.h"
#include "gemm_epilogue.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void* lds_vptr_c;
typedef int i32x4c_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16_c(__amdgpu_buffer_rsrc_t rs, int voff, int soff, void* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr_c)l, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ i32x4c_t rsrc_words_c(const void* base) {
    const unsigned long long p = (unsigned long long)base;
    i32x4c_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(p >> 32) & 0xffffu));
    r.z = (int)0xffffffffu;
    r.w = 0x00020000;
    return r;
}

// one 32-row block of a wave's tile through the wave's private LDS region (as epilogue_rows32, EPI_BF16): blk[th][j] = the accumulator quad of
// X fragment (16 th + t) x W fragment j; bias[j] = a.bias[ncol0 + 16 j + 4 q .. + 3]
template <int GW, int ACT, int FMT>
__device__ __forceinline__ void epilogue_rows32_m16(const GemmArgs& a, const f32x4_t (&blk)[2][GW], const float4 (&bias)[GW], int mrow0, int ncol0,
                                                    char* lds, int lane) {
    constexpr int ROWB = 16 * GW * 2, RS = ROWB + 16, CH = ROWB / 16;
    const int t = lane & 15, qd = lane >> 4;
#pragma unroll
    for (int th = 0; th < 2; ++th)
#pragma unroll
        for (int j = 0; j < GW; ++j) {
            float v0 = blk[th][j][0] + bias[j].x, v1 = blk[th][j][1] + bias[j].y, v2 = blk[th][j][2] + bias[j].z, v3 = blk[th][j][3] + bias[j].w;
            apply_act4<ACT>(v0, v1, v2, v3);
            uint2 pk; pk.x = H16<FMT>::pack2(v0, v1); pk.y = H16<FMT>::pack2(v2, v3);
            *(uint2*)(lds + (16 * th + t) * RS + (16 * j + 4 * qd) * 2) = pk;
        }
#pragma unroll
    for (int it = 0; it < CH / 2; ++it) {
        const int idx = it * 64 + lane;
        const int r = idx / CH, c = idx - r * CH;
        const int mo = mrow0 + r;
        const int n = ncol0 + c * 8;
        if (mo >= a.M || n >= a.N) continue;
        const uint4 raw = *(const uint4*)(lds + r * RS + c * 16);
        *(uint4*)((bf16_t*)a.out0 + (size_t)mo * a.ld0 + n) = raw;
    }
}

// EPI_F32 (plain fp32 rows: the op-level test entry point): straight from the registers, a float4 per accumulator quad
template <int GW, int ACT>
__device__ __forceinline__ void epilogue_rows32_m16_f32(const GemmArgs& a, const f32x4_t (&blk)[2][GW], const float4 (&bias)[GW], int mrow0, int ncol0, int lane) {
    const int t = lane & 15, qd = lane >> 4;
#pragma unroll
    for (int th = 0; th < 2; ++th)
#pragma unroll
        for (int j = 0; j < GW; ++j) {
            float v0 = blk[th][j][0] + bias[j].x, v1 = blk[th][j][1] + bias[j].y, v2 = blk[th][j][2] + bias[j].z, v3 = blk[th][j][3] + bias[j].w;
            apply_act4<ACT>(v0, v1, v2, v3);
            const int m = mrow0 + 16 * th + t, n = ncol0 + 16 * j + 4 * qd;
            if (m < a.M && n < a.N) *(float4*)((float*)a.out0 + (size_t)m * a.ld0 + n) = make_float4(v0, v1, v2, v3);
        }
}

// FMV = 32-row blocks per wave: 4 = tile 47 (256x256), 3 = tile 46 (192x256: the sibling tile 57 is to tile 97, for launches whose 256-row tile
// count leaves a partial last round; gemm_asm.hip)
template <int EPI, int ACT, int FMT, bool TAP, int FMV = 4>
__global__ __launch_bounds__(512, 2) void gemmc_bf16_kernel(const GemmArgs a) {
    static_assert(EPI == EPI_BF16 || EPI == EPI_F32, "tiles 46 / 47: 16-bit rows (staged) or fp32 rows (direct)");
    static_assert(FMV == 4 || (FMV == 3 && !TAP), "192-row tile: linear K order only");
    constexpr int FM = FMV, FN = 2, NW = 8, WN = 4, BM = 64 * FM, BN = 256, RB = 128;
    constexpr int GX = 2 * FM, GW = 2 * FN;                  // 16-row X fragments / 16-column W fragments per wave
    constexpr int XT = BM * RB, WS = BN * RB;
    constexpr int XRING = 3 * XT;
    constexpr int NPW = (BM + BN) / 8 / NW, NXW = BM / 8 / NW;
    constexpr int EPB = 32 * (16 * GW * 2 + 16);              // staging bytes per wave (one 32-row block)
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M - a.m_begin + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;

    const int srow = lane >> 3, spos = lane & 7;
    struct Tile { int m0, n0; int voff[NPW]; i32x4c_t rx, rw; };
    auto setup = [&](int tile_id, Tile& t) {
        const int wg = xcd_remap(tile_id, ntiles);
        t.m0 = a.m_begin + (wg / tiles_n) * BM;
        t.n0 = (wg % tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int p = wave + NW * i;
            const bool isx = i < NXW;
            const int r = (isx ? p : p - BM / 8) * 8 + srow;
            const int c = spos ^ ((r >> 1) & 7);
            if (isx) { int xm = t.m0 + r; xm = xm < a.M ? xm : a.M - 1; t.voff[i] = (int)(((long)(xm - t.m0) * a.ldx + c * 8) * 2); }
            else { int wr = t.n0 + r; wr = wr < a.N ? wr : a.N - 1; t.voff[i] = ((wr - t.n0) * a.K + c * 8) * 2; }
        }
        t.rx = rsrc_words_c(a.X + (size_t)t.m0 * a.ldx);
        t.rw = rsrc_words_c(a.W + (size_t)t.n0 * a.K);
    };
    auto stage_x = [&](const Tile& t, int s, int slot) {
        const __amdgpu_buffer_rsrc_t bx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + (size_t)t.m0 * a.ldx), 0, (int)0xffffffffu, 0x00020000);
#pragma unroll
        for (int i = 0; i < NXW; ++i)
            glds16_c(bx, t.voff[i], TAP ? tap3_offset(s * 128) : s * 128, smem + slot * XT + (wave + NW * i) * 1024);
    };
    auto stage_w = [&](const Tile& t, int s, int slot) {
        const __amdgpu_buffer_rsrc_t bw = __builtin_amdgcn_make_buffer_rsrc((void*)(a.W + (size_t)t.n0 * a.K), 0, (int)0xffffffffu, 0x00020000);
#pragma unroll
        for (int i = NXW; i < NPW; ++i)
            glds16_c(bw, t.voff[i], s * 128, smem + XRING + slot * WS + (wave + NW * (i - NXW)) * 1024);
    };
    auto stage = [&](const Tile& t, int s) { stage_x(t, s, s); stage_w(t, s, s); };
    const int lds0 = (int)(unsigned)(unsigned long long)(lds_vptr_c)smem;
    const int lbase = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);

    // ---- fragment addresses: lane (t, q) reads row t of a 16-row fragment, 16-byte chunk (4 h + q) ^ swizzle(row) of slice h
    const int frow = lane & 15, fq = lane >> 4, swz = (frow >> 1) & 7;
    int ax[2], aw[2], awh[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ko = ((4 * h + fq) ^ swz) << 4;
        ax[h] = lds0 + (wm * 32 * FM + frow) * RB + ko;
        aw[h] = lds0 + XRING + (wn * 32 * FN + frow) * RB + ko;
        awh[h] = aw[h] + WS;
    }

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    const bool whole_tiles = (a.M - a.m_begin) % BM == 0 && a.N % BN == 0;
    Tile cur, nxt;
    setup(tile, cur);
    stage(cur, 0);
    stage(cur, 1);
    stage_x(cur, 2, 2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW + NXW) : "memory");      // my pieces of step 0 (K >= 256 and K % 128 == 0: launcher)
    for (;;) {
        __builtin_amdgcn_s_barrier();
        f32x4_t acc[GX][GW];
        int koff = 2 * 128;
        int nloop = (a.K / 64 - 4) / 2;
        bf16x8_t fx[GX], fw[2][GW];
        {
            const int (&voff)[NPW] = cur.voff;
            i32x4c_t rx, rw;
            rx.x = __builtin_amdgcn_readfirstlane(cur.rx.x); rx.y = __builtin_amdgcn_readfirstlane(cur.rx.y); rx.z = (int)0xffffffffu; rx.w = 0x00020000;
            rw.x = __builtin_amdgcn_readfirstlane(cur.rw.x); rw.y = __builtin_amdgcn_readfirstlane(cur.rw.y); rw.z = (int)0xffffffffu; rw.w = 0x00020000;
            int axc[2] = {ax[0], ax[1]};
            int xr = 0, xwl, dlt, kofx;
            const int cneg = -2 * XT;
            [[maybe_unused]] int ph = 0, dk;
            [[maybe_unused]] const int c2048 = 2048, cm1024 = -1024, cm896 = -896;
            if constexpr (TAP) kofx = tap3_offset(3 * 128);
            if constexpr (FMT == FMT_F16) {
#define MF "v_mfma_f32_16x16x32_f16"
                if constexpr (FMV == 3) {
#include "gemm_asm_y3_m3_w8.inc"
                } else if constexpr (TAP) {
#include "gemm_asm_y3_w8_t.inc"
                } else {
#include "gemm_asm_y3_w8.inc"
                }
#undef MF
            } else {
#define MF "v_mfma_f32_16x16x32_bf16"
                if constexpr (FMV == 3) {
#include "gemm_asm_y3_m3_w8.inc"
                } else if constexpr (TAP) {
#include "gemm_asm_y3_w8_t.inc"
                } else {
#include "gemm_asm_y3_w8.inc"
                }
#undef MF
            }
            (void)xwl; (void)dlt; (void)kofx; (void)dk;
        }
        const int next = tile + (int)gridDim.x;
        const bool more = next < ntiles;
        const int mw = cur.m0 + wm * 32 * FM, nw = cur.n0 + wn * 32 * FN;
        static_assert(NW * EPB <= 2 * XT, "epilogue staging must fit the X slots the prefetch leaves alone");
        float4 bias[GW];
#pragma unroll
        for (int j = 0; j < GW; ++j) {
            int n = nw + 16 * j + 4 * fq;
            n = n < a.N - 4 ? n : a.N - 4;
            bias[j] = a.bias ? *(const float4*)(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (more) { setup(next, nxt); stage(nxt, 0); }
        char* my = smem + XT + wave * EPB;                        // X slots 1 and 2
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            f32x4_t blk[2][GW];
#pragma unroll
            for (int th = 0; th < 2; ++th)
#pragma unroll
                for (int j = 0; j < GW; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v;
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[2 * fm + th][j][r]));
                        blk[th][j][r] = v;
                    }
            if constexpr (EPI == EPI_BF16) epilogue_rows32_m16<GW, ACT, FMT>(a, blk, bias, mw + fm * 32, nw, my, lane);
            else epilogue_rows32_m16_f32<GW, ACT>(a, blk, bias, mw + fm * 32, nw, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!more) break;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage(nxt, 1);
        stage_x(nxt, 2, 2);
        constexpr int NYOUNG = NPW + NXW;
        constexpr int NST = FM * ((16 * GW * 2 / 16) / 2);
        static_assert(NYOUNG + NST < 64, "vmcnt is a 6-bit counter");
        if (EPI == EPI_BF16 && whole_tiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NYOUNG + NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NYOUNG) : "memory");
        cur = nxt;
        tile = next;
    }
}

template <int EPI, int ACT, int FMT, bool TAP, int FMV = 4>
static int launch_c(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 64 * FMV;
    constexpr int LDS = (3 * BM + 2 * 256) * 128;
    const int tiles = ((a.M - a.m_begin + BM - 1) / BM) * ((a.N + 255) / 256);
    static PerDeviceOnce attr_once;
    auto kern = gemmc_bf16_kernel<EPI, ACT, FMT, TAP, FMV>;
    if (attr_once.need()) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    int grid = tiles;
    if (a.tune_persist >= 0 && tiles > 256) grid = 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The small tiles of the 16x16x32 family, hipcc-scheduled: gemm_bf16_kernel's scheme (gemm_bf16.hip: four waves 2 x 2, K step 64 = 128-byte LDS
// rows, two-slot ring, TWO workgroups per CU, buffer-descriptor LDS-DMA with the chunk swizzle on the source address) with 16-row fragments:
// tile ids 13 = 128x128, 14 = 128x192 (what ids 3 / 4 are to the 32x32x16 kernels).  They serve the launches the big tiles cannot fill (small
// batches, ragged tails) with the SAME fp32 chain per output element as tiles 46 / 47: slice 0 (k 0..31) then slice 1 (k 32..63) of every K
// step, one 16x16x32 MFMA each, so a launch's bits do not depend on which member of the family the cost model picks (tests/test_gpu_ops.py).
template <int FMT> struct MF16 {
    static __device__ __forceinline__ f32x4_t mfma(bf16x8_t a, bf16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MF16<FMT_F16> {
    static __device__ __forceinline__ f32x4_t mfma(bf16x8_t a, bf16x8_t b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
template <int N> __device__ __forceinline__ void wait_vmcnt_c() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#define SCHED_FENCE_C() __builtin_amdgcn_sched_barrier(0)

// register -> global epilogue of a wave tile of GX x GW accumulator quads (EPI_BF16: 16-bit rows, EPI_F32: fp32 rows)
template <int GX, int GW, int EPI, int ACT, int FMT>
__device__ __forceinline__ void epilogue_direct_m16(const GemmArgs& a, const f32x4_t (&acc)[GX][GW], int mrow0, int ncol0, int lane) {
    const int t = lane & 15, qd = lane >> 4;
    float4 bias[GW];
    bool nok[GW];
    int ncl[GW];
#pragma unroll
    for (int j = 0; j < GW; ++j) {
        const int n = ncol0 + 16 * j + 4 * qd;
        nok[j] = n < a.N;
        ncl[j] = n < a.N - 4 ? n : a.N - 4;
        bias[j] = a.bias ? *(const float4*)(a.bias + ncl[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < GX; ++i) {
        const int m = mrow0 + 16 * i + t;
        const bool mok = m < a.M;
        const size_t ro = (size_t)(mok ? m : a.M - 1) * a.ld0;
#pragma unroll
        for (int j = 0; j < GW; ++j) {
            float v0 = acc[i][j][0] + bias[j].x, v1 = acc[i][j][1] + bias[j].y, v2 = acc[i][j][2] + bias[j].z, v3 = acc[i][j][3] + bias[j].w;
            apply_act4<ACT>(v0, v1, v2, v3);
            if constexpr (EPI == EPI_BF16) {
                uint2 pk; pk.x = H16<FMT>::pack2(v0, v1); pk.y = H16<FMT>::pack2(v2, v3);
                if (mok && nok[j]) *(uint2*)((bf16_t*)a.out0 + ro + ncl[j]) = pk;
            } else {
                if (mok && nok[j]) *(float4*)((float*)a.out0 + ro + ncl[j]) = make_float4(v0, v1, v2, v3);
            }
        }
    }
}

// NW = 4: waves 2 x 2, wave tile 32 FM x 32 FN (tiles 13 / 14).  NW = 8 (tiles 15 / 16, round 6): waves 2 x 4, wave tile 32 FM x 16 FN -- the SAME tile and the
// same chain order on twice the waves: a workgroup that is ALONE on its CU (small batches: fewer tiles than CUs) is bound by what ONE wave per SIMD has to issue per
// K step -- eight LDS-DMA pieces at 100+ issue cycles each beside 32 sixteen-cycle MFMAs and 16 fragment reads (~2 200 cycles per K step measured against 512 of
// MFMA time); eight waves halve every per-wave count and put two waves on each SIMD
template <int FM, int FN, int EPI, int ACT, int FMT, int NW = 4, int NSTAGE = 2>
__device__ __forceinline__ void gemm16_tile(const GemmArgs& a, const int tile_id, char* smem) {
    constexpr int BM = 64 * FM, BN = 64 * FN, RB = 128;
    constexpr int WNN = NW / 2;                              // waves along n
    constexpr int GX = 2 * FM, GW = 4 * FN / WNN;            // 16-row fragments per wave (wave tile 32 FM x (64 FN / WNN))
    constexpr int XT = BM * RB, WT = BN * RB, STAGE = XT + WT;
    constexpr int NP = (BM + BN) / 8, NPW = NP / NW, XPW = BM / 8 / NW;
    static_assert(NP % NW == 0 && (BM / 8) % NW == 0 && (4 * FN) % WNN == 0, "pieces / fragments must split evenly over the waves");
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave / WNN, wn = wave % WNN;
    const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M - a.m_begin + BM - 1) / BM;
    const int wg = xcd_remap(tile_id, tiles_m * tiles_n);
    const int m0 = a.m_begin + (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;

    const int srow = lane >> 3, spos = lane & 7;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + (size_t)m0 * a.ldx), 0, (int)0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(a.W + (size_t)n0 * a.K), 0, (int)0xffffffffu, 0x00020000);
    int voff[NPW], lds_off[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int p = wave + NW * i;
        const bool isx = i < XPW;
        const int r = (isx ? p : p - BM / 8) * 8 + srow;
        const int c = spos ^ ((r >> 1) & 7);
        if (isx) { int xm = m0 + r; xm = xm < a.M ? xm : a.M - 1; voff[i] = (int)(((long)(xm - m0) * a.ldx + c * 8) * 2); }
        else { int wr = n0 + r; wr = wr < a.N ? wr : a.N - 1; voff[i] = ((wr - n0) * a.K + c * 8) * 2; }
        lds_off[i] = (isx ? 0 : XT) + (isx ? p : p - BM / 8) * 1024;
    }
    const int nt = a.K / 64;
    auto dma = [&](int kt, int i, char* base) {
        const bool isx = i < XPW;
        glds16_c(isx ? rx : rw, voff[i], (a.kpat && isx) ? tap3_offset(kt * 128) : kt * 128, base + lds_off[i]);
    };
    auto stage = [&](int kt, int slot) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) dma(kt, i, smem + slot * STAGE);
    };
    const int frow = lane & 15, fq = lane >> 4, swz = (frow >> 1) & 7;
    int koff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) koff[h] = ((4 * h + fq) ^ swz) << 4;
    const int xrow_off = (wm * 32 * FM + frow) * RB, wrow_off = XT + (wn * 16 * GW + frow) * RB;

    f32x4_t acc[GX][GW];
#pragma unroll
    for (int i = 0; i < GX; ++i)
#pragma unroll
        for (int j = 0; j < GW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t xf[2][GX], wf[2][GW];
    auto read_frags = [&](const char* sb, int h, int buf) {
#pragma unroll
        for (int f = 0; f < GX; ++f) xf[buf][f] = *(const bf16x8_t*)(sb + xrow_off + f * 16 * RB + koff[h]);
#pragma unroll
        for (int f = 0; f < GW; ++f) wf[buf][f] = *(const bf16x8_t*)(sb + wrow_off + f * 16 * RB + koff[h]);
    };
    // MFMAs of X fragments [i0, i1) of one slice with LDS-DMA pieces [p0, p1) of k-tile kt (-> ring slot dslot) spread between them
    auto mfmas_dma = [&](int buf, int i0, int i1, int kt, int dslot, int p0, int p1, bool on) {
        const int NM = (i1 - i0) * GW, np = p1 - p0;
        char* base = smem + dslot * STAGE;
#pragma unroll
        for (int i = 0; i < GX; ++i) {
            if (i < i0 || i >= i1) continue;
#pragma unroll
            for (int j = 0; j < GW; ++j) {
                const int n = (i - i0) * GW + j;
                acc[i][j] = MF16<FMT>::mfma(wf[buf][j], xf[buf][i], acc[i][j]);
#pragma unroll
                for (int p = 0; p < NPW; ++p) {
                    if (np > 0 && p >= p0 && p < p1 && ((p - p0 + 1) * NM + np - 1) / np - 1 == n) {
                        SCHED_FENCE_C();
                        if (on) dma(kt, p, base);
                        SCHED_FENCE_C();
                    }
                }
            }
        }
    };

    stage(0, 0);
    if (nt > 1) stage(1, 1);
    if constexpr (NSTAGE == 3) { if (nt > 2) stage(2, 2); }
    if (NSTAGE == 3 && nt > 2) wait_vmcnt_c<2 * NPW>();
    else if (nt > 1) wait_vmcnt_c<NPW>();
    else wait_vmcnt_c<0>();
    __builtin_amdgcn_s_barrier();
    read_frags(smem, 0, 0);
    int slot = 0;
    // pieces [0, D0) of the next tile to request are issued behind the barrier of tile t (under the second half of its slice 1), the rest under slice 0 of tile t+1
    constexpr int D0 = (NPW * 2 + 4) / 5;
    // TWO-slot ring: the tile requested is t+2 (into the slot tile t just left), waited for -- all of it -- at the barrier of tile t+1: half a K step after its
    // last piece.  THREE-slot ring (NSTAGE = 3, round 6: the 64x64 tile of small batches): the tile requested is t+3, tile t+2 stays in flight across the barrier
    // (counted wait), so a piece is a step and a half old when it is needed -- what a LONE workgroup, with nobody to hide the latency behind, is bound by.
    // One K tile.  STEADY: every "does tile t+k exist" test is true, so the loop that runs nearly all tiles carries no branch around its LDS-DMA instructions
    auto ktile = [&](auto steady, int t) {
        constexpr bool STEADY = decltype(steady)::value;
        constexpr int A = NSTAGE - 1;                        // tiles ahead: the request issued around tile t is for tile t + A (+ 1 behind the barrier)
        const char* sb = smem + slot * STAGE;
        const int nslot = slot == NSTAGE - 1 ? 0 : slot + 1;
        const int rslot = NSTAGE == 2 ? nslot : (nslot == NSTAGE - 1 ? 0 : nslot + 1);      // slot of tile t + A while tile t is consumed
        const bool has_next = STEADY || t + 1 < nt, cont = STEADY || (t >= 1 && t + A < nt), dma2 = STEADY || t + A + 1 < nt;
        SCHED_FENCE_C();
        read_frags(sb, 1, 1);
        SCHED_FENCE_C();
        mfmas_dma(0, 0, GX, t + A, rslot, D0, NPW, cont);
        mfmas_dma(1, 0, GX / 2, 0, 0, 0, 0, false);
        SCHED_FENCE_C();
        if (has_next) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (NSTAGE == 3 && (STEADY || t + 2 < nt)) wait_vmcnt_c<NPW>();      // tile t+2 (all of it requested by now) may stay in flight
            else wait_vmcnt_c<0>();
            __builtin_amdgcn_s_barrier();
            SCHED_FENCE_C();
            read_frags(smem + nslot * STAGE, 0, 0);
            SCHED_FENCE_C();
        }
        mfmas_dma(1, GX / 2, GX, t + A + 1, slot, 0, D0, dma2);
        slot = nslot;
    };
    {
        int t = 0;
        if (nt > 0) ktile(std::false_type{}, t++);
        for (; t < nt - NSTAGE; ++t) ktile(std::true_type{}, t);
        for (; t < nt; ++t) ktile(std::false_type{}, t);
    }
    epilogue_direct_m16<GX, GW, EPI, ACT, FMT>(a, acc, m0 + wm * 32 * FM, n0 + wn * 16 * GW, lane);
}

template <int FM, int FN, int EPI, int ACT, int FMT, int NW = 4, int NSTAGE = 2>
__global__ __launch_bounds__(64 * NW, 2) void gemm16_bf16_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    constexpr int BM = 64 * FM, BN = 64 * FN;
    const int ntiles = ((a.M - a.m_begin + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        gemm16_tile<FM, FN, EPI, ACT, FMT, NW, NSTAGE>(a, tile, smem);
        if (tile + (int)gridDim.x < ntiles) __builtin_amdgcn_s_barrier();
    }
}

template <int FM, int FN, int EPI, int ACT, int FMT, int NW = 4, int NSTAGE = 2>
static int launch_s(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 64 * FM, BN = 64 * FN;
    constexpr int LDS = NSTAGE * (BM + BN) * 128;
    const int tiles = ((a.M - a.m_begin + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    static PerDeviceOnce attr_once;
    auto kern = gemm16_bf16_kernel<FM, FN, EPI, ACT, FMT, NW, NSTAGE>;
    if (attr_once.need()) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    int grid = tiles;
    const int per_cu = a.tune_persist > 0 ? a.tune_persist : (a.tune_persist == 0 ? 2 : 0);
    if (per_cu > 0 && tiles > per_cu * 256) grid = per_cu * 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- the family's dispatch ------------------------------------------------------------------------------------------------------
bool gemm_asm16_has_tile(int epi, const GemmArgs& a, int tile) {
    if (tile != 13 && tile != 14 && tile != 15 && tile != 16 && tile != 17 && tile != 46 && tile != 47) return false;
    if ((epi != EPI_BF16 && epi != EPI_F32) || (a.act != 0 && a.act != 1)) return false;
    if (a.fmt != FMT_BF16 && a.fmt != FMT_F16) return false;
    if (epi == EPI_F32 && (a.fmt != FMT_BF16 || a.kpat)) return false;
    if (a.K % 64 != 0 || a.K < 64) return false;
    if (a.kpat && a.K != 1536) return false;
    if (tile == 13 || tile == 14 || tile == 15 || tile == 16 || tile == 17) return true;
    if (a.K % 128 != 0 || a.K < 256) return false;         // the generated loops: pairs of K steps, at least four
    if (tile == 46 && a.kpat) return false;
    return true;
}

template <int EPI, int ACT, int FMT>
static int launch_family(int tile, const GemmArgs& a, hipStream_t s) {
    if (tile == 13) return launch_s<2, 2, EPI, ACT, FMT>(a, s);
    if (tile == 14) return launch_s<2, 3, EPI, ACT, FMT>(a, s);
    if (tile == 15) return launch_s<2, 2, EPI, ACT, FMT, 8>(a, s);
    if (tile == 16) return launch_s<2, 3, EPI, ACT, FMT, 8>(a, s);
    if (tile == 17) return launch_s<1, 1, EPI, ACT, FMT, 4, 3>(a, s);      // 64x64, three-slot ring: the launches of one or two clips (gemm_bf16.hip tile 1)
    if (tile == 46) return launch_c<EPI, ACT, FMT, false, 3>(a, s);
    if constexpr (EPI == EPI_BF16) { if (a.kpat) return launch_c<EPI, ACT, FMT, true>(a, s); }
    return launch_c<EPI, ACT, FMT, false>(a, s);
}

int launch_gemm_asm16(int epi, const GemmArgs& a, hipStream_t s) {
    const int tile = a.tune_cfg - 1;
    if (!gemm_asm16_has_tile(epi, a, tile)) { syl_set_error("launch_gemm_asm16", "no 16x16x32 instantiation for this tile / epilogue / K"); return 1; }
    if (epi == EPI_F32) return a.act == 1 ? launch_family<EPI_F32, 1, FMT_BF16>(tile, a, s) : launch_family<EPI_F32, 0, FMT_BF16>(tile, a, s);
    if (a.fmt == FMT_F16) return a.act == 1 ? launch_family<EPI_BF16, 1, FMT_F16>(tile, a, s) : launch_family<EPI_BF16, 0, FMT_F16>(tile, a, s);
    return a.act == 1 ? launch_family<EPI_BF16, 1, FMT_BF16>(tile, a, s) : launch_family<EPI_BF16, 0, FMT_BF16>(tile, a, s);
}
