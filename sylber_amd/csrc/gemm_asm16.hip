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

template <int EPI, int ACT, int FMT, bool TAP>
__global__ __launch_bounds__(512, 2) void gemmc_bf16_kernel(const GemmArgs a) {
    static_assert(EPI == EPI_BF16 || EPI == EPI_F32, "tile 47: 16-bit rows (staged) or fp32 rows (direct)");
    constexpr int FM = 4, FN = 2, NW = 8, WN = 4, BM = 256, BN = 256, RB = 128;
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
                if constexpr (TAP) {
#include "gemm_asm_y3_w8_t.inc"
                } else {
#include "gemm_asm_y3_w8.inc"
                }
#undef MF
            } else {
#define MF "v_mfma_f32_16x16x32_bf16"
                if constexpr (TAP) {
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

template <int EPI, int ACT, int FMT, bool TAP>
static int launch_c(const GemmArgs& a, hipStream_t s) {
    constexpr int LDS = (3 * 256 + 2 * 256) * 128;
    const int tiles = ((a.M - a.m_begin + 255) / 256) * ((a.N + 255) / 256);
    static PerDeviceOnce attr_once;
    auto kern = gemmc_bf16_kernel<EPI, ACT, FMT, TAP>;
    if (attr_once.need()) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    int grid = tiles;
    if (a.tune_persist >= 0 && tiles > 256) grid = 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

bool gemm_asm16_has_tile(int epi, const GemmArgs& a) {
    if ((epi != EPI_BF16 && epi != EPI_F32) || (a.act != 0 && a.act != 1)) return false;
    if (a.fmt != FMT_BF16 && a.fmt != FMT_F16) return false;
    if (epi == EPI_F32 && (a.fmt != FMT_BF16 || a.kpat)) return false;
    if (a.K % 128 != 0 || a.K < 256) return false;
    if (a.kpat && a.K != 1536) return false;
    return true;
}

int launch_gemm_asm16(int epi, const GemmArgs& a, hipStream_t s) {
    if (!gemm_asm16_has_tile(epi, a)) { syl_set_error("launch_gemm_asm16", "tile 47: 16-bit / fp32 rows (plain / GELU), K % 128 == 0, K >= 256"); return 1; }
    if (epi == EPI_F32) return a.act == 1 ? launch_c<EPI_F32, 1, FMT_BF16, false>(a, s) : launch_c<EPI_F32, 0, FMT_BF16, false>(a, s);
    if (a.fmt == FMT_F16) {
        if (a.kpat) return a.act == 1 ? launch_c<EPI_BF16, 1, FMT_F16, true>(a, s) : launch_c<EPI_BF16, 0, FMT_F16, true>(a, s);
        return a.act == 1 ? launch_c<EPI_BF16, 1, FMT_F16, false>(a, s) : launch_c<EPI_BF16, 0, FMT_F16, false>(a, s);
    }
    if (a.kpat) return a.act == 1 ? launch_c<EPI_BF16, 1, FMT_BF16, true>(a, s) : launch_c<EPI_BF16, 0, FMT_BF16, true>(a, s);
    return a.act == 1 ? launch_c<EPI_BF16, 1, FMT_BF16, false>(a, s) : launch_c<EPI_BF16, 0, FMT_BF16, false>(a, s);
}
