// bf16 MFMA GEMM for gfx950:  out[m][n] = sum_k X[m][k] * W[n][k]  with fused epilogues.
//
// Covers the dense contractions of the Segmenter hot path (reference call site
// sylber/model/sylber.py:122 -> transformers HubertModel): the six 512->512 strided Conv1d layers
// (TP:112-124, as implicit GEMM on channels-last activations: ldx = stride*512, K = taps*512), the
// feature projection (TP:225-231), q/k/v/out projections (TP:318-342) and the FFN (TP:361-368).
//
// Two kernels share the staging scheme and the epilogues:
//   * gemm_bf16_kernel: 256 threads (4 waves as 2x2), tile (64 FM) x (64 FN), K step 64 (128-byte LDS rows), 2-slot
//     ring, TWO workgroups per CU (tiles 128x128 / 128x192): the second workgroup covers the first one's epilogue,
//     barrier and LDS-DMA issue time.  Used for the N = 768 / 1536 launches (500 / 1000 tiles = 1 / 2 rounds).
//   * gemm8_bf16_kernel (below): 512 threads, tile 256x256, K step 32, 4-slot ring, one workgroup per CU, the two
//     wave groups staggered by one barrier.  Used for the big GEMMs (convs, FFN1).
// Both operands are staged HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip); the loads of the next tile(s)
// are issued between the MFMAs of the current one and stay in flight ACROSS the per-tile barrier (raw s_barrier +
// counted s_waitcnt vmcnt(N), never vmcnt(0) in the steady state).  The LDS image is lane-linear per wave
// instruction (8 rows x 128 B), so the bank-conflict swizzle (16-B chunk ^= (row>>1)&7, conflict-free for
// ds_read_b128's 16-lane groups on 128-B rows; (row>>2)&3 on 64-B rows) is applied to the per-lane SOURCE address
// and again on the fragment read.  Fragments of k-substep kk+1 are read while the MFMAs of kk issue.  Tile shape is
// picked per launch by a measured cost model (rounds over 256 CUs x tile area / efficiency).  Workgroup ids are
// remapped so that each XCD owns a contiguous run of tiles (shared X panel in L2; same row ownership as the
// LayerNorm / attention launches between the GEMMs).
#include "kernels.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((glb_vptr)g, (lds_vptr)l, 16, 0, 0);
}

// LDS-DMA through a buffer descriptor: the per-lane part of the source address is a 32-bit byte offset that stays
// CONSTANT over the K loop (and, in the persistent kernels, over the tiles), the K step / operand plane travel in the scalar
// offset, the tile's first row in the descriptor base.  Against global_load_lds with a 64-bit per-lane address (two VALU
// adds per piece and step, twice the address registers) tools/ubench/mfma_mix2.hip measures the DMA's cost beside the
// MFMAs at +16 instead of +200 issue cycles per two pieces (profiles/r03_lds_dma_cost.md).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0xffffffffu, 0x00020000);
}
__device__ __forceinline__ void glds16b(__amdgpu_buffer_rsrc_t rs, int voff, int soff, void* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)l, 16, voff, soff, 0, 0);
}

#include "gemm_epilogue.h"

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// shader-clock timestamp (development, trace instantiation only): s_memtime shares lgkmcnt with the LDS reads, so a stamp
// sits only where the kernel waits lgkmcnt(0) anyway or has no LDS read in flight
#define TSTAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")

// WM x WN = the wave grid.  2 x 2 (four waves, wave tile 32 FM x 32 FN): tile ids 0 / 3 / 4.  4 x 2 (EIGHT waves, round 6: ids 5 = 128x128 with FM = 1, FN = 2 and
// 6 = 128x192 with FM = 1, FN = 3): the same tiles on twice the waves.  A workgroup that is alone on its CU -- every launch of a small batch -- is bound by what one
// wave per SIMD has to ISSUE per K step (eight LDS-DMA pieces at 100+ issue cycles each, 16 fragment reads, 16 MFMAs: ~2 200 cycles against 512 of MFMA time);
// eight waves halve every per-wave count and put two waves on each SIMD (profiles/r06_small_tiles.md).  Same chain order per output element: bit-identical.
template <int FM, int FN, int BK, int NSTAGE, bool PRIO, int EPI, int ACT, int FMT, int WM = 2, int WN = 2>
__device__ __forceinline__ void gemm_bf16_tile(const GemmArgs& a, const int tile_id, char* smem) {
    constexpr int NWV = WM * WN;
    constexpr int BM = 32 * FM * WM, BN = 32 * FN * WN;
    constexpr int RB = BK * 2;               // bytes per LDS row (128 or 64)
    constexpr int CPR = RB / 16;             // 16-B chunks per row
    constexpr int RPP = 1024 / RB;           // rows per 1-KiB staging piece (one wave instruction)
    constexpr int KK = BK / 16;              // MFMA k-substeps per stage
    constexpr int XT = BM * RB, WT = BN * RB, STAGE = XT + WT;
    constexpr int NP = (BM + BN) / RPP;      // 1-KiB pieces per stage
    constexpr int NPW = NP / NWV;            // pieces per wave
    static_assert(NP % NWV == 0, "tile must split evenly over the waves");
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;

    const int tiles_n = (a.N + BN - 1) / BN;
    const int tiles_m = (a.M - a.m_begin + BM - 1) / BM;
    const int wg = xcd_remap(tile_id, tiles_m * tiles_n);
    const int m0 = a.m_begin + (wg / tiles_n) * BM;
    const int n0 = (wg % tiles_n) * BN;

    // ---- staging: wave w owns pieces w, w+4, ...; piece p < BM/8 is X rows 8p.., else W rows
    const int srow = lane / CPR, spos = lane % CPR;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.X + (size_t)m0 * a.ldx), rw = make_rsrc(a.W + (size_t)n0 * a.K);
    int voff[NPW];                                           // byte offset of this lane's 16 bytes inside its operand's tile rows
    int lds_off[NPW];
    // pieces wave + 4 i with i < XPW are X rows for EVERY wave (the X pieces split evenly over the waves): which operand
    // a piece belongs to is a compile-time property of i, so its descriptor is too
    static_assert((BM / RPP) % NWV == 0, "X pieces must split evenly over the waves");
    constexpr int XPW = BM / RPP / NWV;
    const int pl_x = (int)(unsigned)(a.x_lo * 2), pl_w = (int)(unsigned)(a.w_lo * 2);   // FMT_SPLIT: lo-plane byte offsets
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int p = wave + NWV * i;
        const bool isx_i = i < XPW;
        const int r = (isx_i ? p : p - BM / RPP) * RPP + srow;   // tile-local row
        // source chunk landing at LDS position spos (bank swizzle through the source address)
        const int c = spos ^ (BK == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3));
        if (isx_i) { int xm = m0 + r; xm = xm < a.M ? xm : a.M - 1; voff[i] = (int)(((long)(xm - m0) * a.ldx + c * 8) * 2); }
        else { int wr = n0 + r; wr = wr < a.N ? wr : a.N - 1; voff[i] = ((wr - n0) * a.K + c * 8) * 2; }
        lds_off[i] = (isx_i ? 0 : XT) + (isx_i ? p : p - BM / RPP) * 1024;
    }
    // FMT_SPLIT: the K loop runs three segments over the same K range -- X.hi W.hi, X.lo W.hi, X.hi W.lo -- into one
    // accumulator; a k-tile index beyond K / BK selects the plane through the scalar offset only
    const int ntk = a.K / BK;
    auto ksoff = [&](int kt, bool isx_i) -> int {
        if constexpr (FMT != FMT_SPLIT) return (a.kpat && isx_i) ? tap3_offset(kt * (BK * 2)) : kt * (BK * 2);
        else {
            const int seg = (kt >= ntk ? 1 : 0) + (kt >= 2 * ntk ? 1 : 0);
            return (kt - seg * ntk) * (BK * 2) + (seg == (isx_i ? 1 : 2) ? (isx_i ? pl_x : pl_w) : 0);
        }
    };
    auto dma = [&](int kt, int i, char* base) {
        const bool isx_i = i < XPW;
        glds16b(isx_i ? rx : rw, voff[i], ksoff(kt, isx_i), base + lds_off[i]);
    };
    auto stage = [&](int kt, int slot) {
        char* base = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < NPW; ++i) dma(kt, i, base);
    };

    // ---- fragment read addresses (bytes within a stage)
    const int frow = lane & 31;
    const int swz = BK == 64 ? ((lane >> 1) & 7) : ((lane >> 2) & 3);   // swizzle key of row 32*j + (lane & 31)
    const int fhalf = lane >> 5;
    int koff[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) koff[kk] = (((2 * kk + fhalf) ^ swz) << 4);
    const int xrow_off = (wm * 32 * FM + frow) * RB;
    const int wrow_off = XT + (wn * 32 * FN + frow) * RB;

    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bf16x8_t xf[2][FM], wf[2][FN];
    auto read_frags = [&](const char* sb, int kk, int buf) {
#pragma unroll
        for (int f = 0; f < FM; ++f) xf[buf][f] = *(const bf16x8_t*)(sb + xrow_off + f * 32 * RB + koff[kk]);
#pragma unroll
        for (int f = 0; f < FN; ++f) wf[buf][f] = *(const bf16x8_t*)(sb + wrow_off + f * 32 * RB + koff[kk]);
    };
    auto mfmas = [&](int buf) {
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) {
                acc[fm][fn] = H16<FMT>::mfma(wf[buf][fn], xf[buf][fm], acc[fm][fn]);
            }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    };
    // MFMAs of one k-substep with LDS-DMA pieces [p0, p1) of k-tile `kt` (into ring slot `dslot`) issued
    // between them.  A DMA instruction costs ~70-100 issue cycles (the CU's texture path moves 64 B/clk and
    // is shared by every resident wave); issued in one burst after the barrier they were 43 % of the K step.
    auto mfmas_dma = [&](int buf, int kt, int dslot, int p0, int p1, bool on) {
        constexpr int NM = FM * FN;
        char* base = smem + dslot * STAGE;
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const int fm = i / FN, fn = i % FN;
            acc[fm][fn] = H16<FMT>::mfma(wf[buf][fn], xf[buf][fm], acc[fm][fn]);
            // pieces are spread evenly: piece p0 + j goes after MFMA number ceil((j+1)*NM/np) - 1
#pragma unroll
            for (int p = 0; p < NPW; ++p) {
                const int np = p1 - p0;
                if (p >= p0 && p < p1 && ((p - p0 + 1) * NM + np - 1) / np - 1 == i) {
                    SCHED_FENCE();
                    if (on) dma(kt, p, base);
                    SCHED_FENCE();
                }
            }
        }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    };

    // ---- NSTAGE-slot ring: tiles t+1 .. t+NSTAGE-1 in flight while tile t is consumed
    const int nt = FMT == FMT_SPLIT ? 3 * ntk : ntk;
    stage(0, 0);
    if (nt > 1) stage(1, 1);
    if constexpr (NSTAGE > 2) { if (nt > 2) stage(2, 2); }
    if constexpr (NSTAGE > 3) { if (nt > 3) stage(3, 3); }
    if (NSTAGE > 3 && nt > 3) wait_vmcnt<3 * NPW>();
    else if (NSTAGE > 2 && nt > 2) wait_vmcnt<2 * NPW>();
    else if (nt > 1) wait_vmcnt<NPW>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    read_frags(smem, 0, 0);
    int slot = 0;
    // one K tile.  STEADY (2-slot ring, 1 <= t <= nt - 3): every "does tile t+1 / t+2 exist" test is true, so the loop
    // that runs nearly all tiles carries no branch around its DMA instructions (same-box A/B of this peel on the 8-wave
    // kernel: conv1-4, FFN1 +3 %); the first and the last two tiles take the general form
    auto ktile = [&](auto steady, int t) {
        constexpr bool STEADY = decltype(steady)::value;
        const char* sb = smem + slot * STAGE;
        const int nslot = slot == NSTAGE - 1 ? 0 : slot + 1;
        // 2-slot ring: the DMA of tile t+1 (into the slot tile t-1 left at the previous barrier) was started
        // under the last MFMA group of tile t-1 (pieces [0, D0)) and continues under groups 0 and 1 here
        constexpr int D0 = (NPW * 2 + 4) / 5, D1 = D0 + (NPW - D0 + 1) / 2;   // e.g. NPW = 10 -> 4 | 3 | 3
        const bool has_next = STEADY || (t + 1 < nt);
        const bool cont = STEADY || ((NSTAGE == 2) && (t >= 1) && (t + 1 < nt));
        const bool dma2 = STEADY || ((t + 1 < nt) && (t + 2 < nt));
#pragma unroll
        for (int kk = 0; kk < KK - 1; ++kk) {
            SCHED_FENCE();
            read_frags(sb, kk + 1, (kk + 1) & 1);
            SCHED_FENCE();
            if (NSTAGE == 2 && kk == 0) mfmas_dma(0, t + 1, nslot, D0, D1, cont);
            else if (NSTAGE == 2 && kk == 1) mfmas_dma(1, t + 1, nslot, D1, NPW, cont);
            else mfmas(kk & 1);
        }
        SCHED_FENCE();
        if (has_next) {
            // every ds_read of tile t has been issued; once they have landed (lgkmcnt(0)) and my pieces of
            // tile t+1 have landed (counted vmcnt: tile t+2 stays in flight), meet the other waves.  After
            // the barrier tile t+1 is visible and slot(t) is dead -> refill it with tile t+3, and fetch the
            // first fragments of tile t+1 so that they land under the last 8 MFMAs of tile t.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // outstanding tiles: t+1 .. min(t+NSTAGE-1, nt-1); everything after t+1 may stay in flight
            if (NSTAGE > 3 && t + 3 < nt) wait_vmcnt<2 * NPW>();
            else if (NSTAGE > 2 && t + 2 < nt) wait_vmcnt<NPW>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            SCHED_FENCE();
            read_frags(smem + nslot * STAGE, 0, 0);
            if constexpr (NSTAGE != 2) { if (t + NSTAGE < nt) stage(t + NSTAGE, slot); }
            SCHED_FENCE();
        }
        if constexpr (NSTAGE == 2) mfmas_dma((KK - 1) & 1, t + 2, slot, 0, D0, dma2);
        else mfmas((KK - 1) & 1);
        slot = nslot;
    };
    if constexpr (NSTAGE == 2) {
        int t = 0;
        if (nt > 0) ktile(std::false_type{}, t++);
        for (; t < nt - 2; ++t) ktile(std::true_type{}, t);
        for (; t < nt; ++t) ktile(std::false_type{}, t);
    } else {
        for (int t = 0; t < nt; ++t) ktile(std::false_type{}, t);
    }

    // ---- epilogue
    if constexpr (EPI == EPI_QK || EPI == EPI_PROJ) {
        // measured A/B (MI355X): the LDS-staged, line-coalesced epilogue wins for the scattered head-major /
        // dual-output epilogues (+7 % qk, +40 % proj) and is neutral-to-slightly-negative for the others
        static_assert(NWV * StagedEpi<FN, EPI>::BYTES <= NSTAGE * STAGE, "epilogue staging must fit the ring");
        __builtin_amdgcn_s_barrier();                 // every wave is done reading operand tiles
        char* my = smem + wave * StagedEpi<FN, EPI>::BYTES;
        epilogue_staged<FM, FN, EPI, ACT, FMT>(a, acc, m0 + wm * 32 * FM, n0 + wn * 32 * FN, my, lane);
    } else {
        epilogue_direct<FM, FN, EPI, ACT, FMT>(a, acc, m0 + wm * 32 * FM, n0 + wn * 32 * FN, lane);
    }
}

// One launch = min(#tiles, workgroups_per_cu x 256) persistent workgroups walking the tile list with stride
// gridDim.x.  With `per_cu = 1` a launch holds only ONE of the two LDS slots of a CU, so the GEMM of the other
// in-flight batch (a different kernel, in a different phase) can take the second one: its K loop then runs
// under this launch's HBM-bound epilogue and vice versa, instead of two workgroups of the SAME launch hitting
// their epilogues together.
template <int FM, int FN, int BK, int NSTAGE, int MINB, bool PRIO, int EPI, int ACT, int FMT, int WM = 2, int WN = 2>
__global__ __launch_bounds__(64 * WM * WN, MINB) void gemm_bf16_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    constexpr int BM = 32 * FM * WM, BN = 32 * FN * WN;
    const int ntiles = ((a.M - a.m_begin + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        gemm_bf16_tile<FM, FN, BK, NSTAGE, PRIO, EPI, ACT, FMT, WM, WN>(a, tile, smem);
        if (tile + (int)gridDim.x < ntiles) __builtin_amdgcn_s_barrier();   // LDS (operand ring / epilogue staging) is reused
    }
}

template <int FM, int FN, int BK, int NSTAGE, int MINB, bool PRIO, int EPI, int ACT, int FMT, int WM = 2, int WN = 2>
static int launch_cfg(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 32 * FM * WM, BN = 32 * FN * WN;
    constexpr int LDS = NSTAGE * (BM + BN) * BK * 2;
    if (a.K % BK != 0) { syl_set_error("launch_gemm_bf16", "K must be a multiple of the K step"); return 1; }
    const int tiles = ((a.M - a.m_begin + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    static PerDeviceOnce attr_once;
    auto kern = gemm_bf16_kernel<FM, FN, BK, NSTAGE, MINB, PRIO, EPI, ACT, FMT, WM, WN>;
    if (attr_once.need()) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    // multi-round launches run as 2 x 256 persistent workgroups (q/k/v: 1536 tiles; same-box A/B -1.4 %); tune_persist
    // overrides the workgroups per CU, < 0 = one workgroup per tile
    int grid = tiles;
    const int per_cu = a.tune_persist > 0 ? a.tune_persist : (a.tune_persist == 0 ? MINB : 0);
    if (per_cu > 0 && tiles > per_cu * 256) grid = per_cu * 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// 8-wave variant (512 threads, two waves per SIMD, ONE workgroup per CU) for the big tiles 256x256 and
// 256x192.  Per-CU L1/TA traffic per FLOP falls with the tile area, which is what caps the 4-wave kernel
// above (128x192 needs ~53 B/clk/CU of the 64 B/clk texture path at full MFMA rate).  The two waves
// of a SIMD belong to two groups staggered by ONE barrier: each step is
//     A: ds_read this step's fragments + issue a share of the next stage's LDS-DMA              | barrier
//     B: 2*FM*FN MFMAs from registers                                                       | barrier
// so while group 0 is in B (matrix pipe), group 1 is in A (LDS / VMEM issue) and vice versa.
// Hazards: a step's DMA is retired (vmcnt) two program barriers before its first ds_read (one more than
// usual because the groups are staggered); fragment reads complete (lgkmcnt(0)) before the barrier that
// lets the other group refill that slot.
// VAR 0 = the shipping schedule (the LDS-DMA of step s+3 interleaved with the MFMAs of phase B: issuing it at the head or the tail
// of phase A, or half / half, measured 4-10 % slower: profiles/r03_gemm_variants_ab.md); VAR 10 = the same with s_memtime stamps
template <int FM, int FN, int WM, int WN, int EPI, int ACT, int FMT, int VAR = 0>
__device__ __forceinline__ void gemm8_bf16_tile(const GemmArgs& a, const int tile_id, char* smem) {
    // K step 32 (64-byte LDS rows), 4-slot ring: the DMA of step s+3 is issued in step s and retired with
    // counted vmcnt, never 0 in the steady state.  (A 2-slot ring of 64-wide stages measured 5-12 % slower.)
    constexpr int BM = 32 * FM * WM, BN = 32 * FN * WN;
    constexpr int RB = 64;
    constexpr int XT = BM * RB, WT = BN * RB, STAGE = XT + WT;
    constexpr int NP = (BM + BN) / 16;               // 1-KiB pieces (16 rows x 64 B) per step
    constexpr int NPW_HI = (NP + 7) / 8, NPW_LO = NP / 8;
    static_assert(WM * WN == 8, "8 waves");
    unsigned long long tk_start = 0;
    if constexpr (VAR == 10) TSTAMP(tk_start);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int group = wave >> 2;                     // waves w and w+4 share a SIMD
    const int wm = wave / WN, wn = wave % WN;
    const bool hi = wave < (NP % 8 == 0 ? 8 : NP % 8);   // this wave issues NPW_HI pieces per step

    const int tiles_n = (a.N + BN - 1) / BN;
    const int tiles_m = (a.M - a.m_begin + BM - 1) / BM;
    const int wg = xcd_remap(tile_id, tiles_m * tiles_n);
    const int m0 = a.m_begin + (wg / tiles_n) * BM;
    const int n0 = (wg % tiles_n) * BN;

    const int srow = lane >> 2, spos = lane & 3;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.X + (size_t)m0 * a.ldx), rw = make_rsrc(a.W + (size_t)n0 * a.K);
    int voff[NPW_HI];
    int lds_off[NPW_HI];
    // pieces wave + 8 i with i < XPW are X rows for every wave: the operand (and its descriptor) of piece i is static
    static_assert((BM / 16) % 8 == 0, "X pieces must split evenly over the 8 waves");
    constexpr int XPW = BM / 16 / 8;
    const int pl_x = (int)(unsigned)(a.x_lo * 2), pl_w = (int)(unsigned)(a.w_lo * 2);   // FMT_SPLIT: lo-plane byte offsets
#pragma unroll
    for (int i = 0; i < NPW_HI; ++i) {
        int p = wave + 8 * i;
        p = p < NP ? p : NP - 1;                     // (unused slot of a "lo" wave; never issued)
        const bool isx_i = i < XPW;
        const int r = (isx_i ? p : p - BM / 16) * 16 + srow;
        const int c = spos ^ ((r >> 2) & 3);
        if (isx_i) { int xm = m0 + r; xm = xm < a.M ? xm : a.M - 1; voff[i] = (int)(((long)(xm - m0) * a.ldx + c * 8) * 2); }
        else { int wr = n0 + r; wr = wr < a.N ? wr : a.N - 1; voff[i] = ((wr - n0) * a.K + c * 8) * 2; }
        lds_off[i] = (isx_i ? 0 : XT) + (isx_i ? p : p - BM / 16) * 1024;
    }
    const int ntk = a.K / 32;                        // FMT_SPLIT: three K segments, see gemm_bf16_tile
    auto ksoff = [&](int ks, bool isx_i) -> int {
        if constexpr (FMT != FMT_SPLIT) return (a.kpat && isx_i) ? tap3_offset(ks * 64) : ks * 64;
        else {
            const int seg = (ks >= ntk ? 1 : 0) + (ks >= 2 * ntk ? 1 : 0);
            return (ks - seg * ntk) * 64 + (seg == (isx_i ? 1 : 2) ? (isx_i ? pl_x : pl_w) : 0);
        }
    };
    auto dma1 = [&](int ks, int i, char* base) {
        const bool isx_i = i < XPW;
        glds16b(isx_i ? rx : rw, voff[i], ksoff(ks, isx_i), base + lds_off[i]);
    };
    auto stage = [&](int ks, int slot) {
        char* base = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < NPW_HI; ++i)
            if (i < NPW_LO || hi) dma1(ks, i, base);
    };
    auto wait_steps = [&](int nsteps_in_flight) {    // leave that many of MY steps' pieces outstanding
        if (hi) {
            if (nsteps_in_flight >= 2) wait_vmcnt<2 * NPW_HI>(); else if (nsteps_in_flight == 1) wait_vmcnt<NPW_HI>(); else wait_vmcnt<0>();
        } else {
            if (nsteps_in_flight >= 2) wait_vmcnt<2 * NPW_LO>(); else if (nsteps_in_flight == 1) wait_vmcnt<NPW_LO>(); else wait_vmcnt<0>();
        }
    };

    const int frow = lane & 31;
    const int swz = (lane >> 2) & 3;
    const int fhalf = lane >> 5;
    const int koff0 = (((0 + fhalf) ^ swz) << 4), koff1 = (((2 + fhalf) ^ swz) << 4);
    const int xrow_off = (wm * 32 * FM + frow) * RB;
    const int wrow_off = XT + (wn * 32 * FN + frow) * RB;

    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = FMT == FMT_SPLIT ? 3 * ntk : ntk;
    stage(0, 0);
    if (nt > 1) stage(1, 1);
    if (nt > 2) stage(2, 2);
    wait_steps(nt > 2 ? 2 : nt - 1);                 // step 0 landed (mine)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    if (group == 1) __builtin_amdgcn_s_barrier();    // stagger: group 1 runs one barrier behind
    int slot = 0;
    constexpr bool TRACE = VAR == 10;
    unsigned long long tk0 = tk_start, tl0 = 0, tl1 = 0, ta = 0, tb = 0, tc = 0, td = 0, te = 0, sA = 0, sW1 = 0, sB = 0, sW2 = 0, cal = 0;
    if constexpr (TRACE) { TSTAMP(tl0); TSTAMP(ta); TSTAMP(tb); cal = tb - ta; }
    for (int s = 0; s < nt; ++s) {
        const char* sb = smem + slot * STAGE;
        bf16x8_t xf[2][FM], wf[2][FN];
        if constexpr (TRACE) TSTAMP(ta);
        // ---- A: fragments of step s, DMA of step s+3, retire step s+1
        const bool dma = s + 3 < nt;
        char* dbase = smem + ((slot + 3) & 3) * STAGE;     // slot of step s-1: every wave left A(s-1) >= two program barriers ago
        SCHED_FENCE();
#pragma unroll
        for (int f = 0; f < FM; ++f) {
            xf[0][f] = *(const bf16x8_t*)(sb + xrow_off + f * 32 * RB + koff0);
            xf[1][f] = *(const bf16x8_t*)(sb + xrow_off + f * 32 * RB + koff1);
        }
#pragma unroll
        for (int f = 0; f < FN; ++f) {
            wf[0][f] = *(const bf16x8_t*)(sb + wrow_off + f * 32 * RB + koff0);
            wf[1][f] = *(const bf16x8_t*)(sb + wrow_off + f * 32 * RB + koff1);
        }
        {
            // retire step s+1: step s+3's DMA is issued in phase B below, so the steps issued after s+1 are at most {s+2}
            wait_steps(s + 2 < nt ? 1 : 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SCHED_FENCE();
        if constexpr (TRACE) { TSTAMP(tb); SCHED_FENCE(); }
        __builtin_amdgcn_s_barrier();
        SCHED_FENCE();
        if constexpr (TRACE) { TSTAMP(tc); SCHED_FENCE(); }
        // ---- B: MFMAs of step s
        // VAR 0: the LDS-DMA of step s+3 is spread between the MFMAs (one 1-KiB piece per quarter of the MFMAs): a DMA
        // instruction costs ~70-100 issue cycles because the CU's texture path moves 64 B/clk, and issued in a burst in
        // phase A it made A the longer phase (measured 623 vs 517 cycles).
        constexpr int NMF = 2 * FM * FN;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
            const int kk = i / (FM * FN), fm = (i / FN) % FM, fn = i % FN;
            acc[fm][fn] = H16<FMT>::mfma(wf[kk][fn], xf[kk][fm], acc[fm][fn]);
            // after MFMA number (q+1)*NMF/NPW_HI - 1 issue piece q
            if ((i + 1) % (NMF / NPW_HI) == 0) {
                const int q = (i + 1) / (NMF / NPW_HI) - 1;
                SCHED_FENCE();
                if (dma && q < NPW_HI && (q < NPW_LO || hi)) dma1(s + 3, q, dbase);
                SCHED_FENCE();
            }
        }
        __builtin_amdgcn_s_setprio(0);
        SCHED_FENCE();
        if constexpr (TRACE) { TSTAMP(td); SCHED_FENCE(); }
        __builtin_amdgcn_s_barrier();
        if constexpr (TRACE) { SCHED_FENCE(); TSTAMP(te); sA += tb - ta; sW1 += tc - tb; sB += td - tc; sW2 += te - td; }
        slot = (slot + 1) & 3;
    }
    if constexpr (TRACE) TSTAMP(tl1);
    if (group == 0) __builtin_amdgcn_s_barrier();    // pairs with group 1's extra barrier
    auto trace_out = [&](unsigned long long t_end) {
        if (a.trace && blockIdx.x == 0 && (wave & 3) == 0 && lane == 0) {
            unsigned long long* o = a.trace + group * 10;
            o[0] = sA; o[1] = sW1; o[2] = sB; o[3] = sW2; o[4] = tl1 - tl0; o[5] = (unsigned long long)nt; o[6] = cal;
            o[7] = t_end - tl1; o[8] = tl0 - tk0; o[9] = t_end - tk0;
        }
    };

    if constexpr (EPI == EPI_QK || EPI == EPI_PROJ || EPI == EPI_BF16) {
        // measured A/B (MI355X, instrumented): in this one-workgroup-per-CU kernel the LDS-staged, line-coalesced
        // epilogue cuts the bf16 epilogue from 29.8 k to 19.4 k cycles per 256x256 tile (FFN1 +7-9 %, convs neutral)
        static_assert(8 * StagedEpi<FN, EPI>::BYTES <= 4 * STAGE, "epilogue staging must fit the ring");
        __builtin_amdgcn_s_barrier();                 // every wave is done reading operand tiles
        char* my = smem + wave * StagedEpi<FN, EPI>::BYTES;
        epilogue_staged<FM, FN, EPI, ACT, FMT>(a, acc, m0 + wm * 32 * FM, n0 + wn * 32 * FN, my, lane);
    } else {
        epilogue_direct<FM, FN, EPI, ACT, FMT>(a, acc, m0 + wm * 32 * FM, n0 + wn * 32 * FN, lane);
    }
    if constexpr (TRACE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the epilogue's stores have left
        unsigned long long t_end; TSTAMP(t_end);
        trace_out(t_end);
    }
}


// ---------------------------------------------------------------------------------------------------
// UNSTAGGERED 8-wave 256x256 tile (tile id 40).  The kernel above ping-pongs the two waves of a SIMD between an MFMA
// phase and a load phase with TWO workgroup barriers per K step; its own stamps (profiles/r03_gemm_kloop_trace.md) put a
// step at ~1420 cycles against the 1024 of the 2 x 16 MFMAs: the MFMA phase carries ~100 cycles of LDS-DMA issue, every
// barrier hand-over costs 50-75, and the load phase waits ~250 idle.  Here all eight waves run the SAME stream, software-
// pipelined inside the wave like the 4-wave kernel: fragments of the next k-substep are read while the MFMAs of the
// current one issue (two register sets), the LDS-DMA of step s+3 sits between those MFMAs, and there is ONE barrier per
// K step.  The two waves of a SIMD are not ordered against each other between barriers: whichever has an MFMA ready
// issues it, and one wave's DMA / LDS issue time is covered by the other's MFMAs.
//   step s (ring slot s & 3):   MFMA(s, kk0) x8  +  ds_read frags(s, kk1)      +  DMA(s+3) pieces 0,1  -> slot (s-1) & 3
//                               lgkmcnt(0), vmcnt: my pieces of step s+1 landed;  s_barrier  [B(s)]
//                               MFMA(s, kk1) x8  +  ds_read frags(s+1, kk0)    +  DMA(s+3) pieces 2,3
// Hazards: slot (s-1) is free after B(s-1) (its last reads were waited for before that barrier); step s+1 is read only
// after B(s), before which every wave retired its own pieces of it (younger: step s+2 and the first half of s+3 = 6).
template <int EPI, int ACT, int FMT, bool TRACE = false>
__device__ __forceinline__ void gemm8u_bf16_tile(const GemmArgs& a, const int tile_id, char* smem) {
    constexpr int FM = 4, FN = 2, WM = 2, WN = 4;
    unsigned long long tk0 = 0, tl0 = 0, tl1 = 0, ta = 0, tb = 0, tc = 0, td = 0, te = 0, sA = 0, sW1 = 0, sB = 0, sW2 = 0, cal = 0;
    if constexpr (TRACE) TSTAMP(tk0);
    constexpr int BM = 256, BN = 256, RB = 64;
    constexpr int XT = BM * RB, STAGE = (BM + BN) * RB;
    constexpr int NPW = 4;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M - a.m_begin + BM - 1) / BM;
    const int wg = xcd_remap(tile_id, tiles_m * tiles_n);
    const int m0 = a.m_begin + (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;
    const int srow = lane >> 2, spos = lane & 3;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.X + (size_t)m0 * a.ldx), rw = make_rsrc(a.W + (size_t)n0 * a.K);
    int voff[NPW];
    int lds_off[NPW];
    constexpr int XPW = BM / 16 / 8;                 // pieces wave + 8 i with i < XPW are X rows for every wave
    const int pl_x = (int)(unsigned)(a.x_lo * 2), pl_w = (int)(unsigned)(a.w_lo * 2);
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int p = wave + 8 * i;
        const bool isx_i = i < XPW;
        const int r = (isx_i ? p : p - BM / 16) * 16 + srow;
        const int c = spos ^ ((r >> 2) & 3);
        if (isx_i) { int xm = m0 + r; xm = xm < a.M ? xm : a.M - 1; voff[i] = (int)(((long)(xm - m0) * a.ldx + c * 8) * 2); }
        else { int wr = n0 + r; wr = wr < a.N ? wr : a.N - 1; voff[i] = ((wr - n0) * a.K + c * 8) * 2; }
        lds_off[i] = (isx_i ? 0 : XT) + (isx_i ? p : p - BM / 16) * 1024;
    }
    const int ntk = a.K / 32;
    auto ksoff = [&](int ks, bool isx_i) -> int {
        if constexpr (FMT != FMT_SPLIT) return (a.kpat && isx_i) ? tap3_offset(ks * 64) : ks * 64;
        else {
            const int seg = (ks >= ntk ? 1 : 0) + (ks >= 2 * ntk ? 1 : 0);
            return (ks - seg * ntk) * 64 + (seg == (isx_i ? 1 : 2) ? (isx_i ? pl_x : pl_w) : 0);
        }
    };
    auto dma1 = [&](int ks, int i, char* base) {
        const bool isx_i = i < XPW;
        glds16b(isx_i ? rx : rw, voff[i], ksoff(ks, isx_i), base + lds_off[i]);
    };
    auto stage = [&](int ks, int slot) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) dma1(ks, i, smem + slot * STAGE);
    };
    const int frow = lane & 31, swz = (lane >> 2) & 3, fhalf = lane >> 5;
    const int koff0 = (((0 + fhalf) ^ swz) << 4), koff1 = (((2 + fhalf) ^ swz) << 4);
    const int xrow_off = (wm * 32 * FM + frow) * RB;
    const int wrow_off = XT + (wn * 32 * FN + frow) * RB;
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
    // 8 MFMAs of one k-substep with DMA pieces [p0, p0 + 2) of step `ks` between them
    auto mfmas_dma = [&](int buf, int ks, char* dbase, int p0, bool on) {
#pragma unroll
        for (int i = 0; i < FM * FN; ++i) {
            const int fm = i / FN, fn = i % FN;
            acc[fm][fn] = H16<FMT>::mfma(wf[buf][fn], xf[buf][fm], acc[fm][fn]);
            if (i == 1 || i == 5) {
                const int q = p0 + (i == 5 ? 1 : 0);
                SCHED_FENCE();
                if (on) dma1(ks, q, dbase);
                SCHED_FENCE();
            }
        }
    };
    const int nt = FMT == FMT_SPLIT ? 3 * ntk : ntk;
    stage(0, 0);
    if (nt > 1) stage(1, 1);
    if (nt > 2) stage(2, 2);
    if (nt > 2) wait_vmcnt<2 * NPW>(); else if (nt > 1) wait_vmcnt<NPW>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    read_frags(smem, koff0, 0);
    int slot = 0;
    if constexpr (TRACE) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); TSTAMP(tl0); TSTAMP(ta); TSTAMP(tb); cal = tb - ta; }
    for (int s = 0; s < nt; ++s) {
        const char* sb = smem + slot * STAGE;
        const int nslot = (slot + 1) & 3;
        char* dbase = smem + ((slot + 3) & 3) * STAGE;
        const bool dma = s + 3 < nt;
        if constexpr (TRACE) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); TSTAMP(ta); }
        SCHED_FENCE();
        read_frags(sb, koff1, 1);
        SCHED_FENCE();
        mfmas_dma(0, s + 3, dbase, 0, dma);
        SCHED_FENCE();
        if (s + 1 < nt) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (TRACE) { TSTAMP(tb); SCHED_FENCE(); }
            // retire my pieces of step s+1; younger: step s+2 (4) and pieces 0,1 of step s+3
            if (s + 3 < nt) wait_vmcnt<NPW + 2>(); else if (s + 2 < nt) wait_vmcnt<NPW>(); else wait_vmcnt<0>();
            if constexpr (TRACE) { TSTAMP(tc); SCHED_FENCE(); }
            __builtin_amdgcn_s_barrier();
            SCHED_FENCE();
            if constexpr (TRACE) { TSTAMP(td); SCHED_FENCE(); }
            read_frags(smem + nslot * STAGE, koff0, 0);
            SCHED_FENCE();
        }
        mfmas_dma(1, s + 3, dbase, 2, dma);
        if constexpr (TRACE) { if (s + 1 < nt) { sA += tb - ta; sW1 += tc - tb; sB += td - tc; } }
        slot = nslot;
    }
    if constexpr (TRACE) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); TSTAMP(tl1); }
    __builtin_amdgcn_s_barrier();                     // every wave is done reading operand tiles
    if constexpr (TRACE) {
        // [0] kk0 group (8 MFMA + 6 reads + 2 DMA) incl. the lgkmcnt wait, [1] vmcnt wait, [2] barrier, [3] unused
        if (a.trace && blockIdx.x == 0 && (wave & 3) == 0 && lane == 0) {
            unsigned long long* o = a.trace + (wave >> 2) * 10;
            o[0] = sA; o[1] = sW1; o[2] = sB; o[3] = sW2; o[4] = tl1 - tl0; o[5] = (unsigned long long)nt; o[6] = cal; o[7] = 0; o[8] = tl0 - tk0; o[9] = tl1 - tk0;
        }
    }
    if constexpr (EPI == EPI_QK || EPI == EPI_PROJ || EPI == EPI_BF16) {
        char* my = smem + wave * StagedEpi<FN, EPI>::BYTES;
        epilogue_staged<FM, FN, EPI, ACT, FMT>(a, acc, m0 + wm * 32 * FM, n0 + wn * 32 * FN, my, lane);
    } else {
        epilogue_direct<FM, FN, EPI, ACT, FMT>(a, acc, m0 + wm * 32 * FM, n0 + wn * 32 * FN, lane);
    }
}

template <int EPI, int ACT, int FMT, bool TRACE = false>
__global__ __launch_bounds__(512, 2) void gemm8u_bf16_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int ntiles = ((a.M - a.m_begin + 255) / 256) * ((a.N + 255) / 256);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        gemm8u_bf16_tile<EPI, ACT, FMT, TRACE>(a, tile, smem);
        if (tile + (int)gridDim.x < ntiles) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
}

template <int EPI, int ACT, int FMT, bool TRACE = false>
static int launch_cfg8u(const GemmArgs& a, hipStream_t s) {
    constexpr int LDS = 4 * (256 + 256) * 64;
    const int tiles = ((a.M - a.m_begin + 255) / 256) * ((a.N + 255) / 256);
    static PerDeviceOnce attr_once;
    auto kern = gemm8u_bf16_kernel<EPI, ACT, FMT, TRACE>;
    if (attr_once.need()) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    int grid = tiles;
    if (a.tune_persist > 0 && tiles > 256) grid = 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// One launch = min(#tiles, 256 x per_cu) workgroups walking the tile list with stride gridDim.x (256 % 8 == 0, so a
// workgroup's tiles all map to its own XCD's chunk).  Measured with rocprofv3 PMC (profiles/r01_mfma_util.md): the
// one-tile-per-workgroup launch keeps the matrix pipe busy only 41 % of its resident cycles although the K loop alone
// is at 81 % — every tile boundary costs a workgroup dispatch on a CU that holds nothing else.
template <int FM, int FN, int WM, int WN, int EPI, int ACT, int FMT, int VAR = 0>
__global__ __launch_bounds__(512, 2) void gemm8_bf16_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    constexpr int BM = 32 * FM * WM, BN = 32 * FN * WN;
    const int ntiles = ((a.M - a.m_begin + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        gemm8_bf16_tile<FM, FN, WM, WN, EPI, ACT, FMT, VAR>(a, tile, smem);
        if (tile + (int)gridDim.x < ntiles) {
            // the ring is reused: this tile's epilogue staging reads and its stores' source data are done with LDS
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Persistent form of the 256x256 8-wave kernel for the launches that dominate the forward (conv1-5, FFN1: EPI_BF16,
// whole tiles only).  256 workgroups walk the tile list; the K loop is the one above.  What changes is the seam
// between two tiles of a workgroup: the LDS-DMA of the NEXT tile's first two K steps is requested before the epilogue
// of the current one (into ring slots 2 and 3, which the epilogue's staging area -- the first 36 KB -- does not touch),
// so the operand latency of a tile's prologue (~5 k of the ~50-85 k cycles of a tile) and the workgroup dispatch
// disappear under the epilogue.  Every tile therefore runs its ring from slot 2.
// vmcnt bookkeeping at the seam: VMEM operations of a wave retire in issue order on gfx9 (one counter for loads and
// stores; the compiler's own waitcnt insertion relies on the same rule), the epilogue issues exactly NST = 16 stores per
// wave (whole tiles: every store of the staged epilogue executes) and, with a bias, 8 loads that are consumed before
// the stores.  Issue order per wave: step0' step1' [bias loads] stores x16 step2' | K loop: step3' ...  So
// "step0' landed" = at most NPW + NST + NPW operations younger than it outstanding, "step1' landed" (step 0 of the
// new K loop) = NST + NPW; from step 1 on the stores are older than everything that may remain in flight.
template <int ACT, int FMT>
__global__ __launch_bounds__(512, 2) void gemm8p_bf16_kernel(const GemmArgs a) {
    constexpr int FM = 4, FN = 2, WM = 2, WN = 4, EPI = EPI_BF16;
    constexpr int BM = 256, BN = 256, RB = 64;
    constexpr int XT = BM * RB, WT = BN * RB, STAGE = XT + WT;
    constexpr int NPW = (BM + BN) / 16 / 8;          // 4 one-KiB pieces per wave and step
    constexpr int NST = FM * (StagedEpi<FN, EPI>::CH / 2) * (FMT == FMT_SPLIT ? 2 : 1);   // global stores per wave in the staged epilogue: 16 (two planes: 32)
    static_assert(8 * StagedEpi<FN, EPI>::BYTES <= 2 * STAGE, "staging must stay inside ring slots 0 and 1");
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int group = wave >> 2;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = a.N / BN, tiles_m = (a.M - a.m_begin) / BM;
    const int ntiles = tiles_m * tiles_n;
    const int srow = lane >> 2, spos = lane & 3;
    const int frow = lane & 31;
    const int swz = (lane >> 2) & 3;
    const int fhalf = lane >> 5;
    const int koff0 = (((0 + fhalf) ^ swz) << 4), koff1 = (((2 + fhalf) ^ swz) << 4);
    const int xrow_off = (wm * 32 * FM + frow) * RB;
    const int wrow_off = XT + (wn * 32 * FN + frow) * RB;
    const int ntk = a.K / 32;
    const int nt = FMT == FMT_SPLIT ? 3 * ntk : ntk;

    int lds_off[NPW];
    int voff[NPW];                                   // this lane's byte offset inside its operand's tile rows: the SAME for every tile
    constexpr int XPW = BM / 16 / 8;                 // pieces wave + 8 i with i < XPW (= 2) are X rows for every wave
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int p = wave + 8 * i;
        const bool isx_i = i < XPW;
        const int r = (isx_i ? p : p - BM / 16) * 16 + srow;
        const int c = spos ^ ((r >> 2) & 3);         // source chunk (bank swizzle through the source address)
        voff[i] = isx_i ? (int)(((long)r * a.ldx + c * 8) * 2) : (r * a.K + c * 8) * 2;
        lds_off[i] = (isx_i ? 0 : XT) + (isx_i ? p : p - BM / 16) * 1024;
    }
    const int pl_x = (int)(unsigned)(a.x_lo * 2), pl_w = (int)(unsigned)(a.w_lo * 2);
    // a tile = two buffer descriptors (X rows from m0, W rows from n0): scalar registers only
    struct TileSrc { __amdgpu_buffer_rsrc_t rx, rw; };
    auto setup = [&](int tile_id, TileSrc& g, int& m0, int& n0) {
        const int wg = xcd_remap(tile_id, ntiles);
        m0 = a.m_begin + (wg / tiles_n) * BM;
        n0 = (wg % tiles_n) * BN;
        g.rx = make_rsrc(a.X + (size_t)m0 * a.ldx);
        g.rw = make_rsrc(a.W + (size_t)n0 * a.K);
    };
    auto ksoff = [&](int ks, bool isx_i) -> int {
        if constexpr (FMT != FMT_SPLIT) return (a.kpat && isx_i) ? tap3_offset(ks * 64) : ks * 64;
        else {
            const int seg = (ks >= ntk ? 1 : 0) + (ks >= 2 * ntk ? 1 : 0);
            return (ks - seg * ntk) * 64 + (seg == (isx_i ? 1 : 2) ? (isx_i ? pl_x : pl_w) : 0);
        }
    };
    auto dma1 = [&](const TileSrc& g, int ks, int i, char* base) {
        const bool isx_i = i < XPW;
        glds16b(isx_i ? g.rx : g.rw, voff[i], ksoff(ks, isx_i), base + lds_off[i]);
    };
    auto stage = [&](const TileSrc& g, int ks, int slot) {
        char* base = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < NPW; ++i) dma1(g, ks, i, base);
    };

    TileSrc gp, gn;
    int m0, n0, m0n = 0, n0n = 0;
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    setup(tile, gp, m0, n0);
    stage(gp, 0, 2);
    if (nt > 1) stage(gp, 1, 3);
    if (nt > 2) stage(gp, 2, 0);
    if (nt > 2) wait_vmcnt<2 * NPW>(); else if (nt > 1) wait_vmcnt<NPW>(); else wait_vmcnt<0>();
    bool seam = false;                               // this tile's first K step follows an epilogue (stores in flight)
    for (;;) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        if (group == 1) __builtin_amdgcn_s_barrier();    // stagger: group 1 runs one barrier behind
        f32x16_t acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        int slot = 2;
        // one K step; DMA_ON is a compile-time property of the loop it sits in (main loop: step s+3 exists, tail: it does
        // not), so the steady-state loop carries no branch around its four DMA instructions
        auto kstep = [&](auto dma_on, int s) {
            constexpr bool DMA_ON = decltype(dma_on)::value;
            const char* sb = smem + slot * STAGE;
            bf16x8_t xf[2][FM], wf[2][FN];
            SCHED_FENCE();
#pragma unroll
            for (int f = 0; f < FM; ++f) {
                xf[0][f] = *(const bf16x8_t*)(sb + xrow_off + f * 32 * RB + koff0);
                xf[1][f] = *(const bf16x8_t*)(sb + xrow_off + f * 32 * RB + koff1);
            }
#pragma unroll
            for (int f = 0; f < FN; ++f) {
                wf[0][f] = *(const bf16x8_t*)(sb + wrow_off + f * 32 * RB + koff0);
                wf[1][f] = *(const bf16x8_t*)(sb + wrow_off + f * 32 * RB + koff1);
            }
            // retire step s+1 (step s+2 may stay in flight; at a seam the epilogue's stores sit between them)
            if (DMA_ON || nt - 2 - s >= 1) { if (seam && s == 0) wait_vmcnt<NST + NPW>(); else wait_vmcnt<NPW>(); }
            else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SCHED_FENCE();
            __builtin_amdgcn_s_barrier();
            SCHED_FENCE();
            char* dbase = smem + ((slot + 3) & 3) * STAGE;
            constexpr int NMF = 2 * FM * FN;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < NMF; ++i) {
                const int kk = i / (FM * FN), fm = (i / FN) % FM, fn = i % FN;
                acc[fm][fn] = H16<FMT>::mfma(wf[kk][fn], xf[kk][fm], acc[fm][fn]);
                if constexpr (DMA_ON) {
                    if ((i + 1) % (NMF / NPW) == 0) {
                        const int q = (i + 1) / (NMF / NPW) - 1;
                        SCHED_FENCE();
                        dma1(gp, s + 3, q, dbase);
                        SCHED_FENCE();
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            SCHED_FENCE();
            __builtin_amdgcn_s_barrier();
            slot = (slot + 1) & 3;
        };
        for (int s = 0; s < nt - 3; ++s) kstep(std::true_type{}, s);
        for (int s = nt - 3 > 0 ? nt - 3 : 0; s < nt; ++s) kstep(std::false_type{}, s);
        if (group == 0) __builtin_amdgcn_s_barrier();    // pairs with group 1's extra barrier
        __builtin_amdgcn_s_barrier();                    // every wave is done reading operand tiles
        const int next = tile + (int)gridDim.x;
        const bool more = next < ntiles;
        if (more) {
            setup(next, gn, m0n, n0n);
            stage(gn, 0, 2);
            if (nt > 1) stage(gn, 1, 3);
        }
        char* my = smem + wave * StagedEpi<FN, EPI>::BYTES;
        epilogue_staged<FM, FN, EPI, ACT, FMT>(a, acc, m0 + wm * 32 * FM, n0 + wn * 32 * FN, my, lane);
        if (!more) break;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // the staging area (ring slots 0, 1) is free again
        if (nt > 2) stage(gn, 2, 0);
        // step 0 of the next tile has landed: younger than it are step 1, the NST stores and step 2
        if (nt > 2) wait_vmcnt<NPW + NST + NPW>(); else if (nt > 1) wait_vmcnt<NPW + NST>(); else wait_vmcnt<NST>();
        gp = gn;
        m0 = m0n; n0 = n0n; tile = next; seam = true;
    }
}

template <int FM, int FN, int WM, int WN, int EPI, int ACT, int FMT, int VAR = 0>
static int launch_cfg8(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 32 * FM * WM, BN = 32 * FN * WN;
    constexpr int LDS = 4 * (BM + BN) * 64;
    const int tiles = ((a.M - a.m_begin + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    static PerDeviceOnce attr_once;
    auto kern = gemm8_bf16_kernel<FM, FN, WM, WN, EPI, ACT, FMT, VAR>;
    if (attr_once.need()) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    int grid = tiles;
    if (a.tune_persist > 0 && tiles > 256) grid = 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int ACT, int FMT>
static int launch_cfg8p(const GemmArgs& a, hipStream_t s) {
    constexpr int LDS = 4 * (256 + 256) * 64;
    static PerDeviceOnce attr_once;
    auto kern = gemm8p_bf16_kernel<ACT, FMT>;
    if (attr_once.need()) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// tail policy constants (measured: profiles/r06_tail_policy.md)
#ifndef GEMM_SOLO_EFF
#define GEMM_SOLO_EFF 0.60          // a 2-per-CU tile alone on its CU, relative to its rated efficiency (measured: a lone 128x192x3072 tile 66 us, two co-resident 80 us)
#endif
#ifndef GEMM_LONE_ROUND
#define GEMM_LONE_ROUND 0.90        // a launch of FEWER tiles than CUs on the one-per-CU persistent tiles: every tile runs alone on its CU for about a full round's time, however few
                                    // there are (B = 1: FFN1 as 36 tiles of 192x256 = 19.5-21 us against 10.3 us on 96 tiles of 128x128; profiles/r06_small_tiles.md) -- the
                                    // partial-round rule below (fitted to launches of one round and more) priced it at 0.48 and sent every GEMM of a single-clip call to the big tiles
#endif
#ifndef GEMM_SPLIT_US
#define GEMM_SPLIT_US 4.0           // what the second launch of a split costs (dispatch gap + a second prologue)
#endif
#ifndef GEMM_H192_EFF
#define GEMM_H192_EFF 0.90          // the 192-row tiles' efficiency relative to their 256-row siblings (in-forward: full rounds at 0.845 instead of 0.75 of the time)
#endif
#ifndef GEMM_PR0_ASM
#define GEMM_PR0_ASM 0.4            // cost of an (almost) empty partial round of the persistent hand-scheduled tiles, relative to a full round
#endif
#define GEMM_CU_MACS_PER_US 1.8e6   // one CU's rate on these loops (256 x 256 x 768 MACs in ~28 us)

// launch one tile configuration (by id) over rows [a.m_begin, a.M)
template <int EPI, int ACT, int FMT>
static int launch_tile(int cfg, const GemmArgs& a, hipStream_t s) {
    if constexpr (FMT == FMT_SPLIT) {
        // the split-operand mode instantiates the three tile shapes the cost model picks (and nothing else)
        if (cfg == 3) return launch_cfg<2, 2, 64, 2, 2, false, EPI, ACT, FMT>(a, s);
        if (cfg == 10) {
            if constexpr (EPI == EPI_BF16) {
                if (a.tune_persist >= 0 && (a.M - a.m_begin) % 256 == 0 && a.N % 256 == 0 && a.K >= 96 && (long)((a.M - a.m_begin) / 256) * (a.N / 256) > 256)
                    return launch_cfg8p<ACT, FMT>(a, s);
            }
            return launch_cfg8<4, 2, 2, 4, EPI, ACT, FMT>(a, s);
        }
        return launch_cfg<2, 3, 64, 2, 2, false, EPI, ACT, FMT>(a, s);
    } else
    switch (cfg) {
        case 0: return launch_cfg<4, 2, 64, 3, 1, false, EPI, ACT, FMT>(a, s);   // 256x128, 3-slot ring, 1 WG/CU
        // round 6: the launches of a call on one or two clips are chains of K steps on a few LONE workgroups (fewer tiles than CUs), each step waiting for LDS-DMA pieces
        // issued half a step earlier.  64-row tiles put four times the workgroups on the idle CUs, and a THREE-slot ring lets a step's pieces be one and a half steps old:
        // out-proj of one clip 12.5 -> 7.0 us, FFN2 25.5 -> 18.5, q,k,v 9.4 -> 7.9 (profiles/r06_small_tiles.md).  Same chain per output element: bit-identical.
        case 1: return launch_cfg<1, 1, 64, 3, 2, false, EPI, ACT, FMT>(a, s);   // 64x64, three-slot ring, 2+ WG/CU
        case 2: return launch_cfg<1, 2, 64, 2, 2, false, EPI, ACT, FMT>(a, s);   // 64x128, two-slot ring
        case 3: return launch_cfg<2, 2, 64, 2, 2, false, EPI, ACT, FMT>(a, s);   // 128x128, 2 WG/CU
        // (the projection's dual-output fp32 staging does not fit eight private regions into the two-slot ring: it keeps the four-wave forms)
        case 5:
            if constexpr (EPI == EPI_PROJ) return launch_cfg<2, 2, 64, 2, 2, false, EPI, ACT, FMT>(a, s);
            else return launch_cfg<1, 2, 64, 2, 2, false, EPI, ACT, FMT, 4, 2>(a, s);   // 128x128 on EIGHT waves (4 x 2, wave tile 32x64), 2 WG/CU
        case 6:
            if constexpr (EPI == EPI_PROJ) return launch_cfg<2, 3, 64, 2, 2, false, EPI, ACT, FMT>(a, s);
            else return launch_cfg<1, 3, 64, 2, 2, false, EPI, ACT, FMT, 4, 2>(a, s);   // 128x192 on eight waves (wave tile 32x96), 2 WG/CU
        case 10:
            if constexpr (EPI == EPI_BF16) {
                // whole tiles, more than one round, at least 3 K steps: the persistent kernel with cross-tile prefetch
                if (a.tune_persist >= 0 && (a.M - a.m_begin) % 256 == 0 && a.N % 256 == 0 && a.K >= 96 && (long)((a.M - a.m_begin) / 256) * (a.N / 256) > 256)
                    return launch_cfg8p<ACT, FMT>(a, s);
            }
            return launch_cfg8<4, 2, 2, 4, EPI, ACT, FMT>(a, s);                 // 256x256, 8 waves staggered
        case 11: return launch_cfg8<2, 3, 4, 2, EPI, ACT, FMT>(a, s);            // 256x192, 8 waves staggered
        case 13: case 14: case 15: case 16: case 17: case 46: case 47:
            // the v_mfma_f32_16x16x32 family (gemm_asm16.hip): the 16-bit-output launches by default (launch_f), any instantiated epilogue when forced
            if (gemm_asm16_has_tile(EPI, a, cfg)) { GemmArgs b = a; b.tune_cfg = cfg + 1; return launch_gemm_asm16(EPI, b, s); }
            break;
        case 51: case 57: case 60: case 61: case 62: case 63: case 64: case 65: case 66: case 67: case 68: case 69: case 70: case 71: case 72: case 73: case 74: case 75: case 76: case 77: case 78: case 80: case 81: case 82: case 83: case 85: case 86: case 87: case 88: case 89: case 90: case 91: case 92: case 93: case 94: case 95: case 96: case 97: case 98:
            // hand-scheduled K loop; a forced tile without an instantiation for this epilogue falls back to 128x192 (sylber_hip.h)
            if (gemm_asm_has_tile(EPI, a, cfg) || (cfg != 51 && cfg != 57 && cfg != 60 && cfg != 80 && cfg != 85 && cfg != 86 && cfg != 90 && cfg != 91 && cfg != 95 && cfg != 96 && cfg != 97 && gemm_asm_applicable(EPI, a))) {
                GemmArgs b = a; b.tune_cfg = cfg + 1; return launch_gemm_asm(EPI, b, s);
            }
            break;
        case 40: if constexpr (FMT == FMT_BF16) return launch_cfg8u<EPI, ACT, FMT>(a, s); break;                     // unstaggered 8-wave 256x256
        case 41: if constexpr (FMT == FMT_BF16 && EPI == EPI_BF16 && ACT == 0) return launch_cfg8u<EPI, ACT, FMT, true>(a, s); break;   // its trace
        case 30: if constexpr (FMT == FMT_BF16 && (EPI == EPI_BF16 || EPI == EPI_F32_RESLN)) return launch_cfg8<4, 2, 2, 4, EPI, ACT, FMT, 10>(a, s); break;   // trace
        default: break;
    }
    return launch_cfg<2, 3, 64, 2, 2, false, EPI, ACT, FMT>(a, s);              // 128x192, 2 WG/CU
}

// Tile-shape choice, measured on MI355X with tools/gemm_bench.py (random operands):
//   cfg 3  128x128, 4 waves, 2 workgroups/CU   best when the tile count just fills one round (conv6)
//   cfg 4  128x192, 4 waves, 2 workgroups/CU   best for N = 768 / 1536 (500 / 1000 tiles = 1 / 2 rounds)
//   cfg 10 256x256, 8 waves staggered, 1 WG/CU best for the big GEMMs (conv1-5, FFN1): +20-25 %
// Two co-resident workgroups (cfg 3/4) cover each other's epilogue / barrier / DMA-issue time; the big
// tile (cfg 10) instead halves the per-FLOP L1/TA traffic.  Cost = rounds x (tile area per CU) / eff.
//
// PARTIAL ROUNDS (round 6).  The persistent big tiles walk the tile list 256 at a time, so a launch of r + f rounds (0 < f < 1) costs
// r + 1: the 32 x 10 s headline is exactly 3 rounds (FFN1 / q,k,v) and 1 round (out-proj / FFN2), every other batch shape pays up to a
// whole round per launch (8 x 60 s: 4.4 -> 5 and 1.47 -> 2; profiles/r06_shape_sweep.md).  Two mechanisms, both bit-identical (every
// output element is one fp32 chain over K in the same order whatever tile computes it; tests/test_gpu_ops.py):
//   * a second tile HEIGHT: tiles 51 / 57 = the loops of 91 / 97 on 192-row tiles (gemm_asm.hip); the cost model below picks the height
//     whose tile count wastes less of its last round (8 x 60 s, N = 768: 376 tiles = 2 rounds at 256 rows, 504 tiles = 2 rounds of
//     3/4-size tiles at 192 rows)
//   * a row split: rows of the full rounds on the chosen tile, rows [M1, M) as a second launch on another tile (GemmArgs::m_begin).
//     Measured not to pay with the tiles that exist (see below): kept as a forced option and as the test vehicle of m_begin.
// The tile cost model (shared by launch_f and the CPU-tier regression test through sylber_debug_gemm_pick): configuration table and cost of
// configuration i over `rows` rows of launch `a`, for epilogue EPI and operand format FMT.
struct TileCfg { int id, bm, bn, per_cu; double eff, pr0; };
struct TileModel {
    static constexpr int NCFG = 11;
    TileCfg cfgs[NCFG];
    bool asm_ok, r5, m16, split;
    int EPI;
    const GemmArgs* a;
    TileModel(int EPI_, int FMT, const GemmArgs& a_) : EPI(EPI_), a(&a_) {
        const GemmArgs& a = a_;
        // 80 / 90: the hand-scheduled 4-wave kernels (gemm_asm.hip).  Their K loop runs ~25 % above the 8-wave kernel's, but one
        // wave per SIMD leaves a tile's prologue and epilogue uncovered (~9 us + ~5 us of GELU against a 17 us K = 768 loop), so
        // they are rated for long K only (same-box tools/gemm_bench.py: conv1-4 +3-10 %, FFN2 +7 %, K = 768 shapes -5-20 %).
        split = FMT == FMT_SPLIT;
        asm_ok = FMT != FMT_SPLIT && gemm_asm_applicable(EPI, a) && a.K >= 256;
        const bool long_k = a.K >= 1024;
        // 95: the same loop on eight waves (two per SIMD share the epilogue's VALU work): the GELU GEMMs, K = 768 included.
        // 85 / 91 / 97 = 80 / 90 / 95 with three ring slots for X (bf16 only): never slower hot, 5-15 % faster on cold activations
        const bool x3 = FMT != FMT_SPLIT;
        // 86 = the X3 loop on a 256x128 tile / four waves: slower per FLOP than 97 (923 vs 974 TF on conv1), but a launch of 64 row tiles
        // x 512 columns (conv6) fills 256 CUs with it and half of them with 256x256 tiles: 22.0 vs 29.7 us
        // 51 / 57 = tiles 91 / 97 at 192 rows (round 6), rated GEMM_H192_EFF of their siblings (fewer FLOPs per staged byte)
        //
        // Round 6 re-fitted the model to IN-FORWARD launch times (sequential profile, cold activations, seven batch shapes x every tile forced:
        // tools/tile_pick_sweep.py -> profiles/r06_tile_pick.md) instead of the hot micro-benchmark it came from:
        //   * a PARTIAL last round is not a whole round.  Tiles beyond the last full round fill a fraction f of the launch's slots; that round
        //     costs pr0 + (1 - pr0) f of a full one: pr0 = 0.4 for the persistent hand-scheduled tiles (a 0.41-full round of tile 91 costs 0.63, a
        //     0.47-full one 0.70) and for the two-per-CU kernels, 1 for the 8-wave hipcc kernel (its 3.3 rounds cost 4)
        //   * the hipcc 8-wave kernel (10) was rated 1.20 for every epilogue; with the fp32-residual epilogue it runs at 1.0 (FFN2 1.93 vs
        //     1.15 ms per forward at 8 x 60 s) and was being picked for q,k,v / out-proj wherever its tile count rounded well
        //   * q,k,v: the 128x192 two-per-CU kernel is rated 1.09 (0.614 vs 0.645 ms per forward on tile 91 at 32 x 10 s, 0.89 vs 1.00 at 8 x 60 s)
        // tune_model = 5 (SYLBER_OPT_GEMM_MODEL: "this handle shares the chip with another in-flight batch") = the round-5 constants, whole rounds, no
        // 192-row tiles: with two batches in flight the other stream's kernels take the CUs a partial round leaves idle, and that selection measured
        // fastest there on five batch shapes (profiles/r06_tile_model_ab.md); alone on the chip it is the slower one.
        r5 = a.tune_model == 5;
        const double e10 = r5 || EPI == EPI_BF16 ? 1.20 : (EPI == EPI_QK ? 1.15 : 1.00);
        const double e4 = !r5 && EPI == EPI_QK ? 1.09 : 1.00;
        // tune_model = 2: "throughput" -- the handle is one of several in flight (bench.py's pipeline, Segmenter.stream's neighbours): the CUs a
        // partial round leaves idle are taken by the other stream's kernels, so a partial round costs only its share (pr0 = 0 for every tile)
        const bool thr = a.tune_model == 2;
        // (two-per-CU kernels: q,k,v's 4.4 rounds cost 4.4 round-times, but FFN2's 0.75 of a round costs 0.90 of one and 1.47 rounds cost 1.6: the same
        //  pr0 as the persistent tiles fits all three within 5 %; charging them proportionally made the model take 128x192 for FFN2 where tile 51 is 10-15 % faster)
        const double pa = r5 ? 1.0 : (thr ? 0.0 : GEMM_PR0_ASM), p2 = r5 ? 1.0 : (thr ? 0.0 : GEMM_PR0_ASM), p10 = thr ? 0.0 : 1.0;
        const TileCfg init[NCFG] = {{3, 128, 128, 2, r5 ? 0.93 : 0.95, p2}, {4, 128, 192, 2, e4, p2}, {10, 256, 256, 1, e10, p10},
                                {x3 ? 85 : 80, 256, 256, 1, long_k ? (r5 ? 1.28 : 1.20) : (!r5 && EPI == EPI_QK ? 1.00 : 1.10), pa},   // (q,k,v on 85: 0.795 vs 0.691 ms on 128x192 at 24 x 15 s)
                                {91, 256, 192, 1, long_k ? 1.10 : 1.04, pa},
                                {x3 ? 97 : 95, 256, 256, 1, long_k ? 1.32 : 1.27, pa}, {86, 256, 128, 1, 1.10, pa},
                                {51, 192, 192, 1, (long_k ? 1.10 : 1.04) * GEMM_H192_EFF, pa}, {57, 192, 256, 1, (long_k ? 1.32 : 1.27) * GEMM_H192_EFF, pa},
                                // 64-row tiles (round 6): a quarter / half of the 128x128 tile per workgroup at 0.34 / 0.50 of its per-output efficiency -- they only ever win where a
                                // launch has fewer tiles than the chip has workgroup slots (one to four clips), which is what they exist for (fitted: profiles/r06_small_tiles.md §3)
                                {1, 64, 64, 2, 0.34, p2}, {2, 64, 128, 2, 0.50, p2}};
        for (int i = 0; i < NCFG; ++i) cfgs[i] = init[i];
        // The 16-bit-output GEMMs (EPI_BF16, plain or GELU: conv1-5 and FFN1, 46 % of the 32 x 10 s forward) run on the v_mfma_f32_16x16x32 FAMILY
        // (gemm_asm16.hip; same-box +2.7 ... +7.2 % per launch, profiles/r06_mfma16_loop.md): 13 / 14 = the two-per-CU 128x128 / 128x192 kernels, 47 =
        // tile 97's geometry, 46 = its 192-row sibling.  The family adds 32-k blocks to an element's fp32 chain where the 32x32x16 kernels add 16-k
        // blocks, so the ROLE moves as a whole: every tile a batch shape can pick for these launches is a member, and results stay independent of
        // the batch shape.  tune_mfma16 = -1 (SYLBER_OPT_GEMM_MFMA16) puts the role back on the 32x32x16 kernels.
        m16 = EPI == EPI_BF16 && FMT != FMT_SPLIT && (a.act == 0 || a.act == 1) && a.tune_mfma16 >= 0 && gemm_asm16_has_tile(EPI, a, 15);
        if (m16) {
            // small tiles: 15 = the 128x128 tile on EIGHT waves (+3 ... +21 % over its four-wave form 13 on every shape of the role: a 16-cycle-MFMA loop is bound by
            // what one wave has to issue, profiles/r06_small_tiles.md); 14 = 128x192 on four waves, rated below it (it wins only where 512 of its tiles are exactly
            // one round: 8 clips).  13 and 16 (128x192 on eight waves: 136 VGPRs, one workgroup per CU) stay forced-only members.
            // 17 = 64x64 on a three-slot ring: the launches of one or two clips (the family's form of tile 1 below)
            const TileCfg fam[5] = {{15, 128, 128, 2, 1.00, p2}, {14, 128, 192, 2, 0.88, p2}, {47, 256, 256, 1, long_k ? 1.32 : 1.27, pa},
                                    {46, 192, 256, 1, (long_k ? 1.32 : 1.27) * GEMM_H192_EFF, pa}, {17, 64, 64, 2, 0.36, p2}};
            for (int i = 0; i < NCFG; ++i) cfgs[i] = i < 5 ? fam[i] : TileCfg{-1, 256, 256, 1, 1.0, 1.0};
        }
    }
    // the member of the 16x16x32 family that stands in for a FORCED tile id (SYLBER_OPT_GEMM_TILE is per handle: the parity tests force one id on
    // every launch of a forward and expect the same bits from each; on the role's launches every id maps to a member of the same shape class)
    int family_member(int cfg) const {
        const GemmArgs& a = *this->a;
        int m;
        switch (cfg) {
            case 13: case 14: case 15: case 16: case 17: case 46: case 47: m = cfg; break;
            case 1: m = 17; break;
            case 2: case 3: case 5: m = 15; break;
            case 6: m = 16; break;
            case 51: case 57: m = 46; break;
            case 10: case 30: case 40: case 41: case 60: case 80: case 85: case 95: case 97: case 98: m = 47; break;
            case 4: case 11: case 90: case 91: case 96: case 0: case 86: m = 14; break;
            default: m = 15; break;
        }
        return gemm_asm16_has_tile(EPI, a, m) ? m : 15;
    }
    double cost_of(int i, long rows) const {
        const GemmArgs& a = *this->a;
        const TileCfg& c = cfgs[i];
        if (c.id < 0) return 1e300;
        if (m16) { if (!gemm_asm16_has_tile(EPI, a, c.id) || (c.id == 17 && a.tune_model == 6)) return 1e300; }
        else if (c.id == 1 || c.id == 2) { if (split || a.tune_model == 6) return 1e300; }    // (the split16 mode instantiates three tile shapes; model 6 = the round-6a choice, A/B)
        else if (i >= 3 && (!asm_ok || !gemm_asm_has_tile(EPI, a, c.id))) return 1e300;   // only tiles that exist for this epilogue / format
        if ((c.id == 51 || c.id == 57 || c.id == 46) && (a.tune_h192 < 0 || r5)) return 1e300;
        const long tm = (rows + c.bm - 1) / c.bm, tn = (a.N + c.bn - 1) / c.bn;
        if (c.id == 86 && tm * tn < 192) return 1e300;      // (measured for launches that fill the chip; small batches keep their tiles)
        const long slots = 256L * c.per_cu, tiles = tm * tn;
        const long full = tiles / slots, rem = tiles - full * slots;
        double rounds = (double)full;
        if (rem > 0) {
            // a two-per-CU kernel whose partial round leaves every CU at most ONE workgroup: that workgroup runs alone (nobody covers
            // its prologue / epilogue), GEMM_SOLO_EFF of the paired rate
            if (!r5 && c.per_cu == 2 && full == 0 && rem <= 256) rounds = 0.5 / GEMM_SOLO_EFF;
            else rounds += c.pr0 + (1.0 - c.pr0) * (double)rem / (double)slots;
            if (full == 0 && c.per_cu == 1 && a.tune_model != 6 && rounds < GEMM_LONE_ROUND) rounds = GEMM_LONE_ROUND;   // (tune_model 6 = model 0 without this rule: A/B)
        }
        return rounds * c.per_cu * c.bm * c.bn / c.eff;

    }
    int pick(long rows, double* cost) const {
        int best = 0;
        double best_cost = 1e300;
        for (int i = 0; i < NCFG; ++i) {
            const double ci = cost_of(i, rows);
            if (ci < best_cost) { best_cost = ci; best = i; }
        }
        *cost = best_cost;
        return best;
    }
};
int gemm_pick_tile(int epi, const GemmArgs& a) {
    TileModel m(epi, a.fmt, a);
    double c;
    return m.cfgs[m.pick(a.M - a.m_begin, &c)].id;
}

template <int EPI, int ACT, int FMT>
static int launch_f(const GemmArgs& a, hipStream_t s) {
    const TileModel tm_(EPI, FMT, a);
    const TileCfg* cfgs = tm_.cfgs;
    constexpr int NCFG = TileModel::NCFG;
    auto cost_of = [&](int i, long rows) -> double { return tm_.cost_of(i, rows); };
    auto pick = [&](long rows, double* cost) -> int { return tm_.pick(rows, cost); };
    const long rows = a.M - a.m_begin;
    double whole_cost;
    const int best = pick(rows, &whole_cost);
    int cfg = cfgs[best].id;
    if (a.tune_cfg > 0) cfg = tm_.m16 ? tm_.family_member(a.tune_cfg - 1) : a.tune_cfg - 1;   // per-call override (sylber_set_option / parity tests)
    // ---- row split: rows of the full rounds on the chosen tile, the rest re-tiled.  FORCED ONLY (tune_tail > 0): measured (profiles/r06_tail_policy.md),
    // a smaller tile alone on its CU runs its K loop at the same wall time per step as a big one (a lone 128x192x3072 tile: 66 us; two
    // co-resident: 80 us), so re-tiling the tail buys nothing and the second launch costs 3-5 us; with two batches in flight it is a loss
    // (8 x 60 s: 8.76 -> 9.15 ms).  What does fill a partial round at full CU efficiency is a second tile HEIGHT: tiles 51 / 57 above.
    if (a.tune_tail > 0 && a.m_begin == 0) {
        int bi = -1;
        for (int i = 0; i < NCFG; ++i) if (cfgs[i].id == cfg) bi = i;
        if (bi >= 0 && cost_of(bi, rows) < 1e299) {
            const TileCfg& c = cfgs[bi];
            const long tm = (rows + c.bm - 1) / c.bm, tn = (a.N + c.bn - 1) / c.bn, slots = 256L * c.per_cu;
            const long full = (tm * tn) / slots, rem = tm * tn - full * slots;
            const long main_tiles_m = full * slots / tn;        // whole row tiles inside the full rounds
            const long M1 = main_tiles_m * c.bm;
            if (full >= 1 && rem > 0 && M1 > 0 && M1 < a.M) {
                double tail_cost = 1e300;
                int ti = -1;
                if (a.tune_tail > 0) { for (int i = 0; i < NCFG; ++i) if (cfgs[i].id == a.tune_tail - 1) ti = i; if (ti >= 0) tail_cost = cost_of(ti, a.M - M1); }
                else ti = pick(a.M - M1, &tail_cost);
                // a second launch costs ~GEMM_SPLIT_US of an idle chip: in the model's units (output elements per CU at K) that is
                // GEMM_SPLIT_US x (MACs a CU retires per us) / K
                const double split_cost = (double)full * c.per_cu * c.bm * c.bn / c.eff + tail_cost + GEMM_SPLIT_US * GEMM_CU_MACS_PER_US / a.K;
                const double unsplit = a.tune_cfg > 0 ? cost_of(bi, rows) : whole_cost;
                if (ti >= 0 && tail_cost < 1e299 && (a.tune_tail > 0 || split_cost < 0.95 * unsplit)) {
                    GemmArgs head = a, tail = a;
                    head.M = (int)M1; head.tune_tail = -1;
                    tail.m_begin = (int)M1; tail.tune_tail = -1; tail.tune_cfg = 0;
                    if (launch_tile<EPI, ACT, FMT>(cfg, head, s) != 0) return 1;
                    return launch_tile<EPI, ACT, FMT>(cfgs[ti].id, tail, s);
                }
            }
        }
    }
    return launch_tile<EPI, ACT, FMT>(cfg, a, s);
}

// operand format (GemmArgs::fmt): the fp16 instantiations exist for the epilogues the fp16 forward uses
template <int EPI, int ACT>
static int launch_t(const GemmArgs& a, hipStream_t s) {
    if (a.fmt == FMT_SPLIT) {
        // what the split16 forward launches: erf-GELU 16-bit outputs (convs, FFN1), the projection, q/k/v, the residual
        // GEMMs, and the plain fp32 output of the op-level test entry point
        if constexpr ((EPI == EPI_BF16 && ACT == ACT_GELU_ERF7) || (EPI == EPI_F32 && ACT == ACT_NONE) || EPI == EPI_F32_RESLN ||
                      EPI == EPI_QK || EPI == EPI_PROJ)
            return launch_f<EPI, ACT, FMT_SPLIT>(a, s);
        else { syl_set_error("launch_gemm_bf16", "this epilogue has no split16 instantiation"); return 1; }
    }
    if constexpr (ACT == ACT_GELU_ERF7) { syl_set_error("launch_gemm_bf16", "activation 3 exists for the split16 format only"); return 1; }
    else {
        if (a.fmt == FMT_F16) return launch_f<EPI, ACT, FMT_F16>(a, s);
        return launch_f<EPI, ACT, FMT_BF16>(a, s);
    }
}

int launch_gemm_bf16(int epi, const GemmArgs& a, hipStream_t s) {
    if (a.K % 64 != 0 || a.K <= 0 || a.M <= 0 || a.N <= 0) { syl_set_error("launch_gemm_bf16", "K must be a positive multiple of 64"); return 1; }
    if (a.N % 4 != 0) { syl_set_error("launch_gemm_bf16", "N must be a multiple of 4"); return 1; }
    if (a.fmt == FMT_SPLIT) {
        // the lo planes are reached through the 32-bit scalar offset of the tile's buffer descriptor (unsigned, range-checked
        // against 2^32 - 1 records): plane offset + the largest in-tile offset must stay below 2^32 or the X.lo / W.lo passes
        // would read wrapped addresses.  (~21 M samples per batch on the conv stack; beyond that: split the batch.)
        const unsigned long long span_x = 256ull * (unsigned long long)a.ldx * 2 + (unsigned long long)a.K * 2;
        const unsigned long long span_w = 256ull * (unsigned long long)a.K * 2 + (unsigned long long)a.K * 2;
        if (a.x_lo < 0 || a.w_lo < 0 || (unsigned long long)a.x_lo * 2 + span_x >= (1ull << 32) || (unsigned long long)a.w_lo * 2 + span_w >= (1ull << 32)) {
            syl_set_error("launch_gemm_bf16", "split16: the lo-plane offset of an operand does not fit the 32-bit buffer offset (batch too large for one launch; split the batch)");
            return 1;
        }
    }
    switch (epi) {
        case EPI_BF16:
            if (a.act == 1) return launch_t<EPI_BF16, 1>(a, s);
            if (a.act == 2) return launch_t<EPI_BF16, 2>(a, s);
            if (a.act == 3) return launch_t<EPI_BF16, 3>(a, s);
            return launch_t<EPI_BF16, 0>(a, s);
        case EPI_F32:
            if (a.act == 1) return launch_t<EPI_F32, 1>(a, s);
            if (a.act == 2) return launch_t<EPI_F32, 2>(a, s);
            return launch_t<EPI_F32, 0>(a, s);
        case EPI_F32_RES: return launch_t<EPI_F32_RES, 0>(a, s);
        case EPI_F32_RESLN: return launch_t<EPI_F32_RESLN, 0>(a, s);
        case EPI_QK: return launch_t<EPI_QK, 0>(a, s);
        case EPI_PROJ: return launch_t<EPI_PROJ, 0>(a, s);
    }
    syl_set_error("launch_gemm_bf16", "unknown epilogue");
    return 1;
}
