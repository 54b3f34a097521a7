// MXFP8 GEMM for gfx950 (BASELINE.json configs[4]: fp8 MFMA for the FFN GEMMs):
//     out[m][n] = sum_k (X8[m][k] * 2^(XS[m][k/32]-127)) * (W8[n][k] * 2^(WS[n][k/32]-127))   (+ fused epilogue)
// Operands are OCP microscaling FP8: e4m3 elements with one E8M0 power-of-two scale per 32 elements along K.  The
// contraction runs on v_mfma_scale_f32_32x32x64_f8f6f4 — the only gfx950 matrix instruction that runs fp8 at twice
// the bf16 rate — which applies the block scales in hardware (fp32 accumulation).
//
// Operand layout of that instruction, probed on the hardware (tools/ubench/mfma_f8_probe.hip):
//   * lane l = (r = l & 31, h = l >> 5) supplies row r of its operand; its 32 bytes are k = 16h .. 16h+15 (registers
//     0-3) and k = 32+16h .. 32+16h+15 (registers 4-7) of the 64-wide K slice;
//   * the scale of (row r, K block h) is taken from lane (r, h): byte OPSEL of that lane's scale register;
//   * C/D layout as for every 32x32 MFMA: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h.
//
// Structure (4 waves as 2x2, two workgroups per CU): tile (64 FM) x (64 FN), K step 128 BYTES — the same 128-byte
// LDS rows, global_load_lds_dwordx4 staging and source-side XOR swizzle as the bf16 kernel (gemm_bf16.hip), so a
// stage moves the same bytes but feeds twice the FLOPs.  2-slot ring with ONE barrier per K step: all fragments of
// tile t are read into registers, then (barrier) the slot is refilled with tile t+2 while the MFMAs of tile t issue,
// the LDS-DMA instructions interleaved between them.  Scale words (4 bytes = the 4 K blocks of a stage per row) go
// global -> VGPR one stage ahead (they are 3 % of the operand bytes and every lane needs exactly its own rows').
// "Swapped" orientation (A = weights, B = activations): a lane owns one token and runs of 4 output features.
#include "kernels.h"
#include "gemm_epilogue.h"
#include "gemm_epilogue_f8.h"

typedef __attribute__((address_space(3))) void* lds_vptr8;
typedef const __attribute__((address_space(1))) void* glb_vptr8;
typedef int v8i_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16g(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((glb_vptr8)g, (lds_vptr8)l, 16, 0, 0);
}
// the buffer-descriptor form (see gemm_bf16.hip): constant per-lane offset, the K step in the scalar offset
__device__ __forceinline__ __amdgpu_buffer_rsrc_t f8_make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0xffffffffu, 0x00020000);
}
__device__ __forceinline__ void f8_glds16b(__amdgpu_buffer_rsrc_t rs, int voff, int soff, void* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr8)l, 16, voff, soff, 0, 0);
}
#define F8_FENCE() __builtin_amdgcn_sched_barrier(0)

// ---- quantiser: f32 rows -> MXFP8 (used for the weights at create time and by the op-level entry point) -------
__global__ __launch_bounds__(256) void mx_quant_rows_kernel(const float* __restrict__ in, long ld_in, uint8_t* __restrict__ out,
                                                            long ld_out, uint8_t* __restrict__ sc, long sc_rows, int R, int K) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int qpr = K >> 2;                         // quads per row (multiple of 8)
    const long r = idx / qpr;
    const int c = (int)(idx - r * qpr) * 4;
    const bool ok = r < R;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) v = *(const float4*)(in + r * ld_in + c);
    float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    const unsigned e = mx_e8m0(amax);
    const float inv = mx_inv_scale(e);
    if (!ok) return;
    *(unsigned*)(out + r * ld_out + c) = pack_fp8x4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
    if ((threadIdx.x & 7) == 0) sc[mx_scale_index(r, c >> 5, sc_rows)] = (uint8_t)e;
}

int launch_mx_quant_rows(const float* in, long ld_in, uint8_t* out, long ld_out, uint8_t* sc, long sc_rows, int R, int K, hipStream_t s) {
    if (K % 64 != 0 || K <= 0 || R <= 0) { syl_set_error("launch_mx_quant_rows", "K must be a positive multiple of 64"); return 1; }
    const long quads = (long)R * (K / 4);
    hipLaunchKernelGGL(mx_quant_rows_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, in, ld_in, out, ld_out, sc, sc_rows, R, K);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int FM, int FN, int EPI, int ACT>
__global__ __launch_bounds__(256, 2) void gemm_mxfp8_kernel(const GemmF8Args a) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    constexpr int BM = 64 * FM, BN = 64 * FN;
    constexpr int RB = 128;                  // bytes (= fp8 elements) per LDS row = K step
    constexpr int XT = BM * RB, WT = BN * RB, STAGE = XT + WT;
    constexpr int NP = (BM + BN) / 8;        // 1-KiB pieces per stage (8 rows each)
    constexpr int NPW = NP / 4;
    constexpr int NMT = 2 * FM * FN;         // MFMAs per stage per wave
    static_assert(NP % 4 == 0, "tile must split evenly over 4 waves");
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int M = a.g.M, N = a.g.N, K = a.g.K;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int wg = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;

    // ---- staging (identical byte geometry to the bf16 kernel's 128-byte rows)
    const int srow = lane >> 3, spos = lane & 7;
    static_assert((BM / 8) % 4 == 0, "X pieces must split evenly over the 4 waves");
    constexpr int XPW = BM / 8 / 4;
    const __amdgpu_buffer_rsrc_t rx = f8_make_rsrc(a.X8 + (size_t)m0 * a.ldx8), rw = f8_make_rsrc(a.W8 + (size_t)n0 * K);
    int voff[NPW];
    int lds_off[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int p = wave + 4 * i;
        const bool isx = i < XPW;
        const int r = (isx ? p : p - BM / 8) * 8 + srow;
        const int c = spos ^ ((r >> 1) & 7);
        if (isx) { int xm = m0 + r; xm = xm < M ? xm : M - 1; voff[i] = (int)((long)(xm - m0) * a.ldx8 + c * 16); }
        else { int wr = n0 + r; wr = wr < N ? wr : N - 1; voff[i] = (wr - n0) * K + c * 16; }
        lds_off[i] = (isx ? 0 : XT) + (isx ? p : p - BM / 8) * 1024;
    }
    auto dma1 = [&](int kt, int i, char* base) { f8_glds16b(i < XPW ? rx : rw, voff[i], kt * RB, base + lds_off[i]); };
    auto stage = [&](int kt, int slot) {
        char* base = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < NPW; ++i) dma1(kt, i, base);
    };

    // ---- fragment addresses: MFMA j of a stage reads 16-byte chunks 4j + h and 4j + 2 + h of its row
    const int frow = lane & 31, fhalf = lane >> 5;
    const int swz = (lane >> 1) & 7;
    int koff[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { koff[j][0] = ((4 * j + fhalf) ^ swz) << 4; koff[j][1] = ((4 * j + 2 + fhalf) ^ swz) << 4; }
    const int xrow_off = (wm * 32 * FM + frow) * RB;
    const int wrow_off = XT + (wn * 32 * FN + frow) * RB;

    // ---- scale words: the 4 K blocks of a stage per row = two adjacent-pair u16 loads, straight to registers
    const uint8_t* xsp[FM]; const uint8_t* wsp[FN];
#pragma unroll
    for (int f = 0; f < FM; ++f) { int r = m0 + wm * 32 * FM + f * 32 + frow; r = r < M ? r : M - 1; xsp[f] = a.XS + (size_t)r * 2; }
#pragma unroll
    for (int f = 0; f < FN; ++f) { int r = n0 + wn * 32 * FN + f * 32 + frow; r = r < N ? r : N - 1; wsp[f] = a.WS + (size_t)r * 2; }
    const size_t xs_step = (size_t)a.xs_rows * 2, ws_step = (size_t)a.ws_rows * 2;      // bytes per 64-wide K slice
    unsigned xs[FM], ws[FN], xs_n[FM], ws_n[FN];
    auto load_scales = [&](int kt, unsigned (&x)[FM], unsigned (&w)[FN]) {
#pragma unroll
        for (int f = 0; f < FM; ++f)
            x[f] = (unsigned)*(const unsigned short*)(xsp[f] + (size_t)(2 * kt) * xs_step) |
                   ((unsigned)*(const unsigned short*)(xsp[f] + (size_t)(2 * kt + 1) * xs_step) << 16);
#pragma unroll
        for (int f = 0; f < FN; ++f)
            w[f] = (unsigned)*(const unsigned short*)(wsp[f] + (size_t)(2 * kt) * ws_step) |
                   ((unsigned)*(const unsigned short*)(wsp[f] + (size_t)(2 * kt + 1) * ws_step) << 16);
    };

    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = K / RB;
    stage(0, 0);
    if (nt > 1) stage(1, 1);
    load_scales(0, xs_n, ws_n);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // tiles 0 and 1 are in LDS
    for (int t = 0; t < nt; ++t) {
        const int slot = t & 1;
        const char* sb = smem + slot * STAGE;
        // scales of tile t (loaded one iteration ago); lane half h uses bytes h and 2 + h -> shift once, OPSEL 0 / 2
#pragma unroll
        for (int f = 0; f < FM; ++f) xs[f] = xs_n[f] >> (8 * fhalf);
#pragma unroll
        for (int f = 0; f < FN; ++f) ws[f] = ws_n[f] >> (8 * fhalf);
        v8i_t xf[2][FM], wf[2][FN];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int f = 0; f < FM; ++f) {
                const uint4 lo = *(const uint4*)(sb + xrow_off + f * 32 * RB + koff[j][0]);
                const uint4 hi = *(const uint4*)(sb + xrow_off + f * 32 * RB + koff[j][1]);
                xf[j][f] = (v8i_t){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
            }
#pragma unroll
            for (int f = 0; f < FN; ++f) {
                const uint4 lo = *(const uint4*)(sb + wrow_off + f * 32 * RB + koff[j][0]);
                const uint4 hi = *(const uint4*)(sb + wrow_off + f * 32 * RB + koff[j][1]);
                wf[j][f] = (v8i_t){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
            }
        }
        // ONE barrier per K step: behind it (a) every wave holds its fragments of tile t in registers, so this slot
        // may be refilled with tile t+2, and (b) every wave's pieces of tile t+1 (issued under the MFMAs of tile t-1)
        // have landed, so the next iteration may read the other slot
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        F8_FENCE();
        if (t + 1 < nt) load_scales(t + 1, xs_n, ws_n);
        const bool refill = t + 2 < nt;
        char* dbase = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < NMT; ++i) {
            const int j = i / (FM * FN), fm = (i % (FM * FN)) / FN, fn = i % FN;
            if (j == 0)
                acc[fm][fn] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[0][fn], xf[0][fm], acc[fm][fn], 0, 0, 0, (int)ws[fn], 0, (int)xs[fm]);
            else
                acc[fm][fn] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[1][fn], xf[1][fm], acc[fm][fn], 0, 0, 2, (int)ws[fn], 2, (int)xs[fm]);
            // LDS-DMA pieces of tile t+2 spread evenly between the MFMAs
#pragma unroll
            for (int p = 0; p < NPW; ++p) {
                if (((p + 1) * NMT + NPW - 1) / NPW - 1 == i) {
                    F8_FENCE();
                    if (refill) dma1(t + 2, p, dbase);
                    F8_FENCE();
                }
            }
        }
    }

    // ---- epilogue
    if constexpr (EPI == EPI_MXFP8) {
        static_assert(4 * StagedF8<FN>::BYTES <= 2 * STAGE, "epilogue staging must fit the ring");
        __builtin_amdgcn_s_barrier();
        char* my = smem + wave * StagedF8<FN>::BYTES;
        float4 bias4[FN][4];
        load_colvec<FN>(a.g.bias, n0 + wn * 32 * FN, lane >> 5, a.g.N, bias4);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
            epilogue_mxfp8_rows32<FN, ACT>(a, acc[fm], bias4, m0 + wm * 32 * FM + fm * 32, n0 + wn * 32 * FN, my, lane);
    } else {
        epilogue_direct<FM, FN, EPI, ACT>(a.g, acc, m0 + wm * 32 * FM, n0 + wn * 32 * FN, lane);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// 8-wave variant (one workgroup per CU, tiles 256x256 / 256x192): the structure of gemm8_bf16_kernel (gemm_bf16.hip)
// — K step = one 64-byte LDS row = ONE 64-wide MFMA per fragment pair, 4-slot ring, the two wave groups staggered by
// one barrier (one in the matrix pipe while the other reads LDS) — so a step moves the same bytes and costs the same
// MFMA cycles as the bf16 kernel's 32-wide step but contracts twice the K.  The scales ride along as ONE extra
// 1-KiB LDS-DMA piece per step: lanes 0-31 fetch the X scales of 8 rows each (16 contiguous bytes in the
// K-pair-major layout), lanes 32-63 the W scales; a lane then reads its (row, half) byte with ds_read_u8.
template <int N> __device__ __forceinline__ void f8_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int FM, int FN, int WM, int WN, int EPI, int ACT>
__global__ __launch_bounds__(512, 2) void gemm8_mxfp8_kernel(const GemmF8Args a) {
    constexpr int BM = 32 * FM * WM, BN = 32 * FN * WN;
    constexpr int RB = 64;
    constexpr int XT = BM * RB, WT = BN * RB, SC = 1024, STAGE = XT + WT + SC;
    constexpr int NPT = (BM + BN) / 16;              // operand pieces (16 rows x 64 B) per step
    constexpr int NP = NPT + 1;                      // + the scale piece
    constexpr int NPW_HI = (NP + 7) / 8, NPW_LO = NP / 8;
    constexpr int NMF = FM * FN;
    static_assert(WM * WN == 8, "8 waves");
    static_assert(BM <= 256 && BN <= 256, "one scale piece covers 256 + 256 rows");
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int group = wave >> 2;
    const int wm = wave / WN, wn = wave % WN;
    const bool hi = wave < (NP % 8 == 0 ? 8 : NP % 8);
    const int M = a.g.M, N = a.g.N, K = a.g.K;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int wg = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;

    const int srow = lane >> 2, spos = lane & 3;
    // operand pieces go through buffer descriptors (one per operand, based at the tile's first row): the per-lane
    // offset is a constant 32-bit VGPR and the K step rides in the scalar offset, so a step costs no address VALU.
    // The scale piece keeps the flat form: its two lane halves read two different arrays.
    static_assert((BM / 16) % 8 == 0, "X pieces must split evenly over the 8 waves");
    constexpr int XPW = BM / 16 / 8;                 // pieces wave + 8 i with i < XPW are X rows for every wave
    const __amdgpu_buffer_rsrc_t rx = f8_make_rsrc(a.X8 + (size_t)m0 * a.ldx8), rw = f8_make_rsrc(a.W8 + (size_t)n0 * K);
    int voff[NPW_HI];
    int lds_off[NPW_HI];
    const uint8_t* gsc;                              // scale piece source of this lane and its bytes per K step
    long gsc_step;
    if (lane < 32) { long r = m0 + 8 * lane; r = r + 8 <= a.xs_rows ? r : a.xs_rows - 8; gsc = a.XS + r * 2; gsc_step = a.xs_rows * 2; }
    else { long r = n0 + 8 * (lane - 32); r = r + 8 <= a.ws_rows ? r : a.ws_rows - 8; gsc = a.WS + r * 2; gsc_step = a.ws_rows * 2; }
    bool is_sc[NPW_HI];
#pragma unroll
    for (int i = 0; i < NPW_HI; ++i) {
        int p = wave + 8 * i;
        p = p < NP ? p : NP - 1;
        is_sc[i] = p == NPT;
        if (p == NPT) { voff[i] = 0; lds_off[i] = XT + WT; }
        else {
            const bool isx = i < XPW;
            const int r = (isx ? p : p - BM / 16) * 16 + srow;
            const int c = spos ^ ((r >> 2) & 3);
            if (isx) { int xm = m0 + r; xm = xm < M ? xm : M - 1; voff[i] = (int)((long)(xm - m0) * a.ldx8 + c * 16); }
            else { int wr = n0 + r; wr = wr < N ? wr : N - 1; voff[i] = (wr - n0) * K + c * 16; }
            lds_off[i] = (isx ? 0 : XT) + (isx ? p : p - BM / 16) * 1024;
        }
    }
    auto dma1 = [&](int ks, int i, char* base) {
        // only the last piece slot of a wave can be the scale piece (wave-uniform)
        if (i == NPW_HI - 1 && is_sc[i]) glds16g(gsc + (size_t)ks * gsc_step, base + lds_off[i]);
        else f8_glds16b(i < XPW ? rx : rw, voff[i], ks * RB, base + lds_off[i]);
    };
    auto stage = [&](int ks, int slot) {
        char* base = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < NPW_HI; ++i)
            if (i < NPW_LO || hi) dma1(ks, i, base);
    };
    auto wait_steps = [&](int nsteps_in_flight) {
        if (hi) {
            if (nsteps_in_flight >= 2) f8_wait_vmcnt<2 * NPW_HI>(); else if (nsteps_in_flight == 1) f8_wait_vmcnt<NPW_HI>(); else f8_wait_vmcnt<0>();
        } else {
            if (nsteps_in_flight >= 2) f8_wait_vmcnt<2 * NPW_LO>(); else if (nsteps_in_flight == 1) f8_wait_vmcnt<NPW_LO>(); else f8_wait_vmcnt<0>();
        }
    };

    const int frow = lane & 31;
    const int swz = (lane >> 2) & 3;
    const int fhalf = lane >> 5;
    const int koff0 = (((0 + fhalf) ^ swz) << 4), koff1 = (((2 + fhalf) ^ swz) << 4);
    const int xrow_off = (wm * 32 * FM + frow) * RB;
    const int wrow_off = XT + (wn * 32 * FN + frow) * RB;
    const int xsc_off = XT + WT + (wm * 32 * FM + frow) * 2 + fhalf;
    const int wsc_off = XT + WT + 512 + (wn * 32 * FN + frow) * 2 + fhalf;

    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = K / 64;
    stage(0, 0);
    if (nt > 1) stage(1, 1);
    if (nt > 2) stage(2, 2);
    wait_steps(nt > 2 ? 2 : nt - 1);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    if (group == 1) __builtin_amdgcn_s_barrier();    // stagger: group 1 runs one barrier behind
    int slot = 0;
    for (int s = 0; s < nt; ++s) {
        const char* sb = smem + slot * STAGE;
        // ---- A: fragments + scales of step s; retire step s+1
        F8_FENCE();
        v8i_t xf[FM], wf[FN];
        int xsc[FM], wsc[FN];
#pragma unroll
        for (int f = 0; f < FM; ++f) {
            const uint4 lo = *(const uint4*)(sb + xrow_off + f * 32 * RB + koff0);
            const uint4 hi4 = *(const uint4*)(sb + xrow_off + f * 32 * RB + koff1);
            xf[f] = (v8i_t){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi4.x, (int)hi4.y, (int)hi4.z, (int)hi4.w};
            xsc[f] = *(const uint8_t*)(sb + xsc_off + f * 64);
        }
#pragma unroll
        for (int f = 0; f < FN; ++f) {
            const uint4 lo = *(const uint4*)(sb + wrow_off + f * 32 * RB + koff0);
            const uint4 hi4 = *(const uint4*)(sb + wrow_off + f * 32 * RB + koff1);
            wf[f] = (v8i_t){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi4.x, (int)hi4.y, (int)hi4.z, (int)hi4.w};
            wsc[f] = *(const uint8_t*)(sb + wsc_off + f * 64);
        }
        {
            const int after = nt - 2 - s;
            wait_steps(after >= 1 ? 1 : 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        F8_FENCE();
        __builtin_amdgcn_s_barrier();
        F8_FENCE();
        // ---- B: MFMAs of step s with the LDS-DMA of step s+3 spread between them (slot of step s-1)
        const bool dma = s + 3 < nt;
        char* dbase = smem + ((slot + 3) & 3) * STAGE;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
            const int fm = i / FN, fn = i % FN;
            acc[fm][fn] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[fn], xf[fm], acc[fm][fn], 0, 0, 0, wsc[fn], 0, xsc[fm]);
#pragma unroll
            for (int q = 0; q < NPW_HI; ++q) {
                if (((q + 1) * NMF + NPW_HI - 1) / NPW_HI - 1 == i) {
                    F8_FENCE();
                    if (dma && (q < NPW_LO || hi)) dma1(s + 3, q, dbase);
                    F8_FENCE();
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        F8_FENCE();
        __builtin_amdgcn_s_barrier();
        slot = (slot + 1) & 3;
    }
    if (group == 0) __builtin_amdgcn_s_barrier();    // pairs with group 1's extra barrier

    if constexpr (EPI == EPI_MXFP8) {
        static_assert(8 * StagedF8<FN>::BYTES <= 4 * STAGE, "epilogue staging must fit the ring");
        __builtin_amdgcn_s_barrier();
        char* my = smem + wave * StagedF8<FN>::BYTES;
        float4 bias4[FN][4];
        load_colvec<FN>(a.g.bias, n0 + wn * 32 * FN, lane >> 5, a.g.N, bias4);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
            epilogue_mxfp8_rows32<FN, ACT>(a, acc[fm], bias4, m0 + wm * 32 * FM + fm * 32, n0 + wn * 32 * FN, my, lane);
    } else if constexpr (EPI == EPI_QK) {
        static_assert(8 * StagedEpi<FN, EPI>::BYTES <= 4 * STAGE, "epilogue staging must fit the ring");
        __builtin_amdgcn_s_barrier();
        char* my = smem + wave * StagedEpi<FN, EPI>::BYTES;
        epilogue_staged<FM, FN, EPI, ACT>(a.g, acc, m0 + wm * 32 * FM, n0 + wn * 32 * FN, my, lane);
    } else {
        epilogue_direct<FM, FN, EPI, ACT>(a.g, acc, m0 + wm * 32 * FM, n0 + wn * 32 * FN, lane);
    }
}

template <int FM, int FN, int WM, int WN, int EPI, int ACT>
static int launch_f8_8(const GemmF8Args& a, hipStream_t s) {
    constexpr int BM = 32 * FM * WM, BN = 32 * FN * WN;
    constexpr int LDS = 4 * ((BM + BN) * 64 + 1024);
    if ((a.xs_rows & 7) || (a.ws_rows & 7) || a.xs_rows < 8 || a.ws_rows < 8) { syl_set_error("launch_gemm_mxfp8", "scale pitches must be multiples of 8"); return 1; }
    const int tiles = ((a.g.M + BM - 1) / BM) * ((a.g.N + BN - 1) / BN);
    static PerDeviceOnce attr_once;
    auto kern = gemm8_mxfp8_kernel<FM, FN, WM, WN, EPI, ACT>;
    if (attr_once.need()) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int FM, int FN, int EPI, int ACT>
static int launch_f8(const GemmF8Args& a, hipStream_t s) {
    constexpr int BM = 64 * FM, BN = 64 * FN;
    constexpr int LDS = 2 * (BM + BN) * 128;
    const int tiles = ((a.g.M + BM - 1) / BM) * ((a.g.N + BN - 1) / BN);
    static PerDeviceOnce attr_once;
    auto kern = gemm_mxfp8_kernel<FM, FN, EPI, ACT>;
    if (attr_once.need()) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// tile configurations: 0 = 128x192 (4 waves, 2 workgroups/CU), 1 = 128x128 (same), 2 = 256x256 (8 waves), 3 = 256x192
// (8 waves); -1 = automatic: the 8-wave tile whose launch needs the fewest rounds over 256 CUs
template <int EPI, int ACT>
static int launch_f8_t(const GemmF8Args& a, hipStream_t s) {
    int cfg = a.g.tune_cfg > 0 ? a.g.tune_cfg - 1 : -1;
    // 85 / 91: the hand-scheduled X3 loop (gemm_asm_f8.hip: 256x256 / 256x192, whole tiles, K % 256 == 0); automatic wherever it applies
    {
        const int t = gemm_asm_f8_tile(EPI, a);
        // the quantising GELU GEMM (FFN1) stays on the 8-wave kernel unless forced: its epilogue wants two waves per SIMD (66 vs 71-74 us)
        const bool auto_ok = t && !(EPI == EPI_MXFP8 && a.g.N % 256 == 0);
        if ((cfg < 0 && auto_ok) || ((cfg == 85 || cfg == 91 || cfg == 92 || cfg == 93) && t)) {
            int use = cfg < 0 ? t : cfg;
            if (use == 85 && a.g.N % 256 != 0) use = 91;
            if (use != 85 && a.g.N % 192 != 0) use = 85;
            return launch_gemm_asm_f8(EPI, a, s, use);
        }
        if (cfg == 85 || cfg == 91 || cfg == 92 || cfg == 93) cfg = -1;   // not applicable: the automatic choice among the older kernels
    }
    if constexpr (EPI == EPI_QK) { if (cfg == 0 || cfg == 1) cfg = -1; }   // 8-wave kernels only
    if (cfg < 0) {
        const long t2 = (long)((a.g.M + 255) / 256) * ((a.g.N + 255) / 256), t3 = (long)((a.g.M + 255) / 256) * ((a.g.N + 191) / 192);
        const double c2 = (double)((t2 + 255) / 256) * 256 * 256, c3 = (double)((t3 + 255) / 256) * 256 * 192;
        cfg = c2 <= c3 ? 2 : 3;
    }
    if (cfg == 2) return launch_f8_8<4, 2, 2, 4, EPI, ACT>(a, s);
    if (cfg == 3) return launch_f8_8<2, 3, 4, 2, EPI, ACT>(a, s);
    if constexpr (EPI != EPI_QK) {
        if (cfg == 1) return launch_f8<2, 2, EPI, ACT>(a, s);
        return launch_f8<2, 3, EPI, ACT>(a, s);
    }
    return 1;
}

int launch_gemm_mxfp8(int epi, const GemmF8Args& a, hipStream_t s) {
    if (a.g.K % 128 != 0 || a.g.K <= 0 || a.g.M <= 0 || a.g.N <= 0) { syl_set_error("launch_gemm_mxfp8", "K must be a positive multiple of 128"); return 1; }
    if (a.g.N % 4 != 0) { syl_set_error("launch_gemm_mxfp8", "N must be a multiple of 4"); return 1; }
    if (a.g.K % 128 != 0) { syl_set_error("launch_gemm_mxfp8", "K must be a multiple of 128"); return 1; }
    if (a.ldx8 & 15) { syl_set_error("launch_gemm_mxfp8", "operand rows must be 16-byte aligned"); return 1; }
    switch (epi) {
        case EPI_MXFP8:
            if (a.g.N % 32 != 0) { syl_set_error("launch_gemm_mxfp8", "MXFP8 output needs N % 32 == 0"); return 1; }
            if (a.g.act == 1) return launch_f8_t<EPI_MXFP8, 1>(a, s);
            return launch_f8_t<EPI_MXFP8, 0>(a, s);
        case EPI_F32:
            if (a.g.act == 1) return launch_f8_t<EPI_F32, 1>(a, s);
            return launch_f8_t<EPI_F32, 0>(a, s);
        case EPI_F32_RESLN: return launch_f8_t<EPI_F32_RESLN, 0>(a, s);
        case EPI_QK: return launch_f8_t<EPI_QK, 0>(a, s);
        case EPI_QK8:                                          // MXFP8 q / k / V^T for the fp8 attention core: the asm 256x192 tile only
            if (!gemm_asm_f8_tile(EPI_QK8, a)) { syl_set_error("launch_gemm_mxfp8", "EPI_QK8 needs whole 256-row tiles, N = 2304, K % 256 == 0"); return 1; }
            return launch_gemm_asm_f8(EPI_QK8, a, s, 91);
    }
    syl_set_error("launch_gemm_mxfp8", "unsupported epilogue");
    return 1;
}
