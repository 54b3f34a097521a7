// Positional convolution embedding of the HuBERT encoder for gfx950.
//
// Reference (transformers HubertPositionalConvEmbedding + HubertSamePadLayer, TP:45-103, called from
// HubertEncoder.forward TP:439-440, reached from sylber/model/sylber.py:122):
//     pos = GELU( Conv1d(768, 768, k=128, padding=64, groups=16)(x^T) [:, :, :-1] )^T ;  x = x + pos
// (weight-norm is folded into the weights once at load time).
//
// Per group this is a Toeplitz contraction: out[t][n] = sum_tap sum_c x[t + tap - 64][c] * w[n][c][tap]
// with 48 channels in / out per group.  Rows t and t+1 share 127/128 of their input, so instead of an
// im2col GEMM (K = 6144 per row) the workgroup stages ONE window of 383 frames (256 + 127) x 48 channels in LDS and
// reads the B fragments of all 128 taps from it at shifted row offsets; the per-tap weight slabs
// (64 n x 48 c, n padded 48->64, rows padded to 112 B so ds_read_b128 is conflict-free) stream through
// a double-buffered LDS ring with global_load_lds_dwordx4, two taps per barrier.
// Orientation is "lane = frame" (A = weights, B = activations) so the epilogue adds the fp32 residual
// and stores 16-byte runs.
#include "kernels.h"

#define PC_BM 256                 // frames per workgroup (8 waves x 32): a weight slab read from L2 serves 256 frames
#define PC_XROW 112               // bytes per staged x row (48 bf16 + 16 B pad)
#define PC_XWIN (384 * PC_XROW)   // 43008: the 383-frame window of 256 frames x 128 taps
#define PC_SLAB (64 * 112)        // bytes of one (group, tap) weight slab: 7168
#define PC_TAPS_PER_STEP 2
#define PC_STEP (PC_TAPS_PER_STEP * PC_SLAB)   // 14336
#define PC_LDS (PC_XWIN + 2 * PC_STEP)         // 71680: two workgroups per CU

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;
// LDS-DMA through a buffer descriptor: constant per-lane offset, the weight step in the scalar offset
__device__ __forceinline__ void glds16p(__amdgpu_buffer_rsrc_t rs, int voff, int soff, void* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)l, 16, voff, soff, 0, 0);
}

template <int ACT, int FMT>
__global__ __launch_bounds__(512, 4) void posconv_bf16_kernel(const bf16_t* __restrict__ xpad, const bf16_t* __restrict__ wpk,
                                                              const float* __restrict__ bias, const float* __restrict__ x_f32,
                                                              float* __restrict__ out, int Tp, long x_lo, long w_lo) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    char* xwin = smem;
    char* wring = smem + PC_XWIN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ql = lane & 31, h = lane >> 5;
    const int t0 = blockIdx.x * PC_BM, g = blockIdx.y, b = blockIdx.z;
    const int rows_per_b = Tp + 128;

    f32x16_t acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // FMT_SPLIT: three passes into the same accumulators -- x.hi w.hi, x.lo w.hi, x.hi w.lo (hi / lo half planes)
    constexpr int NPASS = FMT == FMT_SPLIT ? 3 : 1;
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
    if (pass > 0) __syncthreads();                     // the previous pass's window and ring reads are done
    // ---- stage the x window: rows t0 .. t0+382 of xpad[b], channels g*48 .. +47
    {
        const bf16_t* xb = xpad + (pass == 1 ? x_lo : 0L) + (size_t)b * rows_per_b * SYL_HIDDEN + g * SYL_POSC;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int idx = tid + 512 * i;         // 384 rows x 6 16-byte chunks = 2304 (4.5 per thread)
            if (idx < 384 * 6) {
                const int r = idx / 6, ch = idx - r * 6;
                int row = t0 + r; row = row < rows_per_b ? row : rows_per_b - 1;
                const uint4 v = *(const uint4*)(xb + (size_t)row * SYL_HIDDEN + ch * 8);
                *(uint4*)(xwin + r * PC_XROW + ch * 16) = v;
            }
        }
    }
    // ---- weight ring: a step = 2 taps = 14 KiB = 14 wave-instructions; wave w issues pieces w and w + 8
    const char* wg_base = (const char*)(wpk + (pass == 2 ? w_lo : 0L)) + (size_t)g * SYL_POSK * PC_SLAB;
    const __amdgpu_buffer_rsrc_t rwg = __builtin_amdgcn_make_buffer_rsrc((void*)wg_base, 0, (int)0xffffffffu, 0x00020000);
    auto stage = [&](int step, int buf) {
        char* dst = wring + buf * PC_STEP;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wave + 8 * i;
            if (piece < 14) glds16p(rwg, piece * 1024 + lane * 16, step * PC_STEP, dst + piece * 1024);
        }
    };

    const int xfrag = (wave * 32 + ql) * PC_XROW + h * 16;
    const int wfrag = ql * 112 + h * 16;
    const int nsteps = SYL_POSK / PC_TAPS_PER_STEP;
    stage(0, 0);
    for (int st = 0; st < nsteps; ++st) {
        // LDS-DMA completion is NOT covered by __syncthreads(): retire this wave's pieces of step st explicitly
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();      // step st visible; also covers the x window stores
        if (st + 1 < nsteps) stage(st + 1, (st + 1) & 1);
        const char* wb = wring + (st & 1) * PC_STEP;
#pragma unroll
        for (int tt = 0; tt < PC_TAPS_PER_STEP; ++tt) {
            const int tap = st * PC_TAPS_PER_STEP + tt;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const bf16x8_t xf = *(const bf16x8_t*)(xwin + xfrag + tap * PC_XROW + q * 32);
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) {
                    const bf16x8_t wf = *(const bf16x8_t*)(wb + tt * PC_SLAB + nf * 32 * 112 + wfrag + q * 32);
                    acc[nf] = H16<FMT>::mfma(wf, xf, acc[nf]);
                }
            }
        }
    }
    }   // pass
    // ---- epilogue: out = x + gelu(conv + bias); lane owns frame t, runs of 4 output channels
    const int t = t0 + wave * 32 + ql;
    if (t < Tp) {
        const size_t m = (size_t)b * Tp + t;
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int nl = 32 * nf + 8 * gg + 4 * h;
                if (nl >= SYL_POSC) continue;
                const int n = g * SYL_POSC + nl;
                const float4 bb = *(const float4*)(bias + n);
                const float4 rr = *(const float4*)(x_f32 + m * SYL_HIDDEN + n);
                float v0 = acc[nf][4 * gg + 0] + bb.x, v1 = acc[nf][4 * gg + 1] + bb.y;
                float v2 = acc[nf][4 * gg + 2] + bb.z, v3 = acc[nf][4 * gg + 3] + bb.w;
                if constexpr (ACT == 2 && FMT == FMT_SPLIT) { v0 = gelu_erf7(v0); v1 = gelu_erf7(v1); v2 = gelu_erf7(v2); v3 = gelu_erf7(v3); }
                else if constexpr (ACT == 2) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
                else { v0 = gelu_fast(v0); v1 = gelu_fast(v1); v2 = gelu_fast(v2); v3 = gelu_fast(v3); }
                *(float4*)(out + m * SYL_HIDDEN + n) = make_float4(rr.x + v0, rr.y + v1, rr.z + v2, rr.w + v3);
            }
    }
}

int launch_posconv(const bf16_t* xpad, const bf16_t* wpk, const float* bias, const float* x_f32, float* out, int B, int Tp,
                   int act, hipStream_t s, int fmt, long x_lo, long w_lo) {
    dim3 grid((Tp + PC_BM - 1) / PC_BM, SYL_POSG, B);
    static PerDeviceOnce attr_once;
    if (attr_once.need()) {
        HIP_TRY(hipFuncSetAttribute((const void*)posconv_bf16_kernel<1, FMT_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS));
        HIP_TRY(hipFuncSetAttribute((const void*)posconv_bf16_kernel<2, FMT_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS));
        HIP_TRY(hipFuncSetAttribute((const void*)posconv_bf16_kernel<1, FMT_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS));
        HIP_TRY(hipFuncSetAttribute((const void*)posconv_bf16_kernel<2, FMT_SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS));
    }
    if (fmt == FMT_SPLIT) {
        hipLaunchKernelGGL((posconv_bf16_kernel<2, FMT_SPLIT>), grid, dim3(512), PC_LDS, s, xpad, wpk, bias, x_f32, out, Tp, x_lo, w_lo);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (fmt == FMT_F16 && act == 2) { syl_set_error("launch_posconv", "the erf GELU (act 2) has no fp16 instantiation"); return 1; }
    if (fmt == FMT_F16) hipLaunchKernelGGL((posconv_bf16_kernel<1, FMT_F16>), grid, dim3(512), PC_LDS, s, xpad, wpk, bias, x_f32, out, Tp, 0L, 0L);
    else if (act == 2) hipLaunchKernelGGL((posconv_bf16_kernel<2, FMT_BF16>), grid, dim3(512), PC_LDS, s, xpad, wpk, bias, x_f32, out, Tp, 0L, 0L);
    else hipLaunchKernelGGL((posconv_bf16_kernel<1, FMT_BF16>), grid, dim3(512), PC_LDS, s, xpad, wpk, bias, x_f32, out, Tp, 0L, 0L);
    HIP_TRY(hipGetLastError());
    return 0;
}
