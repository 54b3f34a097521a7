// fp32 PARITY MODE (precision = SYLBER_FP32): the same forward path with every contraction in exact fp32
// (v_mfma_f32_32x32x2_f32 = k-ordered fmaf chain, 1/16 of the bf16 MFMA rate) and erf-GELU, so that the
// hidden states agree with the reference's fp32 CPU run to ~1e-5 and the END-TO-END segment tables can be
// compared bit for bit against the reference's goldens (in bf16 a frame within 4e-3 of a threshold may
// flip).  Correctness first: these kernels are simple, LDS-tiled, and not tuned.
#include "kernels.h"

// ---------------------------------------------------------------------------------------------------
// out[m][n] = act(sum_k X[m][k] W[n][k] + bias[n]) (+ res[m][n]);  64x64 tile, 4 waves (one 32x32 fragment
// each), K step 16.  A operand of 32x32x2: lane l holds A[i = l&31][k = l>>5]; B: B[k = l>>5][j = l&31].
// Orientation "lane = token": A = W rows (n), B = X rows (m) -> acc row = n_local, col = m_local.
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgsF32 a) {
    __shared__ float xs[64][17], ws[64][17];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (a.N + 63) / 64;
    const int m0 = (blockIdx.x / tiles_n) * 64, n0 = (blockIdx.x % tiles_n) * 64;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int lr = tid >> 2, lc = (tid & 3) * 4;       // 64 rows x 16 k: one float4 per thread per operand
    for (int k0 = 0; k0 < a.K; k0 += 16) {
        {
            int xm = m0 + lr; xm = xm < a.M ? xm : a.M - 1;
            int wr = n0 + lr; wr = wr < a.N ? wr : a.N - 1;
            const float4 xv = *(const float4*)(a.X + (size_t)xm * a.ldx + k0 + lc);
            const float4 wv = *(const float4*)(a.W + (size_t)wr * a.K + k0 + lc);
            xs[lr][lc] = xv.x; xs[lr][lc + 1] = xv.y; xs[lr][lc + 2] = xv.z; xs[lr][lc + 3] = xv.w;
            ws[lr][lc] = wv.x; ws[lr][lc + 1] = wv.y; ws[lr][lc + 2] = wv.z; ws[lr][lc + 3] = wv.w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const float wf = ws[wn * 32 + (lane & 31)][kk + (lane >> 5)];
            const float xf = xs[wm * 32 + (lane & 31)][kk + (lane >> 5)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf, xf, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int m = m0 + wm * 32 + (lane & 31);
    if (m >= a.M) return;
    const int h = lane >> 5;
    bool zero_row = false;
    int b = 0, t = 0;
    if (a.Tp > 0) {
        b = m / a.Tp; t = m - b * a.Tp;
        if (a.valid) { const int nv = a.valid[b] < a.T ? a.valid[b] : a.T; zero_row = t >= nv; }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 32 + 8 * g + 4 * h;
        if (n >= a.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = acc[4 * g + e] + (a.bias ? a.bias[n + e] : 0.f);
            if (a.act == 1) x = gelu_erf(x);
            else if (a.act == 2) x = fmaxf(x, 0.f);
            if (a.res) x += a.res[(size_t)m * a.ldres + n + e];
            if (zero_row) x = 0.f;
            v[e] = x;
        }
        *(float4*)(a.out0 + (size_t)m * a.ld0 + n) = make_float4(v[0], v[1], v[2], v[3]);
        if (a.xpad) *(float4*)(a.xpad + ((size_t)b * a.xpad_rows + 64 + t) * SYL_HIDDEN + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ---------------------------------------------------------------------------------------------------
// The same contraction for the forward of the parity mode (3.6 TFLOP per 32 x 10 s batch): 128x128 tile, 4 waves as 2x2
// with 2x2 32x32 fragments each, K step 32 floats (128-byte LDS rows -- byte for byte the geometry of the bf16 kernel's
// K step 64: same LDS-DMA staging with the XOR swizzle on the source address, same conflict-free ds_read_b128), 2-slot
// ring, two workgroups per CU.  One ds_read_b128 hands a lane FOUR k values of its row, and v_mfma_f32_32x32x2_f32
// takes k from the lane halves (lanes 0-31: first k, 32-63: second), so MFMA e of a group of four contracts the k pair
// {8j + e, 8j + 4 + e}: every output visits k in the fixed order 0 4 1 5 2 6 3 7 | 8 12 ... (exact fp32 FMAs, the same
// order for every tile shape and batch size).  At 64 cycles per MFMA the matrix pipe is the only bound: staging and
// fragment reads are ~5 % of its time, so a plain "wait, barrier, compute" loop suffices.
typedef __attribute__((address_space(3))) void* lds_vptr32;
typedef const __attribute__((address_space(1))) void* glb_vptr32;
__device__ __forceinline__ void glds16f(const void* g, void* l) { __builtin_amdgcn_global_load_lds((glb_vptr32)g, (lds_vptr32)l, 16, 0, 0); }

__global__ __launch_bounds__(256, 2) void gemm_f32_tiled_kernel(const GemmArgsF32 a) {
    constexpr int BM = 128, BN = 128, RB = 128;          // rows of 32 floats
    constexpr int XT = BM * RB, STAGE = (BM + BN) * RB;  // 32 KB per stage
    constexpr int NPW = (BM + BN) / 8 / 4;               // 1-KiB pieces (8 rows) per wave and stage: 8
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
    const int wg = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;
    const int srow = lane >> 3, spos = lane & 7;
    const float* gp[NPW];
    int lds_off[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int p = wave + 4 * i;
        const bool isx = p < BM / 8;
        const int r = (isx ? p : p - BM / 8) * 8 + srow;
        const int c = spos ^ ((r >> 1) & 7);
        if (isx) { int xm = m0 + r; xm = xm < a.M ? xm : a.M - 1; gp[i] = a.X + (size_t)xm * a.ldx + c * 4; }
        else { int wr = n0 + r; wr = wr < a.N ? wr : a.N - 1; gp[i] = a.W + (size_t)wr * a.K + c * 4; }
        lds_off[i] = (isx ? 0 : XT) + (isx ? p : p - BM / 8) * 1024;
    }
    auto stage = [&](int kt, int slot) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) glds16f(gp[i] + kt * 32, smem + slot * STAGE + lds_off[i]);
    };
    const int frow = lane & 31, fh = lane >> 5, swz = (lane >> 1) & 7;
    const int xrow_off = (wm * 64 + frow) * RB, wrow_off = XT + (wn * 64 + frow) * RB;
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nt = a.K / 32;
    stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my pieces of tile t have landed (LDS-DMA is not covered by the barrier)
        __syncthreads();                                      // ... everybody's; and slot (t+1)&1 is no longer being read
        if (t + 1 < nt) stage(t + 1, (t + 1) & 1);
        const char* sb = smem + (t & 1) * STAGE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ko = (((2 * kk + fh) ^ swz) << 4);
            f32x4_t xf[2], wf[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                xf[f] = *(const f32x4_t*)(sb + xrow_off + f * 32 * RB + ko);
                wf[f] = *(const f32x4_t*)(sb + wrow_off + f * 32 * RB + ko);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int fm = 0; fm < 2; ++fm)
#pragma unroll
                    for (int fn = 0; fn < 2; ++fn)
                        acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[fn][e], xf[fm][e], acc[fm][fn], 0, 0, 0);
        }
    }
    // ---- epilogue (lane = token row, registers = 16 output columns per fragment), as gemm_f32_kernel
    const int h = lane >> 5;
#pragma unroll
    for (int fm = 0; fm < 2; ++fm) {
        const int m = m0 + wm * 64 + fm * 32 + (lane & 31);
        if (m >= a.M) continue;
        bool zero_row = false;
        int b = 0, tt = 0;
        if (a.Tp > 0) {
            b = m / a.Tp; tt = m - b * a.Tp;
            if (a.valid) { const int nv = a.valid[b] < a.T ? a.valid[b] : a.T; zero_row = tt >= nv; }
        }
#pragma unroll
        for (int fn = 0; fn < 2; ++fn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + fn * 32 + 8 * g + 4 * h;
                if (n >= a.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[fm][fn][4 * g + e] + (a.bias ? a.bias[n + e] : 0.f);
                    if (a.act == 1) x = gelu_erf(x);
                    else if (a.act == 2) x = fmaxf(x, 0.f);
                    if (a.res) x += a.res[(size_t)m * a.ldres + n + e];
                    if (zero_row) x = 0.f;
                    v[e] = x;
                }
                *(float4*)(a.out0 + (size_t)m * a.ld0 + n) = make_float4(v[0], v[1], v[2], v[3]);
                if (a.xpad) *(float4*)(a.xpad + ((size_t)b * a.xpad_rows + 64 + tt) * SYL_HIDDEN + n) = make_float4(v[0], v[1], v[2], v[3]);
            }
    }
}

int launch_gemm_f32(const GemmArgsF32& a, hipStream_t s) {
    if (a.K % 16 || a.N % 4) { syl_set_error("launch_gemm_f32", "K % 16 and N % 4 must be 0"); return 1; }
    if (a.tiled && a.K % 32 == 0) {
        static PerDeviceOnce attr_once;
        if (attr_once.need()) HIP_TRY(hipFuncSetAttribute((const void*)gemm_f32_tiled_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 128));
        const int tiles = ((a.M + 127) / 128) * ((a.N + 127) / 128);
        hipLaunchKernelGGL(gemm_f32_tiled_kernel, dim3(tiles), dim3(256), 2 * 256 * 128, s, a);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const int tiles = ((a.M + 63) / 64) * ((a.N + 63) / 64);
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(tiles), dim3(256), 0, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// attention of the parity mode: the flash structure of attention.hip on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32),
// "query = lane" for both contractions:
//   S^T[key][q] = K . Q^T   A = K tile (LDS rows = keys, one ds_read_b128 = four d values -> four MFMAs), B = Q (registers)
//   O^T[d][q]   = V^T . P^T A = V tile (LDS [key][d], one ds_read_b32 per MFMA: lanes of a half read consecutive d), B = P,
//                           which is the S^T accumulator itself (fp32, never rounded, never moved between lanes)
// q, k, v are the three 768-wide column blocks of the fused projection buffer (row stride ld = 2304); 64-key tiles of K and
// V through a double-buffered LDS ring (LDS-DMA; K rows swizzled chunk ^= row & 15 for conflict-free ds_read_b128 on 256-byte
// rows).  Online softmax in fp32 with the lazy running maximum of the 16-bit kernel (p <= 2^8: exact in the fp32 sums).
// 232 GFLOP per 32 x 10 s forward: 30 ms on the one-wave-per-query VALU kernel this replaces, ~2-3 ms here.
#define AF_TILE (64 * 256)
#define AF_LDS (4 * AF_TILE)
__global__ __launch_bounds__(256, 2) void attention_f32_kernel(const float* __restrict__ qkv, int ld, const int* __restrict__ valid,
                                                               float* __restrict__ ctx, int T, int Tp) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ql = lane & 31, h = lane >> 5;
    const int nqb = (T + 127) / 128;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = vid / (nqb * SYL_HEADS), head = (vid / nqb) % SYL_HEADS;
    const int q0 = (vid % nqb) * 128 + wave * 32;
    int nvalid = valid ? valid[b] : T;
    nvalid = nvalid < T ? nvalid : T;
    const float* base = qkv + (size_t)b * Tp * ld + head * 64;
    // Q fragments (B operand), pre-scaled by 64^-0.5 (exact): lane (q, h) holds d = 8j + 4h .. + 3
    f32x4_t qf[8];
    {
        int qr = q0 + ql; qr = qr < Tp ? qr : Tp - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 v = *(const float4*)(base + (size_t)qr * ld + 8 * j + 4 * h);
            qf[j][0] = v.x * 0.125f; qf[j][1] = v.y * 0.125f; qf[j][2] = v.z * 0.125f; qf[j][3] = v.w * 0.125f;
        }
    }
    // staging: a 1-KiB piece = 4 rows x 256 B; wave w fills rows [16w, 16w + 16) of the K tile and of the V tile
    const int srow = lane >> 4, spos = lane & 15;
    const float* gk[4];
    const float* gv[4];
    int krow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wave * 16 + i * 4 + srow;
        krow[i] = r;
        gk[i] = base + SYL_HIDDEN + (spos ^ (r & 15)) * 4;      // + key row * ld per tile (clamped)
        gv[i] = base + 2 * SYL_HIDDEN + spos * 4;
    }
    const int lds_piece = wave * 16 * 256;
    auto stage = [&](int kv0, char* dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int kr = kv0 + krow[i]; kr = kr < Tp ? kr : Tp - 1;
            glds16f(gk[i] + (size_t)kr * ld, dst + lds_piece + i * 1024);
            glds16f(gv[i] + (size_t)kr * ld, dst + AF_TILE + lds_piece + i * 1024);
        }
    };
    f32x16_t oacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float LOG2E = 1.44269504088896341f;
    const int nt = (nvalid + 63) / 64;
    stage(0, smem);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) stage((t + 1) * 64, smem + ((t + 1) & 1) * 2 * AF_TILE);
        const char* kb = smem + (t & 1) * 2 * AF_TILE;
        const char* vb = kb + AF_TILE;
        const int kv0 = t * 64;
        const bool tail = kv0 + 64 > nvalid;
#pragma unroll 1
        for (int s2 = 0; s2 < 2; ++s2) {
            if (tail && kv0 + 32 * s2 >= nvalid) break;
            f32x16_t sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            const int krow_off = (s2 * 32 + ql) * 256;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4_t kf = *(const f32x4_t*)(kb + krow_off + (((2 * j + h) ^ (ql & 15)) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[j][e], sacc, 0, 0, 0);
            }
            if (tail) {
                asm volatile("; key-padding mask (last tile only)");
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + 32 * s2 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (key >= nvalid) sacc[r] = -INFINITY;
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const bool need = (mx - m_run) * LOG2E > 8.0f;
            if (__builtin_amdgcn_ballot_w64(need)) {
                const float m_new = need ? mx : m_run;
                const float alpha = need ? __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E) : 1.0f;
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
            }
            const float mb = m_run * LOG2E;
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], LOG2E, -mb)); psum += sacc[r]; }
            l_run += psum;
            // O^T += V^T . P^T: MFMA (g, e) contracts the key pair {8g + e (lanes 0-31), 8g + 4 + e (lanes 32-63)}
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = s2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float vf = *(const float*)(vb + key * 256 + (32 * db + ql) * 4);
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, sacc[r], oacc[db], 0, 0, 0);
                }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0 + ql;
    if (q < T) {
        float* dst = ctx + ((size_t)b * Tp + q) * SYL_HIDDEN + head * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(dst + 32 * db + 8 * g + 4 * h) = make_float4(oacc[db][4 * g + 0] * inv, oacc[db][4 * g + 1] * inv,
                                                                         oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv);
    }
}

int launch_attention_f32(const float* q, const float* k, const float* v, const int* valid, float* ctx, int B, int T, int Tp,
                         hipStream_t s) {
    // q, k, v are the three 768-wide column blocks of ONE fused projection buffer with row stride 2304
    if (k != q + SYL_HIDDEN || v != q + 2 * SYL_HIDDEN) { syl_set_error("launch_attention_f32", "expects a fused qkv buffer"); return 1; }
    static PerDeviceOnce once;
    if (once.need()) HIP_TRY(hipFuncSetAttribute((const void*)attention_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, AF_LDS));
    const int nqb = (T + 127) / 128;
    hipLaunchKernelGGL(attention_f32_kernel, dim3(nqb * SYL_HEADS * B), dim3(256), AF_LDS, s, q, 3 * SYL_HIDDEN, valid, ctx, T, Tp);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// pos-conv of the parity mode: the Toeplitz structure of posconv.hip on the exact-fp32 MFMA.
//   out[t][n] = sum_tap sum_c x[t + tap - 64][c] * w[n][c][tap]   (48 channels in / out per group, 128 taps)
// xpad: [B][Tp+128][768] f32; w: packed [16 g][128 tap][64 n (48 used)][52 c (48 used)] f32 = 208-byte rows (13 chunks of
// 16 B: an odd chunk stride keeps ds_read_b128 over 32 consecutive rows conflict-free).  One 255-frame window of the
// group's 48 channels sits in LDS (208-byte rows) and serves all 128 taps at shifted row offsets; the per-tap weight
// slabs (13 KiB) stream through a 2-slot LDS-DMA ring.  "lane = frame": A = weight rows (n), B = x rows (frames); one
// ds_read_b128 = four c values -> four v_mfma_f32_32x32x2_f32 (k pairs {8j + e, 8j + 4 + e}).  151 GFLOP per forward:
// 10.3 ms on the VALU kernel this replaces.
#define PF_ROW 208
#define PF_XWIN (256 * PF_ROW)        // 53248
#define PF_SLAB (64 * PF_ROW)         // 13312 = 13 pieces of 1 KiB
#define PF_LDS (PF_XWIN + 2 * PF_SLAB)
__global__ __launch_bounds__(256) void posconv_f32_kernel(const float* __restrict__ xpad, const float* __restrict__ wpk,
                                                          const float* __restrict__ bias, const float* __restrict__ x_f32,
                                                          float* __restrict__ out, int Tp) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    char* xwin = smem;
    char* wring = smem + PF_XWIN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ql = lane & 31, h = lane >> 5;
    const int t0 = blockIdx.x * 128, g = blockIdx.y, b = blockIdx.z;
    const int rows_per_b = Tp + 128;
    {
        const float* xb = xpad + (size_t)b * rows_per_b * SYL_HIDDEN + g * SYL_POSC;
        for (int idx = tid; idx < 255 * 12; idx += 256) {      // 16-byte chunks: 12 per row
            const int r = idx / 12, ch = idx - r * 12;
            int row = t0 + r; row = row < rows_per_b ? row : rows_per_b - 1;
            *(float4*)(xwin + r * PF_ROW + ch * 16) = *(const float4*)(xb + (size_t)row * SYL_HIDDEN + ch * 4);
        }
    }
    const char* wg_base = (const char*)wpk + (size_t)g * SYL_POSK * PF_SLAB;
    auto stage = [&](int tap, int buf) {
        const char* src = wg_base + (size_t)tap * PF_SLAB;
        char* dst = wring + buf * PF_SLAB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave + 4 * i;
            if (piece < 13) glds16f(src + piece * 1024 + lane * 16, dst + piece * 1024);
        }
    };
    f32x16_t acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int xfrag = (wave * 32 + ql) * PF_ROW + h * 16;
    const int wfrag = ql * PF_ROW + h * 16;
    stage(0, 0);
    for (int tap = 0; tap < SYL_POSK; ++tap) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // LDS-DMA completion is not covered by the barrier
        __syncthreads();                                      // slab `tap` visible (and, first time, the x window)
        if (tap + 1 < SYL_POSK) stage(tap + 1, (tap + 1) & 1);
        const char* wb = wring + (tap & 1) * PF_SLAB;
#pragma unroll
        for (int j = 0; j < 6; ++j) {                         // c = 8j + 4h .. + 3
            const f32x4_t xf = *(const f32x4_t*)(xwin + xfrag + tap * PF_ROW + j * 32);
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
                const f32x4_t wf = *(const f32x4_t*)(wb + nf * 32 * PF_ROW + wfrag + j * 32);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[e], xf[e], acc[nf], 0, 0, 0);
            }
        }
    }
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
                *(float4*)(out + m * SYL_HIDDEN + n) = make_float4(rr.x + gelu_erf(acc[nf][4 * gg + 0] + bb.x), rr.y + gelu_erf(acc[nf][4 * gg + 1] + bb.y),
                                                                    rr.z + gelu_erf(acc[nf][4 * gg + 2] + bb.z), rr.w + gelu_erf(acc[nf][4 * gg + 3] + bb.w));
            }
    }
}

int launch_posconv_f32(const float* xpad, const float* w, const float* bias, const float* x_f32, float* out, int B, int Tp,
                       hipStream_t s) {
    static PerDeviceOnce once;
    if (once.need()) HIP_TRY(hipFuncSetAttribute((const void*)posconv_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS));
    hipLaunchKernelGGL(posconv_f32_kernel, dim3((Tp + 127) / 128, SYL_POSG, B), dim3(256), PF_LDS, s, xpad, w, bias, x_f32, out, Tp);
    HIP_TRY(hipGetLastError());
    return 0;
}
