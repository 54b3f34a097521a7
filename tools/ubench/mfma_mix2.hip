// micro-benchmark (round 3): what does the operand movement of a 256x256x32 GEMM step cost beside the MFMAs, per staging
// primitive?  8 waves per CU (two per SIMD), non-zero bf16 operands, no barriers, no epilogue, L2-resident source; per
// 8 MFMAs a wave issues 6 ds_read_b128 fragment reads (conflict-free) plus, depending on MODE,
//   1: 2 global_load_lds_dwordx4 (LDS-DMA, 1 KiB each)                      = what csrc/gemm_bf16.hip does
//   2: 2 global_load_dwordx4 into registers, consumed 4 groups later by 2 ds_write_b128 (register staging, depth 4)
//   3: 2 global_load_dwordx4 only (no LDS write: isolates the load issue cost)
//   4: 2 ds_write_b128 only (isolates the LDS write cost)
//   5: 2 buffer_load_dwordx4 ... lds (LDS-DMA through a buffer descriptor: constant per-lane voffset, scalar soffset per step)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_mix2 tools/ubench/mfma_mix2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, const unsigned short* src, int iters, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) {
        unsigned x = (i * 2654435761u) ^ 0x9e3779b9u; x ^= x >> 13; x *= 0x85ebca6bu; x ^= x >> 16;
        ((unsigned*)smem)[i] = (x & 0x3fff3fffu) | 0x3c003c00u;
    }
    __syncthreads();
    // 64-byte rows as in the 8-wave kernel: row = lane & 31, chunk (0..3) ^= (row >> 2) & 3
    const char* base = smem + wave * 8192 + (lane & 31) * 64 + ((((lane >> 5)) ^ ((lane >> 2) & 3)) << 4);
    bf16x8_t a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8_t*)(base + i * 2048);
    for (int i = 0; i < 2; ++i) b[i] = *(const bf16x8_t*)(base + 4096 + i * 2048);
    f32x16_t c[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    const unsigned short* g = src + (size_t)((blockIdx.x & 7) * 8 + wave) * 16384 + lane * 8;
    char* dma = smem + 65536 + wave * 2048;
    uint4 st[4][2];
    for (int d = 0; d < 4; ++d) { st[d][0] = *(const uint4*)(g + d * 512); st[d][1] = *(const uint4*)(g + d * 512 + 8192); }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    const int voff = (int)(((size_t)((blockIdx.x & 7) * 8 + wave) * 16384 + lane * 8) * 2);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int o = u * 2048 * 0;
            FENCE();
            for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8_t*)(base + o + i * 2048);
            for (int i = 0; i < 2; ++i) b[i] = *(const bf16x8_t*)(base + o + 4096 + i * 2048);
            FENCE();
            for (int i = 0; i < 8; ++i) {
                c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i >> 1], b[i & 1], c[i], 0, 0, 0);
                if (i == 1 || i == 5) {
                    const int j = i == 5;
                    FENCE();
                    if (MODE == 1) __builtin_amdgcn_global_load_lds((glb_vptr)(g + ((it + u) & 15) * 512 + j * 8192), (lds_vptr)(dma + j * 1024), 16, 0, 0);
                    if (MODE == 5) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_vptr)(dma + j * 1024), 16, voff, (((it + u) & 15) * 512 + j * 8192) * 2, 0, 0);
                    if (MODE == 2 || MODE == 4) *(uint4*)(dma + j * 1024 + lane * 16) = st[u][j];          // the load issued 4 groups ago
                    if (MODE == 2 || MODE == 3) st[u][j] = *(const uint4*)(g + ((it + u) & 15) * 512 + j * 8192);
                    FENCE();
                }
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
    for (int d = 0; d < 4; ++d) s += (float)(st[d][0].x + st[d][1].y);
    out[blockIdx.x * 512 + threadIdx.x] = s + ((float*)smem)[threadIdx.x + 16384];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(float* out, unsigned short* src, unsigned long long* cyc, const char* name) {
    const int iters = 4000, blocks = 256;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 16384);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 65536 + 16384, 0, out, src, 400, cyc);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 65536 + 16384, 0, out, src, iters, cyc);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 8 * iters * 8 * 32768.0;
    printf("%-52s %8.1f us %6.0f TFLOP/s  clock %.2f GHz  %.1f cyc per 8-MFMA group per wave (2 waves/SIMD: floor 512)\n", name, ms * 1e3,
           flops / (ms * 1e-3) / 1e12, c / (ms * 1e-3) / 1e9, (double)c / iters);
}
int main() {
    float* out; unsigned long long* cyc; unsigned short* src;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    const size_t n = (size_t)256 * 8 * 65536 + 65536;
    hipMalloc(&src, n * 2);
    hipMemset(src, 0x3c, n * 2);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(out, src, cyc, "8 mfma + 6 ds_read_b128");
        run<1>(out, src, cyc, "  + 2 LDS-DMA (global_load_lds_dwordx4)");
        run<2>(out, src, cyc, "  + 2 global_load_dwordx4 + 2 ds_write_b128 (depth 4)");
        run<3>(out, src, cyc, "  + 2 global_load_dwordx4 only");
        run<4>(out, src, cyc, "  + 2 ds_write_b128 only");
        run<5>(out, src, cyc, "  + 2 LDS-DMA (buffer_load_dwordx4 lds, soffset)");
    }
    return 0;
}
