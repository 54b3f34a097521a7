// micro-benchmark: MFMA throughput / sustained clock with LDS fragment reads and LDS-DMA mixed in at GEMM-like ratios
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;

// MODE bit0: operands re-read from LDS (6 ds_read_b128 per 8 MFMA); bit1: 2 LDS-DMA (1 KiB each) per 8 MFMA
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, const unsigned short* src, int iters, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // fill LDS with pseudo-random bf16
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) {
        unsigned x = (i * 2654435761u) ^ 0x9e3779b9u; x ^= x >> 13; x *= 0x85ebca6bu; x ^= x >> 16;
        ((unsigned*)smem)[i] = (x & 0x3fff3fffu) | 0x3c003c00u;   // bf16 values in [0.0078, 2)
    }
    __syncthreads();
    // conflict-free fragment addressing as in the GEMM: row = lane&31 (128-B rows), chunk ^= (row>>1)&7
    const char* base = smem + wave * 4096 + (lane & 31) * 128 + ((((lane >> 5)) ^ ((lane >> 1) & 7)) << 4);
    bf16x8_t a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8_t*)(base + i * 32);
    for (int i = 0; i < 2; ++i) b[i] = *(const bf16x8_t*)(base + 64 + i * 32);
    f32x16_t c[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    // L2-resident source: all workgroups sweep the same 2 MiB (like the shared W panel / neighbouring X tiles of a GEMM)
    const unsigned short* g = src + (size_t)((blockIdx.x & 7) * 8 + wave) * 16384 + lane * 8;
    char* dma = smem + 65536 + wave * 2048;
    uint4 st0 = *(const uint4*)g, st1 = *(const uint4*)(g + 8192);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) {
            const int o = (it & 3) * 8192;
            for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8_t*)(base + o + (i & 1) * 32 * 128 * 0 + ((i * 32) ^ 0));
            for (int i = 0; i < 2; ++i) b[i] = *(const bf16x8_t*)(base + o + ((64 + i * 32) ^ 0));
        }
        if (MODE & 2) {
            __builtin_amdgcn_global_load_lds((glb_vptr)(g + (it & 15) * 512), (lds_vptr)dma, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_vptr)(g + (it & 15) * 512 + 8192), (lds_vptr)(dma + 1024), 16, 0, 0);
        }
        if (MODE & 4) {   // register staging: the loads issued last iteration are written to LDS now, new loads issued
            *(uint4*)(dma + lane * 16) = st0;
            *(uint4*)(dma + 1024 + lane * 16) = st1;
            st0 = *(const uint4*)(g + (it & 15) * 512);
            st1 = *(const uint4*)(g + (it & 15) * 512 + 8192);
        }
        __builtin_amdgcn_sched_barrier(0);
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i >> 1], b[i & 1], c[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s + (float)st0.x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(float* out, unsigned short* src, unsigned long long* cyc, const char* name) {
    const int iters = 4000, blocks = 256;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 16384);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 65536 + 16384, 0, out, src, 500, cyc);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 65536 + 16384, 0, out, src, iters, cyc);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 8 * iters * 8 * 32768.0;
    printf("%-28s %8.1f us %6.0f TFLOP/s  clock %.2f GHz  %.1f cyc per 8-MFMA group (ideal 256 x2 waves = 512)\n", name, ms * 1e3,
           flops / (ms * 1e-3) / 1e12, c / (ms * 1e-3) / 1e9, (double)c / iters);
}
int main() {
    float* out; unsigned long long* cyc; unsigned short* src;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    const size_t n = (size_t)256 * 8 * 65536 + 65536;
    hipMalloc(&src, n * 2);
    hipMemset(src, 0x3c, n * 2);
    run<0>(out, src, cyc, "mfma only (8 waves/CU)");
    run<1>(out, src, cyc, "+ 6 ds_read_b128 / 8 mfma");
    run<2>(out, src, cyc, "+ 2 LDS-DMA / 8 mfma");
    run<3>(out, src, cyc, "+ both");
    run<4>(out, src, cyc, "+ 2 (global_load + ds_write)");
    run<5>(out, src, cyc, "+ reads + reg-staged loads");
    return 0;
}
