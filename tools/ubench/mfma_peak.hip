// micro-benchmark: pure v_mfma_f32_32x32x16_bf16 throughput (no memory traffic) -> sustained clock under MFMA load
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed, unsigned long long* cyc) {
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * (threadIdx.x % 7 + i) - 1.0f); b[i] = (__bf16)(seed * (threadIdx.x % 5 + 2 * i) - 0.5f); }
    f32x16_t c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 8);
    for (int pass = 0; pass < 2; ++pass) {
        const float seed = pass ? 0.37f : 0.0f;
        for (int blocks : {256, 512}) {
            const int iters = 20000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1000, seed, cyc);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, seed, cyc);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double flops = (double)blocks * 4 * iters * 4 * 32768.0;
            printf("%s data, %d blocks: %.1f us, %.0f TFLOP/s, wave-0 cycles %llu -> %.2f GHz, %.1f cyc/MFMA\n", pass ? "random" : "zero",
                   blocks, ms * 1e3, flops / (ms * 1e-3) / 1e12, c, c / (ms * 1e-3) / 1e9, (double)c / (iters * 4.0));
        }
    }
    return 0;
}
