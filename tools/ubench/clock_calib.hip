// Which clock does s_memtime (__builtin_readcyclecounter / clock64) count on gfx950?  One wave runs N dependent v_fma_f32
// (4-5 shader cycles each when dependent); compare the tick delta with the wall time (hipEvents) at light and at heavy load.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/clock_calib.hip -o tools/ubench/clock_calib
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef short v8s __attribute__((ext_vector_type(8)));

__global__ void chain(float* out, unsigned long long* ticks, int n, int heavy) {
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    v16f acc = {0}; v8s a = {1, 2, 3, 4, 5, 6, 7, 8};
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long w0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        if (heavy) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 64; ++j) x = fmaf(x, y, 1e-7f);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0) { ticks[blockIdx.x * 2] = t1 - t0; ticks[blockIdx.x * 2 + 1] = w1 - w0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + acc[0];
}

int main() {
    float* out; unsigned long long* tk;
    hipMalloc(&out, 4 * 1024 * 1024); hipMalloc(&tk, 16 * 4096);
    for (int heavy = 0; heavy < 2; ++heavy)
        for (int blocks : {1, 1024}) {
            const int threads = heavy ? 256 : 64, n = heavy ? 20000 : 20000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(threads), 0, 0, out, tk, n, heavy);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(threads), 0, 0, out, tk, n, heavy);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, tk, 16, hipMemcpyDeviceToHost);
            const double ops = heavy ? 8.0 * n : 64.0 * n;
            printf("%s blocks=%4d: s_memtime ticks %llu (%.2f per %s), wall_clock64 ticks %llu, event time %.3f ms -> s_memtime rate %.3f GHz, wall_clock64 rate %.1f MHz\n",
                   heavy ? "MFMA " : "v_fma", blocks, h[0], h[0] / ops, heavy ? "mfma" : "fma", h[1], ms, h[0] / (ms * 1e6), h[1] / (ms * 1e3));
        }
    return 0;
}
