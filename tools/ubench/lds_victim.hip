// Victim for the co-residency experiment: every workgroup fills a small static LDS array with a pattern, then re-reads
// it `iters` times (broadcast reads, like conv0_gn_gelu's tap reads) and counts mismatches; registers hold a second copy
// of a pattern that is re-checked too.  Run it on one stream while libsylber_hip's 128x128 GEMM runs on another.
// Built into a shared object and driven from python (tools/_race6.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(256) void lds_victim(unsigned* __restrict__ bad, int iters, unsigned seed) {
    __shared__ unsigned xs[1296];
    for (int i = threadIdx.x; i < 1296; i += 256) xs[i] = seed * 2654435761u + i * 40503u + blockIdx.x;
    unsigned regs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) regs[j] = seed + threadIdx.x * 131u + j * 7919u;
    __syncthreads();
    unsigned nbad = 0;
    const int wave = threadIdx.x >> 6;
    for (int it = 0; it < iters; ++it) {
        for (int r = 0; r < 64; ++r) {
            const unsigned* xr = xs + 5 * (wave * 64 + r);
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const unsigned v = xr[j];
                const int i = 5 * (wave * 64 + r) + j;
                nbad += (v != seed * 2654435761u + i * 40503u + blockIdx.x);
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) { nbad += (regs[j] != seed + threadIdx.x * 131u + j * 7919u); asm volatile("" : "+v"(regs[j])); }
    }
    if (nbad) atomicAdd(bad, nbad);
}
extern "C" int launch_lds_victim(unsigned* bad, int blocks, int iters, unsigned seed, void* stream) {
    hipLaunchKernelGGL(lds_victim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, bad, iters, seed);
    return (int)hipGetLastError();
}
