// Achievable HBM bandwidth on MI355X for the three access mixes the Segmenter path has: pure streaming WRITE (conv0 +
// GroupNorm + GELU writes 1.07 GB and reads 20 MB), pure READ, and COPY (LayerNorm: reads fp32, writes bf16).
// 16-byte accesses per lane, grid-stride, buffers far beyond the 256 MB Infinity Cache.
// build+run: hipcc --offload-arch=gfx950 -O3 tools/ubench/hbm_bw.hip -o /tmp/hbm_bw && /tmp/hbm_bw
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_write(uint4* __restrict__ dst, size_t n, unsigned v) {
    const uint4 x = make_uint4(v, v + 1, v + 2, v + 3);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = x;
}
__global__ void k_write_nt(uint4* __restrict__ dst, size_t n, unsigned v) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 x = {v, v + 1, v + 2, v + 3};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(x, (u32x4*)dst + i);
}
__global__ void k_read(const uint4* __restrict__ src, size_t n, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 x = src[i];
        acc ^= x.x ^ x.y ^ x.z ^ x.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
    const size_t bytes = (size_t)2 << 30;            // 2 GiB per buffer
    const size_t n = bytes / 16;
    uint4 *a, *b; unsigned* out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 64);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // few workgroups = few CUs: is a CU's own store path the limit (rate per CU stays), or the chip's (rate per CU rises)?
    for (int blocks : {32, 64, 128, 256, 512, 1024, 2048, 4096, 16384}) {
        for (int mode = 0; mode < 4; ++mode) {
            if (blocks < 1024 && mode == 1) continue;
            auto run = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, a, n, 7u);
                if (mode == 1) hipLaunchKernelGGL(k_write_nt, dim3(blocks), dim3(256), 0, 0, a, n, 7u);
                if (mode == 2) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n, out);
                if (mode == 3) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n);
            };
            run(); hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            for (int i = 0; i < 5; ++i) run();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            const double moved = mode == 3 ? 2.0 * bytes : (double)bytes;
            const char* nm[4] = {"write", "write(nt)", "read", "copy(r+w)"};
            printf("blocks %5d  %-10s %7.3f ms  %6.2f TB/s  (%5.1f GB/s per workgroup)\n", blocks, nm[mode], ms, moved / (ms * 1e-3) / 1e12,
                   moved / (ms * 1e-3) / 1e9 / blocks);
        }
    }
    return 0;
}
