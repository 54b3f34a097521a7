// What does the shape of a GEMM tile's epilogue stores cost?  Every workgroup (512 threads) writes one 256 x 256 bf16
// tile (128 KB) of a row-major [M][N] matrix, nothing else:
//   pattern 0 "per-wave blocks": wave (wm, wn) of a 2 x 4 layout owns rows wm*128.., cols wn*64..; one store
//             instruction = 8 rows x 128 B (what the per-wave LDS-staged epilogue of gemm8_bf16_kernel issues)
//   pattern 1 "tile rows": wave w owns tile rows 32w..32w+31; one store instruction = 2 rows x 512 B
//   pattern 2 "tile rows, nt": the same with non-temporal stores
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/store_pattern.hip -o tools/ubench/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <int PATTERN>
__global__ __launch_bounds__(512) void tile_store(unsigned short* __restrict__ out, int M, int N) {
    const int tiles_n = N / 256;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (wg / tiles_n) * 256, n0 = (wg % tiles_n) * 256;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32x4 v = {(unsigned)wg, (unsigned)lane, 3u, 4u};
    if (PATTERN == 0) {
        const int wm = wave >> 2, wn = wave & 3;
#pragma unroll
        for (int fm = 0; fm < 4; ++fm)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = it * 64 + lane;
                const int r = idx >> 3, c = idx & 7;                     // 8 chunks of 16 B per 128-B row piece
                u32x4* dst = (u32x4*)(out + (size_t)(m0 + wm * 128 + fm * 32 + r) * N + n0 + wn * 64 + c * 8);
                *dst = v;
            }
    } else {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = wave * 32 + it * 2 + (lane >> 5), c = lane & 31;   // 32 chunks of 16 B per 512-B tile row
            u32x4* dst = (u32x4*)(out + (size_t)(m0 + r) * N + n0 + c * 8);
            if (PATTERN == 2) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
    }
}

int main() {
    unsigned short* out;
    const size_t cap = (size_t)524288 * 512 * 2;
    hipMalloc(&out, cap);
    hipMemset(out, 0, cap);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int shapes[3][2] = {{524288, 512}, {16384, 3072}, {131072, 512}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1];
        const int tiles = (M / 256) * (N / 256);
        for (int p = 0; p < 3; ++p) {
            auto run = [&]() {
                if (p == 0) hipLaunchKernelGGL(tile_store<0>, dim3(tiles), dim3(512), 0, 0, out, M, N);
                if (p == 1) hipLaunchKernelGGL(tile_store<1>, dim3(tiles), dim3(512), 0, 0, out, M, N);
                if (p == 2) hipLaunchKernelGGL(tile_store<2>, dim3(tiles), dim3(512), 0, 0, out, M, N);
            };
            run(); hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            for (int i = 0; i < 5; ++i) run();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            printf("[%6d x %4d] %5d tiles  pattern %d  %7.3f ms  %6.2f TB/s\n", M, N, tiles, p, ms, (double)M * N * 2 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
