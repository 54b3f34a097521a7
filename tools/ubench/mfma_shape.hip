// micro-benchmark: register-only MFMA streams on random operands, v_mfma_f32_32x32x16_bf16 (this library's K loops) against v_mfma_f32_16x16x32_bf16 (the vendor
// library's: profiles/r06_yardstick_pmc.md shows its 4096^3 kernel at the same matrix-pipe utilisation and a 10 % higher clock).  Same FLOPs per launch; what
// differs is register traffic per FLOP: 32x32x16 reads half the operand bytes and moves twice the accumulator bytes (0.31 vs 0.25 B of register file per FLOP).
// On a power-capped part the sustained TFLOP/s of the two streams answers whether the instruction shape is a lever.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip && ./mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed, unsigned long long* cyc) {
    bf16x8_t a[4], b[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) { a[j][i] = (__bf16)(seed * ((threadIdx.x + 3 * j) % 7 + i) - 1.0f); b[j][i] = (__bf16)(seed * ((threadIdx.x + j) % 5 + 2 * i) - 0.5f); }
    float s = 0;
    unsigned long long t0 = __builtin_readcyclecounter(), t1;
    if constexpr (SHAPE == 32) {
        // 4 x 4 fragments of 32x32: 16 accumulators x 16 registers = 256 (a 128x128 wave tile), one k16 step per iteration: 16 MFMAs
        f32x16_t c[4][4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) c[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[n], c[m][n], 0, 0, 0);
        t1 = __builtin_readcyclecounter();
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) s += c[m][n][i];
    } else {
        // 8 x 8 fragments of 16x16: 64 accumulators x 4 registers = 256 (the same wave tile), one k32 step per TWO iterations' worth of FLOPs:
        // 64 MFMAs of 16 KFLOP = 2 x (16 MFMAs of 32 KFLOP); operands: 8 + 8 fragments (cycled over the 4 register sets)
        f32x4_t c[8][8] = {};
        for (int it = 0; it < iters; it += 2)
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 8; ++n) c[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m & 3], b[n & 3], c[m][n], 0, 0, 0);
        t1 = __builtin_readcyclecounter();
        for (int m = 0; m < 8; ++m) for (int n = 0; n < 8; ++n) for (int i = 0; i < 4; ++i) s += c[m][n][i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int SHAPE>
static void run(float* out, unsigned long long* cyc, float seed, const char* what) {
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, 400, seed, cyc);
        hipEventRecord(e0, 0);
        for (int l = 0; l < 5; ++l) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, iters, seed, cyc);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double flops = (double)blocks * 4 * iters * 16 * 32768.0;
        printf("%-22s %s data: %8.1f us  %6.0f TFLOP/s  wave-0 %.2f GHz\n", what, seed != 0.f ? "random" : "zero  ", ms * 1e3, flops / (ms * 1e-3) / 1e12, c / (ms * 1e-3) / 1e9);
    }
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 8);
    // (order matters on a chip that ramps its clock: each shape is measured twice, alternating, the 16x16 stream first)
    for (int pass = 0; pass < 2; ++pass) {
        run<16>(out, cyc, 0.37f, "v_mfma_16x16x32_bf16");
        run<32>(out, cyc, 0.37f, "v_mfma_32x32x16_bf16");
    }
    run<32>(out, cyc, 0.0f, "v_mfma_32x32x16_bf16");
    run<16>(out, cyc, 0.0f, "v_mfma_16x16x32_bf16");
    return 0;
}
