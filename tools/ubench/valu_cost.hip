// micro-benchmark: VALU-port cost of single instructions beside MFMAs: 12 fillers of ONE kind per v_mfma_f32_32x32x16_bf16, three waves per
// SIMD (the regime where the issue port, not the matrix pipe, sets the time): cycles per filler = (t - t_mfma_issue) / 12.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_cost tools/ubench/valu_cost.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
#define NV 12
template <int KIND>
__global__ __launch_bounds__(768) void k(float* out, int iters) {
    f32x16_t acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)0.f; b[i] = (__bf16)0.f; }
    double pk[6] = {1.0 * threadIdx.x, 2.0, 3.0, 4.0, 5.0, 6.0};
    float v[NV], x = (float)threadIdx.x * 1e-9f, y = x + 1e-9f, z = y + 1e-9f;
    for (int i = 0; i < NV; ++i) v[i] = (float)threadIdx.x * 1e-9f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[j]));
                if (KIND == 1) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[j]) : "v"(x), "v"(y), "v"(z));
                if (KIND == 2) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(v[j]) : "v"(x), "v"(y), "v"(z));
                if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[j]) : "v"(x), "v"(y));
                if (KIND == 4) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[j]) : "v"(x), "v"(y));
                if (KIND == 5) asm volatile("v_exp_f32 %0, %1" : "=v"(v[j]) : "v"(x));
                if (KIND == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(v[j]) : "v"(x));
                if (KIND == 7) asm volatile("v_fmamk_f32 %0, %1, 0x3fb8aa3b, %2" : "=v"(v[j]) : "v"(x), "v"(y));
                if (KIND == 8) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[j]) : "v"(x), "v"(y));
                if (KIND == 9) asm volatile("v_rcp_f32 %0, %1" : "=v"(v[j]) : "v"(x));
                if (KIND == 10) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v[j]) : "v"(x), "v"(y));
                if (KIND == 11) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(v[j]) : "v"(x), "v"(y));
                if (KIND == 12) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pk[j % 4]) : "v"(pk[4]), "v"(pk[5]));
                if (KIND == 13) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x), "v"(y) : "vcc");
            }
        }
    }
    float s = x + y + z + (float)(pk[0] + pk[1] + pk[2] + pk[3]);
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < NV; ++i) s += v[i];
    out[blockIdx.x * 768 + threadIdx.x] = s;
}
template <int KIND> float run(float* out) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(768), 0, 0, out, 100);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(768), 0, 0, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / (iters * 8.0f * 3);
}
int main() {
    float* out; hipMalloc(&out, 256 * 768 * 4);
    const char* names[14] = {"v_fma_f32 (one register)", "v_fma_f32 (three registers)", "v_max3_f32", "v_cvt_pk_bf16_f32", "v_add_f32", "v_exp_f32", "v_mov_b32", "v_fmamk_f32", "v_mul_f32", "v_rcp_f32", "v_dot2c_f32_bf16", "v_dot2_f32_bf16", "v_pk_add_f32", "v_cmp_lt_f32"};
    float t[14] = {run<0>(out), run<1>(out), run<2>(out), run<3>(out), run<4>(out), run<5>(out), run<6>(out), run<7>(out), run<8>(out), run<9>(out), run<10>(out), run<11>(out), run<12>(out), run<13>(out)};
    for (int i = 0; i < 14; ++i) printf("%-28s %6.2f ns per MFMA gap of 12 -> %.2f ns = %.1f cycles at 2.4 GHz per filler\n", names[i], t[i], (t[i] - 1.7f) / 12, (t[i] - 1.7f) / 12 * 2.4f);
    return 0;
}
