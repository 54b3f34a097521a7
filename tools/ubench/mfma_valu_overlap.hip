// micro-benchmark: how many plain VALU instructions hide behind one v_mfma_f32_32x32x16_bf16 on one SIMD, by waves per SIMD
// (1, 2, 3) and fillers per MFMA gap (0 .. 12), hand-placed (inline asm) so that the order is the one written here.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_valu_overlap tools/ubench/mfma_valu_overlap.hip && tools/ubench/mfma_valu_overlap
// Operands are zero (maximum clock); time is reported per MFMA in ns and, against the NV = 0 row of the same wave count, as extra ns
// per MFMA -- 32 shader cycles are 13.3 ns at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <int NV, bool TRANS>
__global__ __launch_bounds__(768) void k(float* out, int iters) {
    f32x16_t acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)0.f; b[i] = (__bf16)0.f; }
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = (float)threadIdx.x * 1e-9f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (TRANS && (j & 1)) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
                else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[j]));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * 768 + threadIdx.x] = s;
}

template <int NV, bool TRANS>
float run(int threads, float* out) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, TRANS>), dim3(256), dim3(threads), 0, 0, out, 100);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NV, TRANS>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / (iters * 8.0f * (threads / 256));      // ns per MFMA per SIMD-resident wave ... normalised below
}

int main() {
    float* out; hipMalloc(&out, 256 * 768 * 4);
    printf("ns per MFMA issued by ONE SIMD (all its waves together); 32 cycles = 13.3 ns at 2.4 GHz\n");
    for (int waves = 1; waves <= 3; ++waves) {
        const int th = 256 * waves;
        float t[7] = {run<0, false>(th, out), run<2, false>(th, out), run<4, false>(th, out), run<6, false>(th, out), run<8, false>(th, out),
                      run<10, false>(th, out), run<12, false>(th, out)};
        float u[3] = {run<4, true>(th, out), run<8, true>(th, out), run<12, true>(th, out)};
        printf("%d wave(s)/SIMD  fillers 0/2/4/6/8/10/12: ", waves);
        for (int i = 0; i < 7; ++i) printf("%6.2f ", t[i]);
        printf(" | half of them v_exp, 4/8/12: %6.2f %6.2f %6.2f\n", u[0], u[1], u[2]);
    }
    return 0;
}
