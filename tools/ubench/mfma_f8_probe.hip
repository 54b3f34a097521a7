// Layout probe for v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3, E8M0 block scales) on gfx950:
// which K elements does lane l hold, and which byte of the scale VGPR does OPSEL pick?
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_f8_probe.hip -o /tmp/f8probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int OPA, int OPB>
__global__ void probe(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* C, int variant) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    union { v8i v; uint8_t b[32]; } a, b;
    for (int t = 0; t < 32; ++t) {
        int k = variant == 0 ? 32 * h + t : 16 * h + (t & 15) + 32 * (t >> 4);
        a.b[t] = A[r * 64 + k];
        b.b[t] = B[r * 64 + k];
    }
    // scale word: byte q holds the scale of k-block ((q + h) & 1) so that every byte position is meaningful
    uint32_t sa = 0, sb = 0;
    for (int q = 0; q < 4; ++q) {
        sa |= (uint32_t)SA[r * 2 + ((q + h) & 1)] << (8 * q);
        sb |= (uint32_t)SB[r * 2 + ((q + h) & 1)] << (8 * q);
    }
    v16f c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a.v, b.v, c, 0, 0, OPA, (int)sa, OPB, (int)sb);
    for (int i = 0; i < 16; ++i) C[l * 16 + i] = c[i];
}

static float e4m3(uint8_t v) {
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -x : x;
}

int main() {
    const uint8_t tab[6] = {0x00, 0x38, 0xB8, 0x40, 0x30, 0x44};   // 0, 1, -1, 2, 0.5, 3
    uint8_t A[32 * 64], B[32 * 64], SA[64], SB[64];
    for (int i = 0; i < 32; ++i)
        for (int k = 0; k < 64; ++k) { A[i * 64 + k] = tab[(i * 7 + k * 3 + (k >> 4)) % 6]; B[i * 64 + k] = tab[(i * 5 + k + (k >> 5) * 2) % 6]; }
    for (int i = 0; i < 32; ++i)
        for (int kb = 0; kb < 2; ++kb) { SA[i * 2 + kb] = 127 + ((i + kb) % 3); SB[i * 2 + kb] = 127 - ((i + 2 * kb) % 2) - kb; }
    uint8_t *dA, *dB, *dSA, *dSB; float* dC;
    hipMalloc(&dA, sizeof A); hipMalloc(&dB, sizeof B); hipMalloc(&dSA, 64); hipMalloc(&dSB, 64); hipMalloc(&dC, 64 * 16 * 4);
    hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
    hipMemcpy(dSA, SA, 64, hipMemcpyHostToDevice); hipMemcpy(dSB, SB, 64, hipMemcpyHostToDevice);
    float C[64 * 16];
    for (int variant = 0; variant < 2; ++variant)
        for (int op = 0; op < 4; ++op) {
            if (op == 0) hipLaunchKernelGGL((probe<0, 0>), 1, 64, 0, 0, dA, dB, dSA, dSB, dC, variant);
            if (op == 1) hipLaunchKernelGGL((probe<1, 1>), 1, 64, 0, 0, dA, dB, dSA, dSB, dC, variant);
            if (op == 2) hipLaunchKernelGGL((probe<2, 2>), 1, 64, 0, 0, dA, dB, dSA, dSB, dC, variant);
            if (op == 3) hipLaunchKernelGGL((probe<3, 1>), 1, 64, 0, 0, dA, dB, dSA, dSB, dC, variant);
            hipMemcpy(C, dC, sizeof C, hipMemcpyDeviceToHost);
            // hypotheses: C[i][j] = sum_k A[i][k] sa[i][kb(k)] B[j][k] sb[j][kb(k)], A rows -> C rows (r&3)+8(r>>2)+4h, B rows -> col lane&31
            // scale byte picked for lane half h: byte index q -> k-block ((q + h) & 1); test q in 0..3 for A and B
            for (int qa = 0; qa < 4; ++qa)
                for (int qb = 0; qb < 4; ++qb)
                    for (int swapab = 0; swapab < 2; ++swapab) {
                        double err = 0;
                        for (int l = 0; l < 64; ++l)
                            for (int rr = 0; rr < 16; ++rr) {
                                int row = (rr & 3) + 8 * (rr >> 2) + 4 * (l >> 5), col = l & 31;
                                int i = swapab ? col : row, j = swapab ? row : col;
                                double s = 0;
                                for (int k = 0; k < 64; ++k) {
                                    int kb = k >> 5;
                                    // lane half holding k-block kb is h = kb (variant 0); it used byte q -> block ((q + h) & 1)
                                    int kba = (qa + kb) & 1, kbb = (qb + kb) & 1;
                                    s += (double)e4m3(A[i * 64 + k]) * ldexp(1.0, SA[i * 2 + kba] - 127) * e4m3(B[j * 64 + k]) * ldexp(1.0, SB[j * 2 + kbb] - 127);
                                }
                                err += fabs(s - C[l * 16 + rr]);
                            }
                        if (err < 1e-3) printf("MATCH variant=%d opsel(a,b)=(%d,%d) -> scale byte qa=%d qb=%d swapab=%d\n", variant, op == 3 ? 3 : op, op == 3 ? 1 : op, qa, qb, swapab);
                    }
        }
    printf("probe done; C[0..3]=%g %g %g %g\n", C[0], C[1], C[2], C[3]);
    return 0;
}
