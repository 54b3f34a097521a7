// probe: what v_dot2c_f32_bf16 / v_dot2c_f32_f16 compute on gfx950 (build: hipcc --offload-arch=gfx950 -O2 -o dot2c_probe dot2c_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(const uint32_t* a, const uint32_t* b, float* o) {
    const int i = threadIdx.x;
    float s = 10.0f;
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s) : "v"(a[i]), "v"(b[i]));
    o[2 * i] = s;
    float t = 10.0f;
    t = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a[i + 4]), __builtin_bit_cast(h2, b[i + 4]), t, false);
    o[2 * i + 1] = t;
}
static uint32_t bf2(float lo, float hi) { uint32_t l, h; memcpy(&l, &lo, 4); memcpy(&h, &hi, 4); return (l >> 16) | (h & 0xffff0000u); }
static uint16_t f2h(float f) { _Float16 x = (_Float16)f; uint16_t r; memcpy(&r, &x, 2); return r; }
int main() {
    uint32_t ha[8], hb[8];
    float lo[4] = {1.5f, 3.0f, 256.0f, 0.0078125f}, hi[4] = {2.0f, -1.0f, 0.5f, 100.0f};
    for (int i = 0; i < 4; ++i) { ha[i] = bf2(lo[i], hi[i]); hb[i] = 0x3F803F80u; ha[4 + i] = f2h(lo[i]) | ((uint32_t)f2h(hi[i]) << 16); hb[4 + i] = 0x3C003C00u; }
    uint32_t *a, *b; float* o;
    hipMalloc(&a, 32); hipMalloc(&b, 32); hipMalloc(&o, 32);
    hipMemcpy(a, ha, 32, hipMemcpyHostToDevice); hipMemcpy(b, hb, 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, a, b, o);
    float ho[8]; hipMemcpy(ho, o, 32, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; ++i) printf("lo %g hi %g: 10 + lo + hi = %g   dot2c_bf16 -> %g   dot2c_f16 -> %g\n", lo[i], hi[i], 10 + lo[i] + hi[i], ho[2 * i], ho[2 * i + 1]);
    return 0;
}
