/* Test helper (CPU): compares sylber_amd/csrc/powf_half.h with the host libm powf(x, .5f).
 * usage: powf_replica_check <stride>   (stride 1 = every positive normal float32) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "../sylber_amd/csrc/powf_half.h"

int main(int argc, char** argv) {
    unsigned long stride = argc > 1 ? strtoul(argv[1], 0, 10) : 997;
    unsigned long bad = 0, n = 0;
    #pragma omp parallel for reduction(+:bad,n) schedule(static)
    for (unsigned long u = 0x00800000ul; u < 0x7f800000ul; u += stride) {
        float x = ph_u2f((uint32_t)u);
        float a = powf(x, 0.5f), b = powf_half_glibc(x);
        if (ph_f2u(a) != ph_f2u(b)) bad++;
        n++;
    }
    /* specials */
    float sp[] = {0.0f, 1e-45f, 1e-40f, 1.1754942e-38f, INFINITY};
    for (unsigned i = 0; i < sizeof(sp) / sizeof(sp[0]); i++)
        if (ph_f2u(powf(sp[i], 0.5f)) != ph_f2u(powf_half_glibc(sp[i]))) bad++;
    printf("%lu %lu\n", n, bad);
    return bad != 0;
}
