// Attention-kernel ablation bench (development aid; builds patched COPIES of csrc/attention.hip, see attn_ablate.sh).
// Usage: attn_ablate B T iters
#include ATTN_SRC
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
void syl_set_error(const char* what, const char* detail) { fprintf(stderr, "%s: %s\n", what, detail); }
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, T = argc > 2 ? atoi(argv[2]) : 499, iters = argc > 3 ? atoi(argv[3]) : 50;
    const int Tp = (T + 31) & ~31, Tpv = Tp;
    const size_t n = (size_t)B * 12 * Tp * 64;
    std::vector<unsigned short> h(n);
    srand(1);
    for (auto& x : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 2.f; unsigned u; memcpy((void*)&u, (const void*)&f, 4); x = (unsigned short)(u >> 16); }
    unsigned short *q, *k, *vt, *ctx; int* valid;
    hipMalloc(&q, n * 2); hipMalloc(&k, n * 2); hipMalloc(&vt, n * 2); hipMalloc(&ctx, (size_t)B * Tp * 768 * 2); hipMalloc(&valid, B * 4);
    hipMemcpy(q, h.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(k, h.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(vt, h.data(), n * 2, hipMemcpyHostToDevice);
    std::vector<int> v(B, T); hipMemcpy(valid, v.data(), B * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int qw = 1; qw <= 2; ++qw) {
        for (int i = 0; i < 5; ++i) launch_attention(q, k, vt, valid, ctx, B, T, Tp, Tpv, qw, 0, 0);
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) launch_attention(q, k, vt, valid, ctx, B, T, Tp, Tpv, qw, 0, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 4.0 * T * (double)T * 64 * 12 * B;
        printf("  qw=%d %.1f us %.0f TF", qw, ms / iters * 1e3, fl / (ms / iters * 1e-3) / 1e12);
    }
    printf("\n");
    return 0;
}
