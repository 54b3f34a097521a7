// segment_kernel phase timing (development aid; builds patched COPIES of csrc/segment.hip, see seg_phase.sh).
// Usage: seg_phase hidden.bin B T   (hidden.bin = raw f32 [B][T][768], e.g. written by seg_phase.sh from a real forward)
#include SEG_SRC
#include <cstdio>
#include <cstdlib>
#include <vector>
void syl_set_error(const char* what, const char* detail) { fprintf(stderr, "%s: %s\n", what, detail); }
int main(int argc, char** argv) {
    const int B = atoi(argv[2]), T = atoi(argv[3]);
    const size_t n = (size_t)B * T * 768;
    std::vector<float> h(n);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(h.data(), 4, n, f) != n) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    fclose(f);
    float *hid, *feat; int64_t* seg; int* nseg;
    hipMalloc(&hid, n * 4); hipMalloc(&feat, n * 4); hipMalloc(&seg, (size_t)B * T * 16); hipMalloc(&nseg, B * 4);
    hipMemcpy(hid, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_segment(hid, B, T, 768, 2.6f, 0.8f, seg, nseg, feat, nullptr, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) launch_segment(hid, B, T, 768, 2.6f, 0.8f, seg, nseg, feat, nullptr, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<int> ns(B); hipMemcpy(ns.data(), nseg, B * 4, hipMemcpyDeviceToHost);
    long tot = 0; for (int v : ns) tot += v;
    printf("%.1f us per launch (segments found: %ld)\n", ms / 20 * 1e3, tot);
    return 0;
}
