/*
 * ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * Plain-C restatement of the reference's syllable-boundary detector and segment mean-pool:
 *   get_segment  sylber/utils/segment_utils.py:72-131
 *   cossim       sylber/utils/segment_utils.py:68-69
 *   mean-pool    sylber/model/sylber.py:133   (states[s:e].mean(0))
 * The reference runs these in numpy float32; segment indices are decided by discontinuous
 * float comparisons, so this file reproduces numpy 2.x's float32 evaluation order exactly:
 *   - ndarray.sum over a contiguous axis = 0 + pairwise sum (8 strided accumulators per <=128
 *     element block, recursive halving above) -- numpy/core/src/umath/loops_utils.h.src;
 *   - ndarray.mean(0) over rows = 0 + row0 + row1 + ... (sequential), then / float32(n);
 *   - array ** .5 = correctly rounded sqrtf (fast_scalar_power); numpy-scalar ** .5 = libm powf
 *     (cossim of two 1-D vectors goes through the scalar path, segment_utils.py:97,114);
 *   - thresholds compared in float32 (NEP 50 weak python scalars).
 * Pinned against the imported reference by tools/gen_golden.py and tests/test_oracle_segment.py
 * (bit-exact on every golden case in tests/golden/segment_cases.npz).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* numpy pairwise sum of n contiguous float32 (loops_utils.h.src, FLOAT_pairwise_sum) */
static float pairwise_sum(const float *a, long n)
{
    if (n < 8) {
        float res = 0.f;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        long i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
    }
}

/* ndarray.sum(-1) of a contiguous float32 vector: identity 0 + pairwise */
static float np_sum(const float *a, long n) { return 0.f + pairwise_sum(a, n); }

static float np_dot(const float *x, const float *y, long d, float *tmp)
{
    for (long i = 0; i < d; i++) tmp[i] = x[i] * y[i];
    return np_sum(tmp, d);
}

/* cossim on two 1-D vectors: numpy-scalar path, (s + 1e-8) ** .5 -> powf */
static float cossim_scalar(const float *x, const float *y, long d, float *tmp)
{
    float dot = np_dot(x, y, d, tmp);
    float nx = powf(np_dot(x, x, d, tmp) + 1e-8f, 0.5f);
    float ny = powf(np_dot(y, y, d, tmp) + 1e-8f, 0.5f);
    return dot / nx / ny;
}

/* cossim row of a 2-D array against a [1,d] centre: array path, sqrtf */
static float cossim_array(const float *x, const float *y, long d, float *tmp)
{
    float dot = np_dot(x, y, d, tmp);
    float nx = sqrtf(np_dot(x, x, d, tmp) + 1e-8f);
    float ny = sqrtf(np_dot(y, y, d, tmp) + 1e-8f);
    return dot / nx / ny;
}

/* states[s:e].mean(0): sequential row accumulation from the additive identity, then / n */
static void mean_rows(const float *states, long d, long s, long e, float *out)
{
    for (long j = 0; j < d; j++) out[j] = 0.f;
    for (long r = s; r < e; r++)
        for (long j = 0; j < d; j++) out[j] += states[r * d + j];
    float n = (float)(e - s);
    for (long j = 0; j < d; j++) out[j] = out[j] / n;
}

/*
 * segments_out: capacity T rows of [start,end); returns the number of segments.
 * norms_in may be NULL (computed as sqrt(sum(states**2)+1e-8), segment_utils.py:74-75).
 */
long sylber_oracle_get_segment(const float *states, long T, long d, float norm_thr, float merge_thr,
                               const float *norms_in, int64_t *segments_out)
{
    float *tmp = (float *)malloc(sizeof(float) * (size_t)(d > T ? d : T + 1));
    float *curr = (float *)malloc(sizeof(float) * (size_t)d);
    float *ca = (float *)malloc(sizeof(float) * (size_t)d);
    float *cb = (float *)malloc(sizeof(float) * (size_t)d);
    long *seg = (long *)malloc(sizeof(long) * 2 * (size_t)(T + 1));
    long *mid = (long *)malloc(sizeof(long) * 2 * (size_t)(T + 1));
    char *merged = (char *)calloc((size_t)(T + 1), 1);
    float *simp = (float *)malloc(sizeof(float) * (size_t)(T + 1));
    float *simn = (float *)malloc(sizeof(float) * (size_t)(T + 1));
    long nseg = 0, nmid = 0;

    /* phase 1: greedy scan (segment_utils.py:78-108) */
    long s = -1;
    long seg_cnt = 0;
    for (long i = 0; i < T; i++) {
        const float *st = states + i * d;
        float nrm = norms_in ? norms_in[i] : sqrtf(np_dot(st, st, d, tmp) + 1e-8f);
        int speech = nrm >= norm_thr;
        if (!speech) {
            if (s > -1) { seg[2 * nseg] = s; seg[2 * nseg + 1] = i; nseg++; }
            s = -1;
            seg_cnt = 0;
        } else if (seg_cnt == 0) {
            memcpy(curr, st, sizeof(float) * (size_t)d);
            seg_cnt = 1;
            s = i;
        } else {
            float sim = cossim_scalar(curr, st, d, tmp);
            if (sim >= merge_thr) {
                float c = (float)seg_cnt, c1 = (float)(seg_cnt + 1);
                for (long j = 0; j < d; j++) curr[j] = (curr[j] * c + st[j]) / c1;
                seg_cnt += 1;
            } else {
                memcpy(curr, st, sizeof(float) * (size_t)d);
                seg_cnt += 1; /* NOT reset to 1: segment_utils.py:103 */
                seg[2 * nseg] = s; seg[2 * nseg + 1] = i; nseg++;
                mid[2 * nmid] = i; mid[2 * nmid + 1] = nseg - 1; nmid++;
                s = i;
            }
        }
    }
    if (s > -1) { seg[2 * nseg] = s; seg[2 * nseg + 1] = T; nseg++; }

    /* phase 2: boundary refinement / re-merge (segment_utils.py:110-128) */
    for (long m = 0; m < nmid; m++) {
        long bd = mid[2 * m], si = mid[2 * m + 1];
        if (si >= nseg - 1) continue;
        long a0 = seg[2 * si], a1 = seg[2 * si + 1];
        long b0 = seg[2 * (si + 1)], b1 = seg[2 * (si + 1) + 1];
        mean_rows(states, d, a0, a1, ca);
        mean_rows(states, d, b0, b1, cb);
        if (cossim_scalar(ca, cb, d, tmp) >= merge_thr) {
            seg[2 * (si + 1)] = a0;
            merged[si] = 1;
            continue;
        }
        long la = (a1 - a0) / 2, lb = (b1 - b0) / 2;
        long ws = bd - (la > 1 ? la : 1);
        if (ws < a0) ws = a0;
        long we = bd + (lb > 1 ? lb : 1);
        if (we > b1) we = b1;
        long w = we - ws;
        for (long j = 0; j < w; j++) {
            simp[j] = cossim_array(states + (ws + j) * d, ca, d, tmp);
            simn[j] = cossim_array(states + (ws + j) * d, cb, d, tmp);
        }
        /* argmax_i (sim_prev[:i].sum() + sim_next[i:].sum()), first maximum, NaN-first like numpy */
        long best = 0;
        float bestv = 0.f;
        for (long i = 0; i < w; i++) {
            float v = np_sum(simp, i) + np_sum(simn + i, w - i);
            if (i == 0) { bestv = v; best = 0; if (isnan(v)) break; }
            else if (!(v <= bestv)) { bestv = v; best = i; if (isnan(v)) break; }
        }
        long opt = ws + best;
        seg[2 * si + 1] = opt;
        seg[2 * (si + 1)] = opt;
    }
    long n = 0;
    for (long i = 0; i < nseg; i++) {
        if (merged[i]) continue;
        segments_out[2 * n] = seg[2 * i];
        segments_out[2 * n + 1] = seg[2 * i + 1];
        n++;
    }
    free(tmp); free(curr); free(ca); free(cb); free(seg); free(mid); free(merged); free(simp); free(simn);
    return n;
}

/* segment mean-pool, sylber/model/sylber.py:133 */
void sylber_oracle_mean_pool(const float *states, long d, const int64_t *segments, long nseg, float *out)
{
    for (long i = 0; i < nseg; i++) mean_rows(states, d, (long)segments[2 * i], (long)segments[2 * i + 1], out + i * d);
}

/* exposed for unit tests of the summation order */
float sylber_oracle_np_sum(const float *a, long n) { return np_sum(a, n); }
float sylber_oracle_powf_half(float v) { return powf(v, 0.5f); }
