// Syllable-boundary detection + segment mean-pool on the GPU, bit-exact w.r.t. the reference's numpy
// float32 arithmetic.  One workgroup (4 waves) per utterance.
//
// Reference: get_segment / cossim, sylber/utils/segment_utils.py:68-131, and the mean-pool at
// sylber/model/sylber.py:133.  Segment indices come out of discontinuous float comparisons, so the
// kernel reproduces numpy's evaluation ORDER, not just its maths (spec: SURVEY.md §8(a) row S1):
//   * ndarray.sum over 768 contiguous f32 = 0 + pairwise sum: 8 leaf blocks of 96, each with 8
//     stride-8 accumulators (12 sequential adds), combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), leaves
//     combined by a balanced tree.  That is exactly 64 independent 12-term chains + a 6-level xor
//     butterfly: one wave64 evaluates one 768-term dot product with lane = 8*leaf + accumulator.
//   * no FMA contraction anywhere (this file is compiled with -ffp-contract=off), IEEE division and
//     sqrt (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt), f32 subnormals kept.
//   * 1-D cossim uses numpy-scalar `** .5` = glibc powf (powf_half.h); 2-D cossim uses sqrtf.
//   * mean(0) accumulates rows sequentially from +0, then divides by f32(n).
// Phase 1 (greedy scan) is inherently sequential over frames and runs on wave 0 with frames prefetched
// four ahead; norms, phase 2 (means, window similarities, sweep) and the pooling use all 256 threads
// (pooling: one segment per wave).
#pragma clang fp contract(off)
#include "kernels.h"
#include "powf_half.h"

#define SEG_D 768

// lane l owns elements 96*(l>>3) + (l&7) + 8*i, i = 0..11, of a 768-vector
__device__ __forceinline__ void load_pw(const float* __restrict__ row, int lane, float (&x)[12]) {
    const float* p = row + (lane >> 3) * 96 + (lane & 7);
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = p[8 * i];
}
// cross-lane partner values through DPP (VALU, a few cycles) instead of ds_bpermute (LDS round trip,
// ~50 cycles): these sit on the sequential critical path of the greedy scan, six per dot product.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// numpy (x*y).sum(-1) for 768 contiguous f32; result broadcast to every lane.
// lane = 8*leaf + accumulator.  Tree: ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) per leaf, then leaves pairwise.
// Every add is commutative, so a partner permutation that pairs the right GROUPS is enough once the
// values inside a group are already equal: xor1, xor2 = quad_perm; "xor4" = row_half_mirror (lane i <-> 7-i:
// the other quad of the same leaf); "xor8" = row_mirror (i <-> 15-i: the other leaf of the pair).  The four
// 16-lane row sums R0..R3 (= leaf pairs) are then read as scalars and combined as (R0+R1)+(R2+R3).
__device__ __forceinline__ float pw_dot(const float (&x)[12], const float (&y)[12]) {
    float r = x[0] * y[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) { const float p = x[i] * y[i]; r = r + p; }
    r = r + dpp_f32<0xB1>(r);     // quad_perm [1,0,3,2]
    r = r + dpp_f32<0x4E>(r);     // quad_perm [2,3,0,1]
    r = r + dpp_f32<0x141>(r);    // row_half_mirror
    r = r + dpp_f32<0x140>(r);    // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 48));
    return 0.0f + ((r0 + r1) + (r2 + r3));
}

// numpy pairwise sum of n contiguous f32 read by ONE thread (sweep sums; n is the window length)
__device__ float np_pairwise_leaf(const float* a, int n) {
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res = res + a[i];
        return res;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res = res + a[i];
    return res;
}
__device__ float np_sum_thread(const float* a, int n) {
    if (n <= 128) return 0.0f + np_pairwise_leaf(a, n);
    // explicit post-order walk of the recursion  f(n) = f(n2) + f(n - n2),  n2 = n/2 - (n/2)%8
    int off[16], len[16], st[16];
    float val[16];
    int sp = 0, vp = 0;
    off[0] = 0; len[0] = n; st[0] = 0; sp = 1;
    while (sp > 0) {
        const int o = off[sp - 1], l = len[sp - 1], s = st[sp - 1];
        if (l <= 128) { val[vp++] = np_pairwise_leaf(a + o, l); --sp; }
        else if (s == 0) {
            int n2 = l / 2; n2 -= n2 % 8;
            st[sp - 1] = 1;
            off[sp] = o + n2; len[sp] = l - n2; st[sp] = 0; ++sp;   // right (evaluated second-to-top)
            off[sp] = o; len[sp] = n2; st[sp] = 0; ++sp;            // left first
        } else { const float rgt = val[--vp]; const float lft = val[--vp]; val[vp++] = lft + rgt; --sp; }
    }
    return 0.0f + val[0];
}

// states[s:e].mean(0) -> dst (LDS or global) by a group of NT threads (id 0..NT-1), thread id owning dims id + NT*j:
// rows accumulated sequentially from +0, then one division by f32(n) -- numpy's mean(0) per element
template <int NT>
__device__ __forceinline__ void mean_rows(const float* __restrict__ states, int s, int e, int id, float* dst) {
    constexpr int ND = SEG_D / NT;
    float a[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) a[j] = 0.f;
    const float* p = states + (size_t)s * SEG_D + id;
#pragma unroll 2
    for (int r = s; r < e; ++r) {
#pragma unroll
        for (int j = 0; j < ND; ++j) a[j] = a[j] + p[NT * j];
        p += SEG_D;
    }
    const float n = (float)(e - s);
#pragma unroll
    for (int j = 0; j < ND; ++j) dst[id + NT * j] = a[j] / n;
}

// GS = false: all bookkeeping lives in LDS (dynamic, sized by T): the refinement loop is a chain of dependent reads of
// the segment table, which from global memory cost an L2 round trip per step.  That fits T <= 3940 frames (78.8 s of
// audio) in the 160 KiB of a CU.  GS = true (longer utterances; the reference's get_segment has no length limit): the
// T-sized arrays move to a per-utterance slab of global scratch, only the two 768-float centroids and the counters
// stay in LDS; same arithmetic, same order, same results -- a workgroup's own global writes are visible to all of its
// waves after __syncthreads().
template <bool GS>
__global__ __launch_bounds__(256) void segment_kernel(const float* __restrict__ hidden, int T, float norm_thr, float merge_thr,
                                                      int64_t* __restrict__ seg_out, int* __restrict__ nseg_out,
                                                      float* __restrict__ feat_out, float* __restrict__ scratch, size_t scratch_stride) {
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    float* ca_s = lds_f;                       // [768]
    float* cb_s = ca_s + SEG_D;                // [768]
    int* sh_i = (int*)(cb_s + SEG_D);          // [8]
    float* tbase;
    if constexpr (GS) tbase = scratch + (size_t)blockIdx.x * scratch_stride;
    else tbase = cb_s + SEG_D + 8;
    float* simp_s = tbase;                     // [T]
    float* simn_s = simp_s + T;                // [T]
    float* sweep_s = simn_s + T;               // [T]
    float* nsq = sweep_s + T;                  // [T] sqrt path norms (mask; 2-D cossim)
    float* npw = nsq + T;                      // [T] powf path norms (1-D cossim)
    int* seg = (int*)(npw + T);                // [T+1][2]
    int* mid = seg + 2 * (T + 1);              // [T+1][2]
    int* merged = mid + 2 * (T + 1);           // [T+1]
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* states = hidden + (size_t)b * T * SEG_D;

    // ---- phase 0: frame norms.  Row sums with four rows per wave in flight; the square root and the ~600-cycle glibc
    // powf are then evaluated one ROW PER THREAD (as a per-row tail of the loop they ran once per row on a whole wave)
    for (int i0 = 4 * wave; i0 < T; i0 += 16) {
        float x[4][12];
#pragma unroll
        for (int r = 0; r < 4; ++r) load_pw(states + (size_t)(i0 + r < T ? i0 + r : T - 1) * SEG_D, lane, x[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ss = pw_dot(x[r], x[r]) + 1e-8f;
            if (lane == 0 && i0 + r < T) nsq[i0 + r] = ss;
        }
    }
    __syncthreads();
    for (int i = tid; i < T; i += 256) {
        const float ss = nsq[i];
        nsq[i] = sqrtf(ss); npw[i] = powf_half_glibc(ss);
    }
    for (int i = tid; i <= T; i += 256) merged[i] = 0;
    __syncthreads();

    // ---- phase 1: greedy scan (segment_utils.py:78-108), wave 0
    if (wave == 0) {
        int s = -1, seg_cnt = 0, nseg = 0, nmid = 0;
        float c[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) c[i] = 0.f;
        // frames are processed in groups of 4 with the next group already in flight (two groups ahead measured slower:
        // the scan is bound by its dependent arithmetic, not by memory)
        float g0[4][12], g1[4][12];
        const int T4 = (T + 3) & ~3;
        auto ldg = [&](int i, float (&g)[4][12]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) load_pw(states + (size_t)(i + r < T ? i + r : T - 1) * SEG_D, lane, g[r]);
        };
        ldg(0, g0);
        for (int i0 = 0; i0 < T4; i0 += 4) {
            ldg(i0 + 4, g1);
            auto step = [&](int i, const float (&x)[12]) {
                if (i >= T) return;
                const bool speech = nsq[i] >= norm_thr;
                if (!speech) {
                    if (s > -1) { if (lane == 0) { seg[2 * nseg] = s; seg[2 * nseg + 1] = i; } ++nseg; }
                    s = -1; seg_cnt = 0;
                } else if (seg_cnt == 0) {
#pragma unroll
                    for (int k = 0; k < 12; ++k) c[k] = x[k];
                    seg_cnt = 1; s = i;
                } else {
                    // The decision needs the reference's exact float32 value only near the threshold.  An
                    // estimate with relative error < 1e-5 (v_rsq / v_rcp, 1 ulp each, against <= 2 ulp of the
                    // exact chain) settles every frame outside a +-1e-4 guard band; inside the band the exact
                    // chain (glibc powf + two IEEE divisions, ~600 dependent cycles) is evaluated.  Same
                    // decisions as the reference, bit for bit; the slow path just stops being per-frame.
                    const float dot = pw_dot(c, x);
                    const float cc = pw_dot(c, c) + 1e-8f;
                    const float est = dot * __builtin_amdgcn_rsqf(cc) * __builtin_amdgcn_rcpf(npw[i]);
                    bool merge;
                    if (est > merge_thr + 1e-4f) merge = true;
                    else if (est < merge_thr - 1e-4f) merge = false;
                    else merge = (dot / powf_half_glibc(cc) / npw[i]) >= merge_thr;   // also taken for NaN
                    if (merge) {
                        // c = (c * n + x) / (n + 1), twelve IEEE divisions by the same small integer per frame: ~140 of
                        // the ~250 instructions of a merge step.  Same quotients, bit for bit, from ONE division:
                        // r = RN(1 / c1) (IEEE), q0 = a r, then two residual corrections q <- fma(fma(-q, c1, a), r, q)
                        // (Markstein: with r correctly rounded and q faithful, the corrected quotient is RN(a / c1);
                        // checked against a / c1 for every divisor <= 8192 x 3.2e8 dividends on the CPU, 0 mismatches).
                        // Dividends that are zero, below 2^-100 (inexact residuals) or non-finite take the division.
                        const float cf = (float)seg_cnt, c1 = (float)(seg_cnt + 1);
                        const float r = 1.0f / c1;
                        float a[12];
                        float amin = INFINITY, amax = 0.f;
#pragma unroll
                        for (int k = 0; k < 12; ++k) {
                            a[k] = c[k] * cf + x[k];
                            amin = fminf(amin, fabsf(a[k])); amax = fmaxf(amax, fabsf(a[k]));
                        }
                        if (__builtin_amdgcn_ballot_w64(!(amin >= 0x1p-100f) || !(amax <= 0x1p126f))) {
#pragma unroll
                            for (int k = 0; k < 12; ++k) c[k] = a[k] / c1;
                        } else {
#pragma unroll
                            for (int k = 0; k < 12; ++k) {
                                const float q0 = a[k] * r;
                                const float q1 = fmaf(fmaf(-q0, c1, a[k]), r, q0);
                                c[k] = fmaf(fmaf(-q1, c1, a[k]), r, q1);
                            }
                        }
                        seg_cnt += 1;
                    } else {
#pragma unroll
                        for (int k = 0; k < 12; ++k) c[k] = x[k];
                        seg_cnt += 1;                      // NOT reset (segment_utils.py:103)
                        if (lane == 0) { seg[2 * nseg] = s; seg[2 * nseg + 1] = i; mid[2 * nmid] = i; mid[2 * nmid + 1] = nseg; }
                        ++nseg; ++nmid;
                        s = i;
                    }
                }
            };
            step(i0, g0[0]); step(i0 + 1, g0[1]); step(i0 + 2, g0[2]); step(i0 + 3, g0[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 12; ++k) g0[r][k] = g1[r][k];
        }
        if (s > -1) { if (lane == 0) { seg[2 * nseg] = s; seg[2 * nseg + 1] = T; } ++nseg; }
        if (lane == 0) { sh_i[0] = nseg; sh_i[1] = nmid; }
    }
    __syncthreads();
    const int nseg = sh_i[0], nmid = sh_i[1];

    // ---- phase 2: boundary refinement / re-merge (segment_utils.py:110-128), whole workgroup
    for (int m = 0; m < nmid; ++m) {
        const int bd = mid[2 * m], si = mid[2 * m + 1];
        if (si >= nseg - 1) continue;
        const int a0 = seg[2 * si], a1 = seg[2 * si + 1];
        const int b0 = seg[2 * si + 2], b1 = seg[2 * si + 3];
        // the window of the sweep depends on the segment table only, so this wave's first window row is requested now and
        // arrives under the centroid means (one global round trip per boundary instead of two)
        const int la = (a1 - a0) / 2, lb = (b1 - b0) / 2;
        int ws = bd - (la > 1 ? la : 1); ws = ws < a0 ? a0 : ws;
        int we = bd + (lb > 1 ? lb : 1); we = we > b1 ? b1 : we;
        const int w = we - ws;
        float xw[12];
        load_pw(states + (size_t)(wave < w ? ws + wave : a0) * SEG_D, lane, xw);
        if (tid < 128) mean_rows<128>(states, a0, a1, tid, ca_s);          // the two centroids concurrently, one per
        else mean_rows<128>(states, b0, b1, tid - 128, cb_s);              // half workgroup
        __syncthreads();
        float ca[12], cb[12];
        load_pw(ca_s, lane, ca);
        load_pw(cb_s, lane, cb);
        const float saa = pw_dot(ca, ca) + 1e-8f, sbb = pw_dot(cb, cb) + 1e-8f;
        const float sim_ab = pw_dot(ca, cb) / powf_half_glibc(saa) / powf_half_glibc(sbb);   // every wave, identical
        if (sim_ab >= merge_thr) {
            __syncthreads();
            if (tid == 0) { seg[2 * si + 2] = a0; merged[si] = 1; }
            __syncthreads();
            continue;
        }
        const float nca = sqrtf(saa), ncb = sqrtf(sbb);
        for (int j = wave; j < w; j += 4) {
            float x[12];
            if (j == wave) {
#pragma unroll
                for (int k = 0; k < 12; ++k) x[k] = xw[k];
            } else load_pw(states + (size_t)(ws + j) * SEG_D, lane, x);
            const float nx = nsq[ws + j];
            const float sp = pw_dot(x, ca) / nx / nca;
            const float sn = pw_dot(x, cb) / nx / ncb;
            if (lane == 0) { simp_s[j] = sp; simn_s[j] = sn; }
        }
        __syncthreads();
        for (int i = tid; i < w; i += 256) sweep_s[i] = np_sum_thread(simp_s, i) + np_sum_thread(simn_s + i, w - i);
        __syncthreads();
        if (tid == 0) {
            int best = 0;
            float bv = sweep_s[0];
            if (!(bv != bv)) {
                for (int i = 1; i < w; ++i) {
                    const float v = sweep_s[i];
                    if (!(v <= bv)) { bv = v; best = i; if (v != v) break; }
                }
            }
            seg[2 * si + 1] = ws + best;
            seg[2 * si + 2] = ws + best;
        }
        __syncthreads();
    }
    __syncthreads();

    // ---- compaction + mean-pool (sylber.py:133)
    if (tid == 0) {
        int n = 0;
        for (int i = 0; i < nseg; ++i) {
            if (merged[i]) continue;
            seg_out[((size_t)b * T + n) * 2 + 0] = seg[2 * i];
            seg_out[((size_t)b * T + n) * 2 + 1] = seg[2 * i + 1];
            // reuse mid[] as the compacted list for the pooling loop
            mid[2 * n] = seg[2 * i]; mid[2 * n + 1] = seg[2 * i + 1];
            ++n;
        }
        nseg_out[b] = n;
        sh_i[2] = n;
    }
    __syncthreads();
    if (feat_out) {
        const int n = sh_i[2];
        for (int k = wave; k < n; k += 4)                                   // one segment per wave, four in flight
            mean_rows<64>(states, mid[2 * k], mid[2 * k + 1], lane, feat_out + ((size_t)b * T + k) * SEG_D);
    }
}

// ================================================================================================================================
// Round 6: the same algorithm WIDE.  The one-workgroup-per-utterance kernel above keeps 8 (8 x 60 s) or 32 (32 x 10 s) of the 256 CUs
// busy for 1.2 / 0.23 ms; on the critical path of the synchronous Segmenter.__call__ that is 12 % of a long-form call.  What the
// reference computes decomposes further than "one utterance":
//   * frame norms: independent rows                                                     -> segment_norms_kernel, all CUs
//   * greedy scan + refinement: the scan's state is RESET by every non-speech frame (segment_utils.py:84-90: s = -1, seg_cnt = 0; `curr`
//     is overwritten by the next speech frame before it is read), and a mid-boundary only ever pairs two segments of the same unbroken
//     run of speech frames (it is recorded where a run continues, :102-106).  So every maximal run of speech frames is an independent
//     instance of phases 1 and 2: one workgroup per RUN (segment_runs_kernel), the arithmetic of a run statement for statement the
//     code above (same wave layout, same order, same slow-path decisions); run-local segment tables land in a per-utterance slot table
//     indexed by the run's first frame (a run of n frames has at most n segments, runs are disjoint: no two runs share a slot)
//   * the table in frame order = the live slots in slot order                           -> segment_compact_kernel (prefix count)
//   * mean-pool: independent segments                                                   -> segment_pool_kernel, one wave per segment, all CUs
// Bit-identical to the kernel above and to the oracle on every golden (tests/test_gpu_segment.py runs both).
// A run's bookkeeping lives in LDS up to SEG_LCAP frames (10 s of unbroken speech); longer runs are taken by the GS = true launch of
// the same kernel, which keeps it in the utterance's global slab (launched only when T > SEG_LCAP can hold such a run at all).
#define SEG_LCAP 512

// per-utterance slab (floats): nsq [T], npw [T], slot table [(T+1) x 2] ints, live [T+1] ints, then (GS runs) simp / simn / sweep [T] each,
// mid [(T+1) x 2], merged [T+1]
__host__ __device__ inline size_t seg_wide_slab_floats(int T) { return (5 * (size_t)T + 6 * ((size_t)T + 1) + 63) & ~(size_t)63; }

__global__ __launch_bounds__(256) void segment_norms_kernel(const float* __restrict__ hidden, int B, int T, float* __restrict__ scratch, size_t slab) {
    __shared__ float ss_s[64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long rows = (long)B * T, r0 = (long)blockIdx.x * 64 + wave * 16;
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        float x[4][12];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const long g = r0 + 4 * q + r; load_pw(hidden + (size_t)(g < rows ? g : rows - 1) * SEG_D, lane, x[r]); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ss = pw_dot(x[r], x[r]) + 1e-8f;
            if (lane == 0) ss_s[wave * 16 + 4 * q + r] = ss;
        }
    }
    __syncthreads();
    if (tid < 64) {                                          // the square root and the ~600-cycle glibc powf: one ROW per thread
        const long g = (long)blockIdx.x * 64 + tid;
        if (g < rows) {
            const int b = (int)(g / T), t = (int)(g % T);
            float* sl = scratch + (size_t)b * slab;
            const float ss = ss_s[tid];
            sl[t] = sqrtf(ss);
            sl[T + t] = powf_half_glibc(ss);
            int* live = (int*)(sl + 2 * (size_t)T) + 2 * ((size_t)T + 1);
            live[t] = 0;
            if (t == T - 1) live[T] = 0;
        }
    }
}

template <bool GS>
__global__ __launch_bounds__(256) void segment_runs_kernel(const float* __restrict__ hidden, int T, float norm_thr, float merge_thr,
                                                           float* __restrict__ scratch, size_t slab, int list_cap) {
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    float* ca_s = lds_f;                       // [768]
    float* cb_s = ca_s + SEG_D;                // [768]
    int* sh_i = (int*)(cb_s + SEG_D);          // [16]
    int* mystart = sh_i + 16;                  // [list_cap]
    const int b = blockIdx.y, rx = blockIdx.x, G = gridDim.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* states = hidden + (size_t)b * T * SEG_D;
    float* sl = scratch + (size_t)b * slab;
    const float* g_nsq = sl;
    const float* g_npw = sl + T;
    int* g_slot = (int*)(sl + 2 * (size_t)T);  // [(T+1)][2]
    int* g_live = g_slot + 2 * ((size_t)T + 1);

    // ---- which runs are mine: run r (in frame order) belongs to workgroup r % G
    if (tid < 16) sh_i[tid] = 0;
    __syncthreads();
    int running = 0;
    for (int c0 = 0; c0 < T; c0 += 256) {
        const int i = c0 + tid;
        const bool sp = i < T && g_nsq[i] >= norm_thr;
        const bool pv = i > 0 && i < T && g_nsq[i - 1] >= norm_thr;
        const bool st = sp && !pv;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(st);
        if (lane == 0) sh_i[4 + wave] = __builtin_popcountll(bal);
        __syncthreads();
        int off = running;
        for (int w = 0; w < wave; ++w) off += sh_i[4 + w];
        const int tot = sh_i[4] + sh_i[5] + sh_i[6] + sh_i[7];
        if (st) {
            const int ord = off + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
            if (ord % G == rx) mystart[ord / G] = i;
        }
        running += tot;
        __syncthreads();
    }
    const int mine = running > rx ? (running - rx + G - 1) / G : 0;

    for (int jr = 0; jr < mine; ++jr) {
        __syncthreads();                                     // the previous run's LDS is free
        const int base = mystart[jr];
        // ---- the run's end: the first non-speech frame behind its start, or T
        if (tid == 0) sh_i[8] = T;
        __syncthreads();
        for (int c0 = base + 1; c0 < T; c0 += 256) {
            const int i = c0 + tid;
            const bool ns = i < T && !(g_nsq[i] >= norm_thr);
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(ns);
            if (bal != 0ull && lane == 0) atomicMin(&sh_i[8], c0 + wave * 64 + (int)__builtin_ctzll(bal));
            __syncthreads();
            const int e = sh_i[8];
            __syncthreads();
            if (e < T) break;
        }
        const int end = sh_i[8];
        const int len = end - base;
        if ((len > SEG_LCAP) != GS) continue;                // (uniform) the other launch's run

        float *simp_s, *simn_s, *sweep_s;
        const float *nsq, *npw;                              // run-relative: frame i at [i - base]
        int *seg, *mid, *merged;
        if constexpr (GS) {
            float* g_simp = (float*)(g_live + ((size_t)T + 1));
            simp_s = g_simp + base; simn_s = g_simp + T + base; sweep_s = g_simp + 2 * (size_t)T + base;
            nsq = g_nsq + base; npw = g_npw + base;
            seg = g_slot + 2 * (size_t)base;                 // the slot table itself
            mid = (int*)(g_simp + 3 * (size_t)T) + 2 * (size_t)base;
            merged = (int*)(g_simp + 3 * (size_t)T) + 2 * ((size_t)T + 1) + base;
        } else {
            float* tb = (float*)(mystart + list_cap);
            simp_s = tb; simn_s = simp_s + SEG_LCAP; sweep_s = simn_s + SEG_LCAP;
            float* nsq_w = sweep_s + SEG_LCAP; float* npw_w = nsq_w + SEG_LCAP;
            seg = (int*)(npw_w + SEG_LCAP); mid = seg + 2 * (SEG_LCAP + 1); merged = mid + 2 * (SEG_LCAP + 1);
            for (int i = tid; i < len; i += 256) { nsq_w[i] = g_nsq[base + i]; npw_w[i] = g_npw[base + i]; }
            nsq = nsq_w; npw = npw_w;
        }
        for (int i = tid; i < len; i += 256) merged[i] = 0;
        __syncthreads();

        // ---- phase 1: greedy scan over the run's frames [base, end), all of them speech (segment_utils.py:78-108), wave 0
        if (wave == 0) {
            int s = -1, seg_cnt = 0, nseg = 0, nmid = 0;
            float c[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) c[i] = 0.f;
            float g0[4][12], g1[4][12];
            auto ldg = [&](int i, float (&g)[4][12]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) load_pw(states + (size_t)(i + r < T ? i + r : T - 1) * SEG_D, lane, g[r]);
            };
            ldg(base, g0);
            for (int i0 = base; i0 < end; i0 += 4) {
                ldg(i0 + 4, g1);
                auto step = [&](int i, const float (&x)[12]) {
                    if (i >= end) return;
                    if (seg_cnt == 0) {
#pragma unroll
                        for (int k = 0; k < 12; ++k) c[k] = x[k];
                        seg_cnt = 1; s = i;
                    } else {
                        // (estimate + guard band, exact chain inside the band; one division + two FMA corrections per element: see segment_kernel)
                        const float dot = pw_dot(c, x);
                        const float cc = pw_dot(c, c) + 1e-8f;
                        const float est = dot * __builtin_amdgcn_rsqf(cc) * __builtin_amdgcn_rcpf(npw[i - base]);
                        bool merge;
                        if (est > merge_thr + 1e-4f) merge = true;
                        else if (est < merge_thr - 1e-4f) merge = false;
                        else merge = (dot / powf_half_glibc(cc) / npw[i - base]) >= merge_thr;   // also taken for NaN
                        if (merge) {
                            const float cf = (float)seg_cnt, c1 = (float)(seg_cnt + 1);
                            const float r = 1.0f / c1;
                            float a[12];
                            float amin = INFINITY, amax = 0.f;
#pragma unroll
                            for (int k = 0; k < 12; ++k) {
                                a[k] = c[k] * cf + x[k];
                                amin = fminf(amin, fabsf(a[k])); amax = fmaxf(amax, fabsf(a[k]));
                            }
                            if (__builtin_amdgcn_ballot_w64(!(amin >= 0x1p-100f) || !(amax <= 0x1p126f))) {
#pragma unroll
                                for (int k = 0; k < 12; ++k) c[k] = a[k] / c1;
                            } else {
#pragma unroll
                                for (int k = 0; k < 12; ++k) {
                                    const float q0 = a[k] * r;
                                    const float q1 = fmaf(fmaf(-q0, c1, a[k]), r, q0);
                                    c[k] = fmaf(fmaf(-q1, c1, a[k]), r, q1);
                                }
                            }
                            seg_cnt += 1;
                        } else {
#pragma unroll
                            for (int k = 0; k < 12; ++k) c[k] = x[k];
                            seg_cnt += 1;                      // NOT reset (segment_utils.py:103)
                            if (lane == 0) { seg[2 * nseg] = s; seg[2 * nseg + 1] = i; mid[2 * nmid] = i; mid[2 * nmid + 1] = nseg; }
                            ++nseg; ++nmid;
                            s = i;
                        }
                    }
                };
                step(i0, g0[0]); step(i0 + 1, g0[1]); step(i0 + 2, g0[2]); step(i0 + 3, g0[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int k = 0; k < 12; ++k) g0[r][k] = g1[r][k];
            }
            if (s > -1) { if (lane == 0) { seg[2 * nseg] = s; seg[2 * nseg + 1] = end; } ++nseg; }   // closed by the silence frame at `end`, or by T
            if (lane == 0) { sh_i[0] = nseg; sh_i[1] = nmid; }
        }
        __syncthreads();
        const int nseg = sh_i[0], nmid = sh_i[1];

        // ---- phase 2: boundary refinement / re-merge (segment_utils.py:110-128), whole workgroup
        for (int m = 0; m < nmid; ++m) {
            const int bd = mid[2 * m], si = mid[2 * m + 1];
            if (si >= nseg - 1) continue;
            const int a0 = seg[2 * si], a1 = seg[2 * si + 1];
            const int b0 = seg[2 * si + 2], b1 = seg[2 * si + 3];
            const int la = (a1 - a0) / 2, lb = (b1 - b0) / 2;
            int ws = bd - (la > 1 ? la : 1); ws = ws < a0 ? a0 : ws;
            int we = bd + (lb > 1 ? lb : 1); we = we > b1 ? b1 : we;
            const int w = we - ws;
            float xw[12];
            load_pw(states + (size_t)(wave < w ? ws + wave : a0) * SEG_D, lane, xw);
            if (tid < 128) mean_rows<128>(states, a0, a1, tid, ca_s);
            else mean_rows<128>(states, b0, b1, tid - 128, cb_s);
            __syncthreads();
            float ca[12], cb[12];
            load_pw(ca_s, lane, ca);
            load_pw(cb_s, lane, cb);
            const float saa = pw_dot(ca, ca) + 1e-8f, sbb = pw_dot(cb, cb) + 1e-8f;
            const float sim_ab = pw_dot(ca, cb) / powf_half_glibc(saa) / powf_half_glibc(sbb);   // every wave, identical
            if (sim_ab >= merge_thr) {
                __syncthreads();
                if (tid == 0) { seg[2 * si + 2] = a0; merged[si] = 1; }
                __syncthreads();
                continue;
            }
            const float nca = sqrtf(saa), ncb = sqrtf(sbb);
            for (int j = wave; j < w; j += 4) {
                float x[12];
                if (j == wave) {
#pragma unroll
                    for (int k = 0; k < 12; ++k) x[k] = xw[k];
                } else load_pw(states + (size_t)(ws + j) * SEG_D, lane, x);
                const float nx = nsq[ws + j - base];
                const float sp = pw_dot(x, ca) / nx / nca;
                const float sn = pw_dot(x, cb) / nx / ncb;
                if (lane == 0) { simp_s[j] = sp; simn_s[j] = sn; }
            }
            __syncthreads();
            for (int i = tid; i < w; i += 256) sweep_s[i] = np_sum_thread(simp_s, i) + np_sum_thread(simn_s + i, w - i);
            __syncthreads();
            if (tid == 0) {
                int best = 0;
                float bv = sweep_s[0];
                if (!(bv != bv)) {
                    for (int i = 1; i < w; ++i) {
                        const float v = sweep_s[i];
                        if (!(v <= bv)) { bv = v; best = i; if (v != v) break; }
                    }
                }
                seg[2 * si + 1] = ws + best;
                seg[2 * si + 2] = ws + best;
            }
            __syncthreads();
        }
        __syncthreads();
        // ---- the run's segments into the utterance's slot table (slots base .. base + nseg - 1; the rest of the run's slots stay dead)
        for (int k = tid; k < nseg; k += 256) {
            if constexpr (!GS) { g_slot[2 * (size_t)(base + k)] = seg[2 * k]; g_slot[2 * (size_t)(base + k) + 1] = seg[2 * k + 1]; }
            g_live[base + k] = merged[k] ? 0 : 1;
        }
    }
}

// live slots in slot order = the reference's table (np.array(segments) without the merged indices, segment_utils.py:130-131)
__global__ __launch_bounds__(256) void segment_compact_kernel(int T, const float* __restrict__ scratch, size_t slab, int64_t* __restrict__ seg_out,
                                                              int* __restrict__ nseg_out) {
    __shared__ int cnt_s[4];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* sl = scratch + (size_t)b * slab;
    const int* g_slot = (const int*)(sl + 2 * (size_t)T);
    const int* g_live = g_slot + 2 * ((size_t)T + 1);
    int running = 0;
    for (int c0 = 0; c0 < T; c0 += 256) {
        const int i = c0 + tid;
        const bool lv = i < T && g_live[i] != 0;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(lv);
        if (lane == 0) cnt_s[wave] = __builtin_popcountll(bal);
        __syncthreads();
        int off = running;
        for (int w = 0; w < wave; ++w) off += cnt_s[w];
        const int tot = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3];
        if (lv) {
            const int n = off + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
            seg_out[((size_t)b * T + n) * 2 + 0] = g_slot[2 * (size_t)i];
            seg_out[((size_t)b * T + n) * 2 + 1] = g_slot[2 * (size_t)i + 1];
        }
        running += tot;
        __syncthreads();
    }
    if (tid == 0) nseg_out[b] = running;
}

// states[s:e].mean(0) per segment (sylber.py:133): one wave per segment, the utterance's segments spread over gridDim.x workgroups
__global__ __launch_bounds__(256) void segment_pool_kernel(const float* __restrict__ hidden, int T, const int64_t* __restrict__ seg_out,
                                                           const int* __restrict__ nseg_out, float* __restrict__ feat_out) {
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* states = hidden + (size_t)b * T * SEG_D;
    const int n = nseg_out[b];
    for (int k = blockIdx.x * 4 + wave; k < n; k += gridDim.x * 4) {
        const int s = (int)seg_out[((size_t)b * T + k) * 2], e = (int)seg_out[((size_t)b * T + k) * 2 + 1];
        mean_rows<64>(states, s, e, lane, feat_out + ((size_t)b * T + k) * SEG_D);
    }
}


static size_t segment_tsized_floats(int T) { return 5 * (size_t)T + 5 * ((size_t)T + 1); }
static size_t segment_lds_bytes(int T) { return ((size_t)2 * SEG_D + 8 + segment_tsized_floats(T)) * 4; }
static size_t segment_slab_floats(int T) { return (segment_tsized_floats(T) + 63) & ~(size_t)63; }

// floats of global scratch launch_segment needs: the wide path's per-utterance slab (norms, slot table, and the bookkeeping of runs
// beyond SEG_LCAP frames); it also covers the one-workgroup-per-utterance kernel's slab for utterances beyond 3940 frames
size_t segment_scratch_floats(int B, int T, int D) {
    (void)D;
    const size_t a = seg_wide_slab_floats(T), o = segment_slab_floats(T);
    return (size_t)B * (a > o ? a : o);
}

// mode 0: the wide path (norms / runs / compaction / pooling on all CUs); mode -1: one workgroup per utterance (rounds 1-5; A/B and bitwise reference)
int launch_segment(const float* hidden, int B, int T, int D, float norm_thr, float merge_thr, int64_t* seg, int* nseg,
                   float* feat, float* scratch, hipStream_t s, int mode) {
    if (D != SEG_D) { syl_set_error("launch_segment", "feature dim must be 768"); return 1; }
    if (T < 1) { syl_set_error("launch_segment", "T must be >= 1"); return 1; }
    if (mode >= 0) {
        if (!scratch) { syl_set_error("launch_segment", "the wide path needs its scratch slab (segment_scratch_floats)"); return 1; }
        const size_t slab = seg_wide_slab_floats(T);
        const long rows = (long)B * T;
        hipLaunchKernelGGL(segment_norms_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, s, hidden, B, T, scratch, slab);
        int G = (T + 7) / 8; G = G < 8 ? 8 : (G > 128 ? 128 : G);
        const int list_cap = ((T + 1) / 2 + G - 1) / G + 1;
        const size_t lds_l = ((size_t)2 * SEG_D + 16 + list_cap + 5 * SEG_LCAP + 5 * (SEG_LCAP + 1)) * 4;
        const size_t lds_g = ((size_t)2 * SEG_D + 16 + list_cap) * 4;
        if (lds_l > 64 * 1024) {
            static PerDeviceOnce once;
            if (once.need()) HIP_TRY(hipFuncSetAttribute((const void*)segment_runs_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        hipLaunchKernelGGL(segment_runs_kernel<false>, dim3(G, B), dim3(256), lds_l, s, hidden, T, norm_thr, merge_thr, scratch, slab, list_cap);
        if (T > SEG_LCAP)
            hipLaunchKernelGGL(segment_runs_kernel<true>, dim3(G, B), dim3(256), lds_g, s, hidden, T, norm_thr, merge_thr, scratch, slab, list_cap);
        hipLaunchKernelGGL(segment_compact_kernel, dim3(B), dim3(256), 0, s, T, scratch, slab, seg, nseg);
        if (feat) {
            int PX = (T + 15) / 16; PX = PX < 1 ? 1 : (PX > 64 ? 64 : PX);
            hipLaunchKernelGGL(segment_pool_kernel, dim3(PX, B), dim3(256), 0, s, hidden, T, seg, nseg, feat);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const size_t lds = segment_lds_bytes(T);
    if (lds <= 160 * 1024) {
        static PerDeviceOnce once;                       // raise the limit to the full 160 KiB once per device
        if (once.need()) HIP_TRY(hipFuncSetAttribute((const void*)segment_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(segment_kernel<false>, dim3(B), dim3(256), lds, s, hidden, T, norm_thr, merge_thr, seg, nseg, feat, nullptr, (size_t)0);
    } else {
        if (!scratch) { syl_set_error("launch_segment", "utterances beyond 3940 frames need the global scratch slab"); return 1; }
        hipLaunchKernelGGL(segment_kernel<true>, dim3(B), dim3(256), (2 * SEG_D + 8) * 4, s, hidden, T, norm_thr, merge_thr, seg, nseg, feat,
                           scratch, segment_slab_floats(T));
    }
    HIP_TRY(hipGetLastError());
    return 0;
}
