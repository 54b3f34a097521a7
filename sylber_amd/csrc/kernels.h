// Internal launcher API between the HIP translation units of libsylber_hip.so.
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------------
// bf16 MFMA GEMM  out[m][n] = sum_k X[m][k] * W[n][k]  (+ fused epilogue)
// X rows may overlap (ldx < K): the strided 1-D convolutions are run as this GEMM on a channels-last
// activation buffer with ldx = stride*512, K = taps*512 (implicit GEMM, no im2col).
// ------------------------------------------------------------------------------------------------
enum GemmAct { ACT_NONE = 0, ACT_GELU_FAST = 1, ACT_GELU_ERF = 2, ACT_GELU_ERF7 = 3 };   // GemmArgs::act (bf16 / MXFP8 GEMMs); 3 = erf form, 1.5e-7 (split16)
enum GemmEpi {
    EPI_BF16 = 0,       // out0 bf16 [M][ld0] = act(acc + bias)
    EPI_F32 = 1,        // out0 f32  [M][ld0] = act(acc + bias)
    EPI_F32_RES = 2,    // out0 f32  [M][ld0] = acc + bias + res[m][n]
    EPI_QK = 3,         // fused q/k/v projection: columns [0,768) q (x SYL_Q_SCALE = log2(e) / 8) -> out0, [768,1536) k -> out1, both
                        // [B,H,Tp,64] bf16; columns [1536,2304) (N = 2304 only; needs Tp % 32 == 0) v -> out2 = Vt
                        // [B,H,64,Tpv] bf16 with the key axis bit-swapped, transposed through LDS
    EPI_PROJ = 4,       // feature projection: zero padded frames; out0 f32 [M][768]; out1 bf16 xpad
    EPI_MXFP8 = 7,      // MXFP8 GEMM only: out0 e4m3 [M][ld0] + out_scale e8m0 [M][N/32] = mx(act(acc + bias))
    EPI_QK8 = 8,        // MXFP8 GEMM only (gemm_asm_f8.hip, 256x192 tile): EPI_QK with MXFP8 outputs for the fp8 attention core:
                        // q (x0.125) -> out0, k -> out1: e4m3 [B,H,Tp,64] + scales qs / ks [B,H,Tp,2] (per 32 features);
                        // v -> out2 = Vt e4m3 [B,H,64,Tpv], natural key order, + scales vs [B,H,64,Tpv/32] (per 32 keys)
    EPI_F32_RESLN = 6,  // out0 f32 = acc + bias + LN(res[m][n]) with LN = (x - mean[m]) * rstd[m] * gamma[n] + beta[n]
                        // (the residual IS a LayerNorm output that is never materialised in fp32; out0 may alias res)
};

struct GemmArgs {
    const bf16_t* X; long ldx;
    const bf16_t* W;              // [N][K] row-major
    int M, N, K;
    int M_store;                  // EPI_QK8 only: > 0 = rows [M_store, M) are computed but not stored (M padded up to whole 256-row tiles)
    const float* bias;            // [N] or nullptr
    int act;                      // GemmAct
    void* out0; long ld0;
    void* out1; void* out2;
    const float* res; long ldres;
    int Tp, Tpv, T;               // rows per utterance in M, Vt row stride, real frames
    const int* valid;             // [B] valid frames (EPI_PROJ)
    int xpad_rows;                // rows per utterance of the zero-padded pos-conv input (EPI_PROJ)
    const float* ln_stats;        // [M][2] (mean, rstd) of the rows of `res` (EPI_F32_RESLN)
    const float* ln_gamma; const float* ln_beta;
    float* ln_stats_out;          // gemm_rowln: [M][2] (mean, rstd) of the rows this launch produces
    const float* ln_gamma_out; const float* ln_beta_out;   // gemm_rowln: affine of the LayerNorm applied to them
    int fmt;                      // FMT_BF16 (0), FMT_F16 or FMT_SPLIT: 16-bit format of X, W and of bf16-typed outputs
    long x_lo, w_lo;              // FMT_SPLIT: element offsets of the lo planes of X and W (hi plane at the pointer)
    long out_lo;                  // FMT_SPLIT: element offset of the lo plane of the 16-bit outputs out0 / out1
    long out2_lo;                 //            ... of out2 (V^T)
    unsigned long long* trace;    // development: per-phase cycle sums of two waves of workgroup 0 (trace instantiation only)
    int tune_cfg;                 // 0 = tile shape chosen by the cost model; k > 0 forces tile configuration k - 1
    int tune_persist;             // > 0: persistent launch (that many workgroups per CU walk the tile list)
    int kpat;                     // 1: the 3-tap stride-2 conv layers' chunk-major K order (K = 1536, 512 channels): W is packed
                                  // [out][64-channel chunk][tap 0, 2, 1][64] and the X byte offset of K position o follows tap3_offset(o)
    int tune_pre;                 // residual GEMMs on tile 91: -1 no residual prefetch in the K loop, 1..3 fragment columns prefetched, 0 default
    int m_begin;                  // the launch covers rows [m_begin, M) (0 = all): the tail launch of a row-split GEMM (launch_f, "tail policy");
                                  // row indices stay absolute everywhere (operands, epilogues, the row -> (utterance, frame) maps)
    int tune_model;               // 5: the round-5 cost model of launch_f (A/B switch); 0: the current one
    int tune_h192;                // -1: the cost model leaves the 192-row tiles (51 / 57) out (A/B switch)
    int tune_mfma16;              // 0 (default): the 16-bit-output GEMMs (EPI_BF16: conv1-5, FFN1) run on the v_mfma_f32_16x16x32 family (tiles 13 / 14 / 46 / 47,
                                  // gemm_asm16.hip); -1: on the 32x32x16 kernels of rounds 1-6a (A/B switch; the two families group an element's fp32 chain differently)
    int tune_tail;                // tail policy of multi-round launches: 0 automatic, -1 never split, k > 0 = force a split with tail tile id k - 1
};

int launch_gemm_bf16(int epi, const GemmArgs& a, hipStream_t s);
// the tile id launch_gemm_bf16's cost model picks for this launch (host arithmetic only: no GPU needed; tests/test_abi.py pins the headline shapes)
int gemm_pick_tile(int epi, const GemmArgs& a);
// residual GEMM + the LayerNorm that follows it in one launch (gemm_rowln.hip): N = 768, full rows per workgroup
bool gemm_rowln_applicable(const GemmArgs& a);
int launch_gemm_rowln(const GemmArgs& a, hipStream_t s);
// 256x256 tile, four waves x 128x128, K loop scheduled by hand (gemm_asm.hip, tile id 60)
bool gemm_asm_applicable(int epi, const GemmArgs& a);
bool gemm_asm_has_tile(int epi, const GemmArgs& a, int tile);
int launch_gemm_asm(int epi, const GemmArgs& a, hipStream_t s);
// tile 47 (gemm_asm16.hip): tile 97's geometry on v_mfma_f32_16x16x32; forced only (its own fp32 grouping over K)
bool gemm_asm16_has_tile(int epi, const GemmArgs& a, int tile);
int launch_gemm_asm16(int epi, const GemmArgs& a, hipStream_t s);

// MXFP8 GEMM (gemm_mxfp8.hip): e4m3 operands [rows][K] with one E8M0 scale per 32 elements along K stored
// K-pair-major [K/64][rows_pitch][2] (common.h mx_scale_index; pitches are multiples of 8, and the scale arrays of
// the activations are allocated for M rounded up to 256 rows);
// `g` carries M, N, K, bias, act, the outputs and the residual / LayerNorm fields (g.X, g.W, g.ldx are unused)
struct GemmF8Args {
    GemmArgs g;
    const uint8_t* X8; long ldx8;
    const uint8_t* XS; long xs_rows;
    const uint8_t* W8;              // [N][K]
    const uint8_t* WS; long ws_rows;
    uint8_t* out_scale; long os_rows;  // EPI_MXFP8: scales of the output, pitch os_rows
    uint8_t* qs; uint8_t* ks; uint8_t* vs;   // EPI_QK8: scales of q, k and V^T
};
int launch_gemm_mxfp8(int epi, const GemmF8Args& a, hipStream_t s);   // epi: EPI_MXFP8, EPI_F32, EPI_F32_RESLN, EPI_QK, EPI_V (g.tune_cfg as above)
// the hand-scheduled X3 loop for MXFP8 operands (gemm_asm_f8.hip): which tile it has for this launch (0 = none), and the launch
int gemm_asm_f8_tile(int epi, const GemmF8Args& a);
int launch_gemm_asm_f8(int epi, const GemmF8Args& a, hipStream_t s, int tile);
int launch_mx_quant_rows(const float* in, long ld_in, uint8_t* out, long ld_out, uint8_t* sc, long sc_rows, int R, int K, hipStream_t s);

// fp32 parity-mode GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains), same epilogues on fp32 tensors
enum GemmActF32 { ACTF_NONE = 0, ACTF_GELU_ERF = 1, ACTF_RELU = 2 };
struct GemmArgsF32 {
    const float* X; long ldx;
    const float* W;
    int M, N, K;
    const float* bias; int act;       // GemmActF32 (NOT GemmAct: the parity path has erf-GELU and the conditioner's ReLU)
    float* out0; long ld0;
    const float* res; long ldres;
    int Tp, T; const int* valid;      // feature projection: zero frames t >= min(valid[b], T) (Tp > 0 enables row -> (b,t))
    float* xpad; int xpad_rows;       // optional second copy into the zero-padded pos-conv input
    int tiled;                        // 1: the LDS-DMA 128x128 kernel (forward of the parity mode; K % 32 == 0); 0: the simple 64x64 one
};
int launch_gemm_f32(const GemmArgsF32& a, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// conv frontend layer 0: Conv1d(1->512,k10,s5) + GroupNorm(per (b,c) over time) + GELU, channels-last out
// ------------------------------------------------------------------------------------------------
int launch_conv0_stats(const float* wav, int B, int Lmax, int L0, double* partials, int nchunk, hipStream_t s);
int launch_conv0_finalize(const double* partials, int nchunk, const float* w0, const float* gn_w, const float* gn_b,
                          int B, int L0, float* scale_shift, hipStream_t s);
// out: [B][R0][512] (bf16 or f32); rows l >= L0 are written as zeros
// fmt FMT_SPLIT: erf GELU, hi halves at out, lo halves at out + out_lo (element offset)
int launch_conv0_gn_gelu(const float* wav, int B, int Lmax, int L0, int R0, const float* w0,
                         const float* scale_shift, void* out, int out_f32, hipStream_t s, int fmt = 0, long out_lo = 0, int valu16 = 0);   // valu16: 16-bit modes on the VALU kernel (A/B)

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dim (512 or 768), eps 1e-5, one wave per row
//   in: f32 or bf16 rows (ld_in); optional residual add (f32); outputs: f32 and/or bf16
//   remap: if Tp>0, input row m=(b,t) with t>=T is skipped and outputs are written compact at b*T+t
// ------------------------------------------------------------------------------------------------
struct LnArgs {
    const void* in; int in_bf16; long ld_in;     // in_bf16: the input rows are 16-bit words of format `fmt`
    int fmt;                                      // FMT_BF16 / FMT_F16: format of a 16-bit input and of out_bf16
    int fmt_in;                                   // format of a 16-bit input when it differs from fmt (set = fmt otherwise)
    const float* res; long ld_res;
    const float* gamma; const float* beta;
    float* out_f32; long ld_f32;
    bf16_t* out_bf16; long ld_bf16;
    uint8_t* out_fp8; long ld_fp8;        // optional MXFP8 copy of the output (D = 768): e4m3 rows +
    uint8_t* out_scale; long scale_rows;  //   one E8M0 scale per 32 features, K-pair-major with this row pitch
    float* out_stats;             // optional [M][2] (mean, rstd): lets a later GEMM epilogue re-apply this LayerNorm
    int M, D;
    int Tp, T;        // compaction of the f32 output (final hidden states); 0 = none
    long in_lo, out_lo;   // FMT_SPLIT: element offsets of the lo planes of a 16-bit input / of out_bf16
};
int launch_layernorm(const LnArgs& a, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// flash attention: softmax(q k^T + key mask) v, 12 heads x 64, q pre-scaled by log2(e) / 8 (scores in log2 units)
//   q,k: [B,H,Tp,64] bf16; vt: [B,H,64,Tpv] bf16; ctx out: [B*Tp][768] bf16
// ------------------------------------------------------------------------------------------------
// qw: 0 = automatic, 1 / 2 = 32 / 64 queries per wave
// fmt FMT_SPLIT: q / k / vt / ctx are hi planes, their lo planes lo_qk (q, k), lo_vt and lo_ctx elements further on
int launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* vt, const int* valid, bf16_t* ctx, int B, int T,
                     int Tp, int Tpv, int qw, hipStream_t s, int fmt = 0, long lo_qk = 0, long lo_vt = 0, long lo_ctx = 0);
// the attention core on MXFP8 operands (configs[4]): q8 / k8 [B,H,Tp,64] e4m3 + scales [B,H,Tp,2]; vt8 [B,H,64,Tpv] e4m3 (natural key
// order) + scales [B,H,64,Tpv/32]; ctx bf16 [B*Tp][768], or MXFP8 when ctx_scale is given.  k8 / its scales need one 64-key tile of slack.
int launch_attention_f8(const uint8_t* q8, const uint8_t* qs, const uint8_t* k8, const uint8_t* ks, const uint8_t* v8, const uint8_t* vs,
                        const int* valid, void* ctx, uint8_t* ctx_scale, long scale_rows, int B, int T, int Tp, int Tpv, hipStream_t s);
int launch_attention_f8out(const bf16_t* q, const bf16_t* k, const bf16_t* vt, const int* valid, uint8_t* ctx8, uint8_t* ctx_scale,
                           long scale_rows, int B, int T, int Tp, int Tpv, int qw, hipStream_t s);   // context as MXFP8 (SYLBER_FP8)
int launch_attention_f32(const float* q, const float* k, const float* v, const int* valid, float* ctx, int B, int T,
                         int Tp, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// positional conv embedding: grouped Conv1d(768,768,k=128,pad=64,groups=16)+bias, drop last, GELU, + residual
//   xpad: [B][Tp+128][768] bf16 with 64 zero rows in front and zeros behind the valid frames
//   wpk : packed weights [16 groups][128 taps][64 n (48 used)][48 c] bf16
//   out : f32 [B*Tp][768] = x_f32 + gelu(conv + bias)
// ------------------------------------------------------------------------------------------------
// fmt FMT_SPLIT: xpad / wpk are hi planes with lo planes x_lo / w_lo elements further on (three passes), erf GELU
int launch_posconv(const bf16_t* xpad, const bf16_t* wpk, const float* bias, const float* x_f32, float* out, int B,
                   int Tp, int act, hipStream_t s, int fmt = 0, long x_lo = 0, long w_lo = 0);
int launch_posconv_f32(const float* xpad, const float* w, const float* bias, const float* x_f32, float* out, int B,
                       int Tp, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// segmentation (get_segment + mean-pool), numpy-f32 bit-exact
// ------------------------------------------------------------------------------------------------
// mode 0: wide (frame norms, one workgroup per run of speech frames, compaction, pooling: all CUs); -1: one workgroup per utterance
int launch_segment(const float* hidden, int B, int T, int D, float norm_thr, float merge_thr, int64_t* seg, int* nseg,
                   float* feat, float* scratch, hipStream_t s, int mode = 0);
size_t segment_scratch_floats(int B, int T, int D);

// misc elementwise
int launch_f32_to_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s);
int launch_f32_to_split16(const float* in, bf16_t* out, long lo_off, size_t n, hipStream_t s);   // hi plane at out, lo plane at out + lo_off
int launch_bf16_to_f32_rows(const bf16_t* in, long ld_in, float* out, int B, int Tp, int T, int D, hipStream_t s, int fmt = 0, long in_lo = 0);
