/*
 * sylber_hip_dev.h -- development aids exported by libsylber_hip.so that are NOT part of the drop-in C-ABI
 * (include/sylber_hip.h): micro-benchmarks used by tools/.  Nothing on the product path or in the parity tests
 * needs them (one GPU test uses the workspace-poisoning aid), and they keep no process-global state.
 */
#ifndef SYLBER_HIP_DEV_H
#define SYLBER_HIP_DEV_H
#include <stdint.h>
#include "sylber_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* average ms of one launch of the bf16 GEMM kernel (M x N x K, activation row stride ldx) on pseudo-random
 * operands; epi / act as in csrc/kernels.h; cfg: -1 = automatic tile shape, else persist * 1000 + tile id;
 * cfg in [100, 200): the MXFP8 GEMM with tile configuration cfg - 100; cfg + 200000: every timed launch finds its
 * operands cold (1 GiB written between the launches, each launch timed by its own event pair) */
int sylber_debug_gemm_bench(int32_t M, int32_t N, int32_t K, int32_t ldx, int32_t epi, int32_t act, int32_t cfg,
                            int32_t iters, float* ms_out);
/* the tile id the cost model of the 16-bit GEMM launcher picks for a launch (csrc/gemm_bf16.hip TileModel): host arithmetic only, no device touched.
 * epi / act as the forward passes them (0 = 16-bit out, 3 = q,k,v, 6 = fp32 residual + LayerNorm; act 1 = GELU), fmt 0 bf16 / 1 fp16,
 * model = SYLBER_OPT_GEMM_MODEL, kpat = 1 for the 3-tap conv K order.  tests/test_abi.py pins the choices of the headline shapes. */
int sylber_debug_gemm_pick(int32_t M, int32_t N, int32_t K, int32_t epi, int32_t act, int32_t fmt, int32_t model, int32_t kpat);
/* the same launch through the TRACE instantiation of the 8-wave 256x256 kernel (s_memtime stamps around the phases of its K
 * loop): out20 = 2 x 10 shader-cycle counters of workgroup 0's waves 0 and 4 (csrc/api.hip sylber_debug_gemm_trace) */
int sylber_debug_gemm_trace(int32_t M, int32_t N, int32_t K, int32_t ldx, int32_t epi, int32_t act, unsigned long long* out20,
                            float* ms_out);
/* test aid: fill the handle's activation workspace with `byte` (0xFF = NaN patterns) and force the next forward to redo the zeroing it
 * does after a batch-shape change; results must not change (tests/test_gpu_encoder.py) */
int sylber_debug_poison_workspace(sylber_t h, int32_t byte);
/* average ms of one launch of the attention core (12 heads x 64, B utterances of T frames, no key mask) on pseudo-random packed
 * operands; precision: SYLBER_BF16 (bf16 operands) or SYLBER_FP8 (MXFP8 q / k / V^T, e4m3 P) */
int sylber_debug_attention_bench(int32_t B, int32_t T, int32_t precision, int32_t iters, float* ms_out);
#ifdef __cplusplus
}
#endif
#endif /* SYLBER_HIP_DEV_H */
