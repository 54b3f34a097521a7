/*
 * sylber_hip.h — C-ABI of libsylber_hip.so: the MI355X (gfx950) implementation of the SYLBER
 * Segmenter forward path.
 *
 * The reference has no FFI layer; its de-facto operator boundary is three Python call sites inside
 * Segmenter.__call__ (reference: sylber/model/sylber.py):
 *   (1) sylber.py:122   hidden = self.speech_model(batch, attention_mask=mask).last_hidden_state
 *                       (transformers.HubertModel, 9 layers)           -> sylber_forward()
 *   (2) sylber.py:126   get_segment(states, norm_threshold, merge_threshold)
 *                       (sylber/utils/segment_utils.py:72-131)         -> sylber_segment()
 *   (3) sylber.py:133   states[s:e].mean(0) per segment                -> sylber_segment() (fused)
 * and the weight hand-over at construction, sylber.py:41-54
 *   (HubertModel(config); load_state_dict(strict=False); eval().to(device)) -> sylber_create().
 *
 * Conventions: extern "C", opaque handle, int status (0 = ok, non-zero = error; text via
 * sylber_last_error()), no exceptions and no torch / HIP types in the signatures.  All *_dev
 * pointers are device pointers owned by the caller; `stream` is a hipStream_t passed as void*
 * (NULL = default stream).  Calls are stream-ordered and asynchronous.  A handle is bound to one
 * GPU and is not thread-safe; distinct handles are independent.
 */
#ifndef SYLBER_HIP_H
#define SYLBER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SYLBER_MAX_LAYERS 12
#define SYLBER_HIDDEN 768
#define SYLBER_CONV_DIM 512

typedef struct sylber_ctx* sylber_t;

/* compute precision of the encoder GEMMs */
/* SYLBER_FP8 (BASELINE.json configs[4]): as SYLBER_BF16, but the four projection GEMMs around the attention (q/k/v, out) and the two FFN
 * GEMMs of every encoder layer (all of the encoder's weight GEMMs) run on OCP microscaling FP8 operands (e4m3 elements, one E8M0 power-of-two scale per 32 elements along K) through the
 * block-scaled gfx950 MFMA, fp32 accumulation; tolerance vs the bf16 mode is stated in tests/test_gpu_fp8.py */
/* SYLBER_FP16: exactly the SYLBER_BF16 path (same kernels, same matrix-pipe rate) with IEEE half as the 16-bit operand /
 * activation format: 10 instead of 7 mantissa bits at every hand-over, conversions saturate at +-65504 (measured agreement
 * with the fp32 reference: DESIGN.md) */
/* SYLBER_MIXED16: the conv stack as SYLBER_FP16 (that is where the bf16 mode makes its error: 13 hand-overs of O(1)
 * activations), the encoder as SYLBER_BF16; the feature-projection LayerNorm converts (measured cost / agreement: DESIGN.md) */
/* SYLBER_SPLIT16: every MFMA operand (activations and weights) is a PAIR of IEEE halves, hi = half(x) and lo = half(x - hi)
 * (22 significand bits), and every contraction is three fp16 MFMA passes into one fp32 accumulator (hi.hi + lo.hi + hi.lo);
 * erf GELU, fp32 residual stream / statistics as everywhere.  fp32-grade decisions (segment tables as the fp32 mode's)
 * at about three times the fp16 step instead of the fourteen times of SYLBER_FP32's f32 MFMA. */
enum { SYLBER_BF16 = 0, SYLBER_FP32 = 1, SYLBER_FP8 = 2, SYLBER_FP16 = 3, SYLBER_MIXED16 = 4, SYLBER_SPLIT16 = 5 };

/* HOST pointers to fp32 weights in the layout of HubertModel.state_dict().
 * Replaces the state_dict hand-over at sylber.py:51-54. */
typedef struct {
    const float* q_w; const float* q_b;       /* [768,768],[768] */
    const float* k_w; const float* k_b;
    const float* v_w; const float* v_b;
    const float* o_w; const float* o_b;
    const float* ln1_w; const float* ln1_b;   /* layer_norm */
    const float* ff1_w; const float* ff1_b;   /* intermediate_dense [3072,768],[3072] */
    const float* ff2_w; const float* ff2_b;   /* output_dense [768,3072],[768] */
    const float* ln2_w; const float* ln2_b;   /* final_layer_norm */
} SylberLayerWeights;

typedef struct {
    int32_t num_layers;                        /* encoding_layer, sylber.py:34 (default 9) */
    const float* conv_w[7];                    /* [512,1,10], 4x[512,512,3], 2x[512,512,2] */
    const float* gn_w; const float* gn_b;      /* conv_layers.0.layer_norm (GroupNorm affine) [512] */
    const float* fp_ln_w; const float* fp_ln_b;/* feature_projection.layer_norm [512] */
    const float* fp_w; const float* fp_b;      /* feature_projection.projection [768,512],[768] */
    const float* pos_w;                        /* EFFECTIVE pos-conv weight g*v/||v|| [768,48,128] */
    const float* pos_b;                        /* [768] */
    const float* enc_ln_w; const float* enc_ln_b;
    SylberLayerWeights layers[SYLBER_MAX_LAYERS];
} SylberWeights;

/* number of 50 Hz frames the 7-layer conv stack yields for n_samples (TP:664-677) */
int32_t sylber_num_frames(int32_t n_samples);
/* frame pitch per utterance of the library's internal activation buffers for a batch padded to n_samples (>= the
 * frame count, a multiple of 32); informational: workspace sizing and roofline byte counts */
int32_t sylber_padded_frames(int32_t n_samples);

/* copies + packs the weights onto `device` (bf16 MFMA layouts, conv taps interleaved) */
int sylber_create(const SylberWeights* w, int device, int precision, sylber_t* out);
void sylber_destroy(sylber_t h);
const char* sylber_last_error(void);

/* (1) waveform batch -> last_hidden_state.
 *   wav_dev      [B, Lmax] fp32, rows right-padded with zeros (sylber.py:99-117)
 *   lengths_host [B] valid samples per row (the attention_mask of sylber.py:104-118, run-length
 *                coded), or NULL for "all rows full" (identical numerics to an all-ones mask)
 *   hidden_dev   [B, T, 768] fp32, T = sylber_num_frames(Lmax); padded frames are computed and
 *                returned exactly like the reference does (sylber.py:125-135)
 */
int sylber_forward(sylber_t h, const float* wav_dev, const int32_t* lengths_host, int32_t B, int32_t Lmax,
                   float* hidden_dev, void* stream);
/* lengths_host is read before the call returns (the frame counts travel to the device as kernel arguments: no
 * pageable copy, no host synchronisation).  A handle is single-stream: it owns ONE workspace, so two forwards of the
 * same handle must be ordered on the same stream; use one handle per in-flight batch to overlap batches. */

/* (2)+(3) boundary detection + segment mean-pool, bit-exact w.r.t. the reference's numpy float32 evaluation order.
 * Round 6: frame norms, one workgroup per unbroken run of speech frames (independent instances of get_segment's two phases: a
 * non-speech frame resets the scan, segment_utils.py:84-90), compaction and pooling as four launches that use the whole chip
 * (SYLBER_OPT_SEGMENT = -1: the one-workgroup-per-utterance kernel of rounds 1-5).  Touches no encoder workspace, so it may run on a
 * side stream under the same handle's next sylber_forward; it does use a per-handle scratch slab (frame norms, slot table): the
 * sylber_segment calls of ONE handle must be ordered on one stream.
 *   hidden_dev [B, T, D] fp32 (D = 768 on the product path)
 *   seg_dev    [B, T, 2] int64  (start, end-exclusive) frame indices, first nseg_dev[b] rows valid
 *   nseg_dev   [B] int32
 *   feat_dev   [B, T, D] fp32 mean-pooled features, first nseg_dev[b] rows valid (may be NULL)
 */
int sylber_segment(sylber_t h, const float* hidden_dev, int32_t B, int32_t T, int32_t D, float norm_thr,
                   float merge_thr, int64_t* seg_dev, int32_t* nseg_dev, float* feat_dev, void* stream);

/* ---- file ingest on the device (replaces sylber.py:83-86) -------------------
 *   wav, sr = torchaudio.load(file); if sr != 16000: wav = torchaudio.transforms.Resample(sr, 16000)(wav);
 *   wav = (wav - wav.mean()) / wav.std()
 * pcm_dev      interleaved little-endian PCM frames as they sit in the file's data chunk, on the device
 * sample_width bytes per sample: 1 (uint8), 2 (int16), 3 (int24), 4 (int32), scaled to [-1, 1) like torchaudio.load;
 *              -4 / -8: IEEE float32 / float64 samples (WAVE_FORMAT_IEEE_FLOAT), taken as they are
 * normalize    non-zero: (x - mean) / unbiased std over all channels x frames
 * wav_out_dev  [channels, sylber_ingest_num_frames(frames_in, sr_in)] fp32, every channel one future batch row
 * workspace_dev sylber_ingest_workspace_bytes(sr_in) bytes, 8-byte aligned, owned by the caller
 * Stateless (no handle); stream-ordered.  Resampler: torchaudio's sinc_interp_hann defaults (csrc/ingest.hip). */
int64_t sylber_ingest_num_frames(int64_t frames_in, int32_t sr_in);
int64_t sylber_ingest_workspace_bytes(int32_t sr_in);
int sylber_ingest(const void* pcm_dev, int32_t sample_width, int32_t channels, int64_t frames_in, int32_t sr_in,
                  int32_t normalize, float* wav_out_dev, void* workspace_dev, void* stream);

/* FLAC container -> interleaved integer PCM on the HOST (torchaudio.load at sylber.py:83 decodes .flac files too; the reference's data code falls back from .wav to
 * .flac, sylber/dataset/collective_audio_segment.py:61-67).  A FLAC stream is a serial bit-granular entropy code: host work, like the RIFF header parse; the decoded
 * samples then go through sylber_ingest as 16- or 32-bit PCM scaled to full range.  data = the whole file in memory (an ID3v2 tag in front is skipped).
 * sylber_flac_info: sample rate, channels, bits per sample (4..32), frames (0 = the encoder did not know).  sylber_flac_decode: out_host [capacity_frames][channels] int32 at
 * the samples' own scale, or NULL to verify and count only; every frame's CRC-8 / CRC-16 is checked and, when STREAMINFO carries the encoder's MD5 of the unencoded audio,
 * the decoded PCM against it.  Stateless, no device is touched.  Parity unpinned (no codec / FLAC file in the build image): csrc/flac_host.hip restates RFC 9639. */
int sylber_flac_info(const uint8_t* data_host, int64_t size, int32_t* sample_rate, int32_t* channels, int32_t* bits_per_sample, int64_t* frames);
int sylber_flac_decode(const uint8_t* data_host, int64_t size, int32_t* out_host, int64_t capacity_frames, int64_t* frames_out);

/* ---- the two callers right behind the path, on device-resident outputs of sylber_segment --- */
/* N4: k-means tokenisation, KMQuantizer.get_indices (sylber/model/quantizer.py:86-111; codebook look-up of
 * vector_quantize_pytorch's EuclideanCodebook: argmin_c ||x - c||, first index on ties).
 *   feats_dev [n, D] fp32 (D % 16 == 0), centroids_dev [K, D] fp32, normalize != 0: x / sqrt(sum x^2 + 1e-8) * 6 first
 *   idx_dev [n] int32; workspace_dev: sylber_km_workspace_floats(n, K, D) floats.  Exact-fp32 contraction. */
int64_t sylber_km_workspace_floats(int32_t n, int32_t K, int32_t D);
int sylber_km_assign(const float* feats_dev, int32_t n, const float* centroids_dev, int32_t K, int32_t D, int32_t normalize,
                     int32_t* idx_dev, float* workspace_dev, void* stream);
/* KMQuantizer.decode (quantizer.py:127-133): out[r] = centroids[clip(idx[r], 0)] */
int sylber_km_decode(const int32_t* idx_dev, int32_t n, const float* centroids_dev, int32_t K, int32_t D, float* out_dev, void* stream);

/* N3: front half of SegmentSynthesis.resynthesize (sylber/model/segment_synthesis.py:103-140): segment means broadcast
 * back to their frames -> `MLP` conditioner (Linear -> RFF -> ... -> Linear, segment_synthesis.py:17-53) -> frames with
 * hidden-state norm < norm_thr zeroed.  HOST pointers to fp32 tensors in nn.Linear / nn.LayerNorm layout. */
#define SYLBER_MLP_MAX_HIDDEN 4
typedef struct {
    int32_t input_dim, output_dim, num_hidden;
    int32_t hidden_dims[SYLBER_MLP_MAX_HIDDEN];                 /* each 512 or 768 */
    struct {
        const float *lin_w, *lin_b;                             /* mlp.{2i}: Linear(in, dim) */
        const float *ff1_w, *ff1_b, *ff2_w, *ff2_b, *ln_w, *ln_b; /* mlp.{2i+1}: RFF.linear1 / linear2 / norm */
    } hidden[SYLBER_MLP_MAX_HIDDEN];
    const float *out_w, *out_b;                                 /* mlp.{2 num_hidden}: Linear(dim, output_dim) */
} SylberMlpWeights;
typedef struct sylber_mlp* sylber_mlp_t;
int sylber_mlp_create(const SylberMlpWeights* w, int device, sylber_mlp_t* out);
void sylber_mlp_destroy(sylber_mlp_t m);
int64_t sylber_condition_workspace_floats(sylber_mlp_t m, int32_t B, int32_t S);
/* hidden_dev [B,T,D], seg_dev [B,T,2], nseg_dev [B], feat_dev [B,T,D]: the buffers of sylber_forward / sylber_segment;
 * S: segment slots per utterance to run through the MLP (max nseg <= S <= T);
 * avg_hidden_dev [B,T,D] (nullable): averaged_target_hidden_states; cond_dev [B,T,output_dim]: the conditioning input */
int sylber_condition(sylber_mlp_t m, const float* hidden_dev, const int64_t* seg_dev, const int32_t* nseg_dev, const float* feat_dev,
                     int32_t B, int32_t T, int32_t S, float norm_thr, float* avg_hidden_dev, float* cond_dev, float* workspace_dev,
                     void* stream);

/* ---- per-handle options ------------------------------------------------------------------------ */
/* Tuning / test overrides, scoped to ONE handle (nothing process-global).  SYLBER_OPT_GEMM_TILE: value < 0 restores the
 * automatic choice (0 is a tile id); the other keys: 0 = automatic.
 *   SYLBER_OPT_GEMM_TILE               tile configuration id of the bf16 GEMM launches (csrc/gemm_bf16.hip launch_t:
 *                                      0 = 256x128, 3 = 128x128, 4 = 128x192, 10 = 256x256 8-wave, 11 = 256x192 8-wave;
 *                                      hand-scheduled K loops (csrc/gemm_asm.hip; a launch whose epilogue / K has no such
 *                                      instantiation falls back to 128x192): 80 = 256x256 4-wave, 90 = 256x192, 95 = 256x256
 *                                      8-wave, 85 / 91 / 97 = the same with a three-slot X ring, 51 / 57 = 91 / 97 on 192-row tiles, 86 = that ring on a 256x128 tile / four waves (91 also requests the fp32 residual rows of out-proj / FFN2 from
 *                                      inside its K loop when the launch has whole tiles; 96 = 91 without that), 60 = the 64-byte-row first cut;
 *                                      round 6: 5 / 6 = 128x128 / 128x192 on EIGHT waves (same bits; measured no gain for these kernels, forced only);
 *                                      the v_mfma_f32_16x16x32 family (csrc/gemm_asm16.hip; 16-bit-output and plain fp32 epilogues): 47 = tile 97's geometry, 46 = its 192-row
 *                                      sibling, 13 / 14 = 128x128 / 128x192 hipcc-scheduled on four waves, 15 / 16 = the same on eight waves.  On a 16-bit-output launch
 *                                      (conv1-5, FFN1) ANY forced id runs on the family member of the same shape class unless SYLBER_OPT_GEMM_MFMA16 = -1)
 *   SYLBER_OPT_ATTN_QUERIES_PER_WAVE   0 (default): the hand-scheduled key loop (csrc/attention.hip attention_asm_kernel, generated by
 *                                      tools/gen_attn_asm.py; bf16 / fp16 modes); 32 or 64: the compiler-scheduled kernels with that many
 *                                      queries per wave (its reference; split16 always runs the 32-query one)
 *   SYLBER_OPT_GEMM_PERSISTENT         0 (automatic): GEMM launches of more than one round run as persistent workgroups
 *                                      walking the tile list (4-wave kernels: two per CU; the 256x256 kernel: one per CU
 *                                      with cross-tile operand prefetch); k > 0: k workgroups per CU for the 4-wave
 *                                      kernels; k < 0: one workgroup per tile everywhere (A/B switch)
 *   SYLBER_OPT_FUSE_OUTPROJ_LN         1: the attention out-projection and the LayerNorm behind it run as ONE launch on
 *                                      full-row tiles (csrc/gemm_rowln.hip; bit-identical outputs) where the shape allows;
 *                                      0 / -1 (default): GEMM launch + LayerNorm launch (faster with two batches in flight)
 *   SYLBER_OPT_CONV0_VALU              1: conv layer 0 of the 16-bit modes on the VALU kernel instead of the matrix-pipe one (A/B switch)
 *   SYLBER_OPT_RESLN_PREFETCH          the K loops of the attention out-projection and FFN2 request the rows of the fp32 residual stream
 *                                      they update while they run (csrc/gemm_asm.hip, bit-identical outputs): 1..3 = fragment columns
 *                                      (of 3 per wave) prefetched, -1 = none (the epilogue loads them), 0 = the default (1: measured
 *                                      best with two batches in flight; 3 is best with one, profiles/r04_resln_prefetch.md)
 *   SYLBER_OPT_FP8_ATTENTION           SYLBER_FP8 only: 1 (and 0 = the default) = the attention core on MXFP8 operands too -- the q / k / v projection
 *                                      quantises its outputs (e4m3, one power-of-two scale per 32 features of q / k and per 32 keys of v),
 *                                      P is e4m3: BASELINE configs[4] as worded, whatever the batch shape (round 5: the q / k / v launch pads its rows
 *                                      up to whole 256-row tiles and does not store the padding); -1 = always the bf16 core (q, k, v, P in bf16)
 *   SYLBER_OPT_GEMM_TAIL               k > 0: a GEMM launch whose tile count is not a whole number of rounds of 256 persistent workgroups is split BY ROWS --
 *                                      the rows of the full rounds on the chosen tile, the remaining rows as a second launch on tile id k (bit-identical
 *                                      outputs: every element is one fp32 chain over K whatever tile computes it).  0 / -1 (default): one launch.  Measured
 *                                      not to pay (a small tile alone on a CU is no faster than a big one; csrc/gemm_bf16.hip launch_f): the partial
 *                                      rounds are handled by the 192-row tiles below; this stays as a test vehicle
 *   SYLBER_OPT_GEMM_H192               0 (default): the cost model may pick the 192-row siblings of the hand-scheduled tiles (ids 51 = 192x192, 57 = 192x256:
 *                                      a second tile HEIGHT, for launches whose 256-row tile count leaves a partial last round); -1: 256-row tiles only (A/B)
 *   SYLBER_OPT_GEMM_MFMA16             0 (default): the GEMMs with 16-bit outputs (conv1-5, FFN1: csrc/gemm_bf16.hip EPI_BF16) run on the v_mfma_f32_16x16x32 family of
 *                                      kernels (csrc/gemm_asm16.hip, tile ids 13 / 14 / 46 / 47: the instruction shape that is cheapest per FLOP under the package power
 *                                      cap, profiles/r06_mfma16_loop.md); a forced SYLBER_OPT_GEMM_TILE id is mapped to the member of the same shape class on these
 *                                      launches.  -1: those launches on the 32x32x16 kernels of the earlier rounds (A/B switch).  The two families group an output
 *                                      element's fp32 sum over K differently (32-k against 16-k blocks): results agree to fp32 rounding of the accumulator, not bit
 *                                      for bit -- WITHIN either setting results do not depend on the batch shape or the tile.
 *   SYLBER_OPT_GEMM_MODEL              which tiles the GEMM launches of this handle get (csrc/gemm_bf16.hip launch_f).  0 (default): the handle OWNS the chip -- one batch in
 *                                      flight, as in a synchronous Segmenter.__call__: a launch is charged its partial last round (measured per kernel
 *                                      family) and may use the 192-row tiles; 5: the handle SHARES the chip with another in-flight batch (bench.py's
 *                                      pipeline, ShardedSegmenter's two engines): the CUs a partial round leaves idle go to the other stream's
 *                                      kernels, whole-round accounting with the round-5 constants measured fastest there (8 x 60 s: 8.57 vs 8.78 ms
 *                                      per step; alone on the chip the same handle is 5 % slower with it).  Outputs are bit-identical either way.
 *                                      (6 = 0 without the lone-round rule of round 6 -- a launch of fewer tiles than CUs on the one-per-CU persistent tiles costs a full round --
 *                                      i.e. the tile choice of rounds 5-6a for small batches: A/B switch.)
 *   SYLBER_OPT_SEGMENT                 0 (default): sylber_segment as wide kernels -- frame norms, one workgroup per unbroken run of speech frames
 *                                      (independent instances of get_segment's two phases), compaction, pooling one wave per segment, all on all
 *                                      CUs; -1: one workgroup per utterance (rounds 1-5; bit-identical, A/B switch and reference)
 *   SYLBER_OPT_FP16_AUDIT              1: (re)start the fp16 headroom audit -- from now on every forward of a SYLBER_FP16 / SYLBER_MIXED16 handle scans each
 *                                      16-bit activation buffer right behind its producer and accumulates, per stage, the number of values AT the
 *                                      format's saturation value (+-65504: the fp16 modes clamp on conversion, they never produce infinities) and the
 *                                      largest magnitude seen; read with sylber_get_fp16_audit.  0 (default): off, nothing is launched. */
enum { SYLBER_OPT_GEMM_TILE = 1, SYLBER_OPT_ATTN_QUERIES_PER_WAVE = 2, SYLBER_OPT_GEMM_PERSISTENT = 3, SYLBER_OPT_FUSE_OUTPROJ_LN = 4,
       SYLBER_OPT_CONV0_VALU = 5, SYLBER_OPT_RESLN_PREFETCH = 6, SYLBER_OPT_FP8_ATTENTION = 7, SYLBER_OPT_GEMM_TAIL = 8, SYLBER_OPT_SEGMENT = 9, SYLBER_OPT_FP16_AUDIT = 10, SYLBER_OPT_GEMM_H192 = 11, SYLBER_OPT_GEMM_MODEL = 12, SYLBER_OPT_GEMM_MFMA16 = 13 };
int sylber_set_option(sylber_t h, int32_t key, int32_t value);
/* the audit's counters since SYLBER_OPT_FP16_AUDIT was last set (synchronises the device): names[i] (static strings: conv0 .. conv6, ln512,
 * proj_xpad, layernorm, q, k, v, context, ffn1), saturated[i] values clamped at +-65504, max_abs[i] largest magnitude; returns the number of
 * stages written (<= cap), 0 when the audit never ran, -1 on error.  A non-zero `saturated` means this checkpoint / input does not fit
 * IEEE half at that stage: use SYLBER_BF16 (8 exponent bits) or SYLBER_SPLIT16. */
int sylber_get_fp16_audit(sylber_t h, const char** names, uint32_t* saturated, float* max_abs, int32_t cap);

/* the `features is not None` branch of resynthesize (segment_synthesis.py:135-140): features_dev [rows, input_dim] frame
 * features supplied by the caller (e.g. decoded unit embeddings) -> cond_dev [rows, output_dim] = MLP(features), rows with
 * ((f**2).sum(-1))**.5 < 1e-4 zeroed (no 1e-8 under the root, threshold fixed at 1e-4, :136-137);
 * workspace_dev: sylber_condition_workspace_floats(m, rows, 1) floats */
int sylber_condition_features(sylber_mlp_t m, const float* features_dev, int32_t rows, float* cond_dev, float* workspace_dev,
                              void* stream);

/* ---- introspection used by parity tests and the benchmark ------------------------------------ */
/* run sylber_forward only up to a stage: 0 = all, 1 = conv stack, 2 = +projection/pos-conv/LN,
 * 3 + l = through encoder layer l.  The stage output is written to hidden_dev in place of the final
 * hidden states: stage 1 -> [B,T,512] conv features, otherwise [B,T,768]. */
int sylber_set_stop_stage(sylber_t h, int32_t stage);
/* per-kernel device time of the last forward, measured with HIP events on the launch stream.
 * names/ms arrays of capacity cap; returns the number of entries (<=cap) or <0 on error. */
int sylber_set_profiling(sylber_t h, int32_t enable);
int sylber_get_profile(sylber_t h, const char** names, float* ms, int32_t cap);
/* graph mode (bf16 / fp8 handles): the second sylber_forward with the same (B, Lmax, wav_dev, hidden_dev) on a
 * non-default stream captures its ~110 launches into a hipGraph and later calls replay it — for launch-bound small
 * batches (one 3 s utterance: 1.5 ms eager).  Up to 8 shapes are cached per handle; disabling frees them. */
int sylber_set_graph_mode(sylber_t h, int32_t enable);
/* bytes of device workspace currently held by the handle */
int64_t sylber_workspace_bytes(sylber_t h);

/* ---- single-op entry points (unit parity tests; same kernels the forward path launches) ------ */
/* C[M,N] (fp32) = A[M,K] (fp32, cast to bf16) x W[N,K]^T (fp32, cast to bf16) + bias[N] (nullable); act: 0 none,
 * 1 gelu (the bf16 path's polynomial), 2 gelu (erf); precision: SYLBER_BF16 or SYLBER_FP8 (K % 128 == 0);
 * tile: -1 = automatic, else the tile configuration id to run (parity tests sweep every configuration), + 1000 k for a
 * persistent launch of k x 256 workgroups (9000 + id: never persistent) */
int sylber_op_linear(const float* a_dev, const float* w_dev, const float* bias_dev, float* c_dev, int32_t M,
                     int32_t N, int32_t K, int32_t act, int32_t precision, int32_t tile, void* stream);
/* the same contraction through the 16-bit output epilogue the conv layers and FFN1 use: c16_dev [M,N] bf16 words */
int sylber_op_linear16(const float* a_dev, const float* w_dev, const float* bias_dev, uint16_t* c16_dev, int32_t M,
                       int32_t N, int32_t K, int32_t act, int32_t precision, int32_t tile, void* stream);
/* one 3-tap stride-2 layer of the conv feature extractor (transformers modeling_hubert.py TP:160-175: Conv1d(512, 512, 3, stride 2,
 * bias=False) + GELU) as the 16-bit forward runs it: implicit GEMM over channels-last rows, chunk-major K order, 16-bit out.
 * x_dev [R, 512] fp32 rows (R >= 2 M + 1), w_host [512 out][512 in][3] fp32 in HOST memory (torch layout), y16_dev [M, 512] bf16 words */
int sylber_op_conv3(const float* x_dev, const float* w_host, uint16_t* y16_dev, int32_t R, int32_t M, int32_t tile, void* stream);
/* the residual GEMM of an encoder block (attention out-projection, FFN2; transformers modeling_hubert.py TP:361-397 reached from
 * sylber/model/sylber.py:122), in place: pre[M,N] (fp32) <- A[M,K] W[N,K]^T + bias + LayerNorm(pre), the LayerNorm re-derived per
 * element as ((pre - mean) * rstd) * gamma + beta from stats[M,2] = (mean, rstd) and gamma / beta [N] -- the form the forward
 * uses, where the previous LayerNorm launch stores only its 16-bit output and the row statistics; bf16 operands, tile as above */
int sylber_op_linear_resln(const float* a_dev, const float* w_dev, const float* bias_dev, float* pre_dev, const float* stats_dev,
                           const float* gamma_dev, const float* beta_dev, int32_t M, int32_t N, int32_t K, int32_t tile, void* stream);
/* MXFP8 quantiser used by SYLBER_FP8: x [R,K] fp32 -> data [R,K] e4m3 + E8M0 scales, one per 32 elements along K,
 * stored K-pair-major [K/64, R, 2] (K % 64 == 0; the layout the GEMM's scale fetch wants); the block scale is the
 * smallest power of two 2^e with amax <= 448 * 2^e, elements are x / 2^e rounded to nearest even */
int sylber_op_mx_quantize(const float* x_dev, int32_t R, int32_t K, uint8_t* data_dev, uint8_t* scale_dev, void* stream);
/* y = LayerNorm(x [+ res]) over the last dim D (512 or 768), eps 1e-5 */
int sylber_op_layernorm(const float* x_dev, const float* res_dev, const float* g_dev, const float* b_dev,
                        float* y_dev, int32_t M, int32_t D, void* stream);
/* softmax(q k^T / 8 + key mask) v ; q,k,v,o: [B,T,768] fp32 (12 heads x 64); valid_dev [B] int32;
 * queries_per_wave: 0 = automatic, 32 or 64; precision: SYLBER_BF16 (bf16 q, k, v, P) or SYLBER_FP8 (q, k, v quantised to MXFP8
 * -- e4m3 with one power-of-two scale per 32 features of q / k and per 32 keys of v -- and P to e4m3: BASELINE configs[4]) */
int sylber_op_attention(const float* q_dev, const float* k_dev, const float* v_dev, const int32_t* valid_dev,
                        float* o_dev, int32_t B, int32_t T, int32_t precision, int32_t queries_per_wave, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SYLBER_HIP_H */
