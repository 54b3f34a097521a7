// C-ABI of libsylber_hip.so (see include/sylber_hip.h): handle, weight packing, workspace, and the
// launch sequence of the Segmenter forward path
//   sylber/model/sylber.py:122  speech_model(batch, attention_mask).last_hidden_state
//   sylber/model/sylber.py:126  get_segment(...)            sylber.py:133  segment mean-pool
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/sylber_hip.h"
#include "../../include/sylber_hip_dev.h"
#include "kernels.h"

static thread_local char g_err[512] = "";
void syl_set_error(const char* what, const char* detail) { snprintf(g_err, sizeof(g_err), "%s: %s", what, detail); }
extern "C" const char* sylber_last_error(void) { return g_err; }

// every entry point runs on the handle's GPU and leaves the caller's current device as it found it (a process may
// hold handles on several GPUs; sylber_destroy runs from garbage collectors at arbitrary times)
struct DeviceGuard {
    int prev = -1; bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { int cur = -1; if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev); }
};
#define GUARD_DEVICE(dev)                                                                             \
    DeviceGuard _dg(dev);                                                                             \
    if (!_dg.ok) { syl_set_error("hipSetDevice", "cannot select the handle's device"); return 1; }

static const int CK[7] = {10, 3, 3, 3, 3, 2, 2};
static const int CS[7] = {5, 2, 2, 2, 2, 2, 2};

extern "C" int32_t sylber_num_frames(int32_t n) {
    for (int i = 0; i < 7; ++i) n = (n - CK[i]) / CS[i] + 1;
    return n;
}

// frame pitch per utterance of the internal activation buffers: enough rows for every conv layer's valid outputs at
// its 2^(6-i) rows-per-frame pitch, rounded up to 32 (a 32-row MFMA block then never straddles two utterances, and
// 32 x 10 s = 16384 rows is a whole number of 256-row tiles)
static int padded_frames(int Lmax) {
    int n = Lmax, tp = 0;
    for (int i = 0; i < 7; ++i) {
        n = (n - CK[i]) / CS[i] + 1;
        const int f = 1 << (6 - i);
        const int need = (n + f - 1) / f;
        tp = need > tp ? need : tp;
    }
    return (tp + 31) & ~31;
}
extern "C" int32_t sylber_padded_frames(int32_t n_samples) { return n_samples < 400 ? 0 : padded_frames(n_samples); }

struct LayerDev {
    bf16_t *wqkv, *wo, *w1, *w2;
    uint8_t *w1q = nullptr, *w1s = nullptr, *w2q = nullptr, *w2s = nullptr;   // SYLBER_FP8: MXFP8 FFN weights + E8M0 scales
    uint8_t *wqkvq = nullptr, *wqkvs = nullptr;                                // and the fused q/k/v projection
    uint8_t *woq = nullptr, *wos = nullptr;                                    // and the attention out-projection
    float *bqkv, *bo, *b1, *b2, *ln1w, *ln1b, *ln2w, *ln2b;
};

struct ProfEntry { std::string name; hipEvent_t e0, e1; };
struct GraphEntry { int B, Lmax, stop_stage; const void* in; void* out; hipGraphExec_t exec; unsigned long long stamp; };

struct sylber_ctx {
    int device = 0, precision = 0, num_layers = 9;
    int fmt = FMT_BF16;           // 16-bit operand format of the MFMA path (encoder)
    int fmt_conv = FMT_BF16;      // ... of the conv stack (differs from fmt only for SYLBER_MIXED16)
    // weights
    char* wbase = nullptr; size_t wbytes = 0;
    char* f8base = nullptr; size_t f8bytes = 0;
    float *conv0_w, *gn_w, *gn_b, *fp_ln_w, *fp_ln_b, *fp_b, *pos_b, *enc_ln_w, *enc_ln_b;
    bf16_t* conv_w[7];
    bf16_t *fp_w, *pos_w;
    LayerDev L[SYLBER_MAX_LAYERS];
    // fp32 parity mode: the same tensors kept in fp32
    float* conv_w32[7]; float *fp_w32, *pos_w32;
    struct { float *wqkv, *wo, *w1, *w2; } L32[SYLBER_MAX_LAYERS];
    // workspace
    char* ws = nullptr; size_t ws_bytes = 0;
    int ws_B = 0, ws_Lmax = 0;
    float* seg_scratch = nullptr; size_t seg_scratch_floats = 0;
    int stop_stage = 0;
    int opt_gemm_cfg = 0, opt_attn_qw = 0, opt_gemm_persist = 0;   // sylber_set_option (0 = automatic)
    int opt_fuse_ln = 0;                                           // out-projection + LayerNorm in one launch: 0 auto, 1 always, -1 never
    int opt_conv0_valu = 0;                                        // 1: conv0 of the 16-bit modes on the VALU kernel (A/B switch)
    // fp16 headroom audit (SYLBER_OPT_FP16_AUDIT): per stage, how many 16-bit activations sit AT the saturation value and the largest magnitude
    int opt_audit16 = 0;
    unsigned* audit_dev = nullptr;                                 // [AUDIT_STAGES][2]: saturated count, max |x| as half bits
    int opt_segment = 0;                                           // boundary detection: 0 wide (all CUs), -1 one workgroup per utterance
    int opt_gemm_model = 0;                                        // 5: round-5 tile cost model (A/B switch)
    int opt_gemm_mfma16 = 0;                                       // -1: the 16-bit-output GEMMs on the 32x32x16 kernels (A/B switch; GemmArgs::tune_mfma16)
    int opt_gemm_h192 = 0;                                         // -1: no 192-row tiles in the cost model (A/B switch)
    int opt_gemm_tail = 0;                                         // row split of multi-round GEMM launches: 0 auto, -1 never, k + 1 = tail tile id k
    int opt_attn8 = 0;                                             // SYLBER_FP8: attention core on MXFP8 operands (0 / 1 on, -1 off)
    int opt_resln_pre = 0;                                         // residual prefetch of the out-proj / FFN2 K loops: 0 default, -1 off, 1..3 columns
    bool graph_mode = false;
    std::vector<GraphEntry> graphs; unsigned long long graph_clock = 0;
    // profiling
    int profiling = 0;
    std::vector<ProfEntry> prof;
    std::vector<hipEvent_t> ev_pool;
    std::vector<std::string> prof_names; std::vector<float> prof_ms;
};

// ------------------------------------------------------------------------------------------------
struct Packer {
    std::vector<char> host;
    size_t add(size_t bytes) { size_t off = (host.size() + 255) & ~(size_t)255; host.resize(off + bytes); return off; }
    size_t add_f32(const float* src, size_t n) { size_t o = add(n * 4); memcpy(host.data() + o, src, n * 4); return o; }
    int fmt = FMT_BF16;           // 16-bit format of the MFMA operands (FMT_F16 for SYLBER_FP16)
    size_t add_bf16(const float* src, size_t n) {
        if (fmt == FMT_SPLIT) {
            // two half planes: hi = half(w) at [0, n), lo = half(w - hi) at [n, 2n)
            size_t o = add(n * 4);
            bf16_t* d = (bf16_t*)(host.data() + o);
            for (size_t i = 0; i < n; ++i) { d[i] = f2h_host(src[i]); d[n + i] = f2h_host(src[i] - h2f_host(d[i])); }
            return o;
        }
        size_t o = add(n * 2);
        bf16_t* d = (bf16_t*)(host.data() + o);
        if (fmt == FMT_F16) for (size_t i = 0; i < n; ++i) d[i] = f2h_host(src[i]);
        else for (size_t i = 0; i < n; ++i) d[i] = f2bf(src[i]);
        return o;
    }
};

extern "C" int sylber_create(const SylberWeights* w, int device, int precision, sylber_t* out) {
    if (!w || !out) { syl_set_error("sylber_create", "null argument"); return 1; }
    if (w->num_layers < 1 || w->num_layers > SYLBER_MAX_LAYERS) { syl_set_error("sylber_create", "num_layers out of range"); return 1; }
    if (precision != SYLBER_BF16 && precision != SYLBER_FP32 && precision != SYLBER_FP8 && precision != SYLBER_FP16 && precision != SYLBER_MIXED16 && precision != SYLBER_SPLIT16) { syl_set_error("sylber_create", "unknown precision"); return 1; }
    const bool f32 = precision == SYLBER_FP32;
    GUARD_DEVICE(device);
    sylber_ctx* c = new sylber_ctx();
    c->device = device; c->precision = precision; c->num_layers = w->num_layers;
    c->fmt = precision == SYLBER_FP16 ? FMT_F16 : (precision == SYLBER_SPLIT16 ? FMT_SPLIT : FMT_BF16);
    c->fmt_conv = (precision == SYLBER_FP16 || precision == SYLBER_MIXED16) ? FMT_F16 : (precision == SYLBER_SPLIT16 ? FMT_SPLIT : FMT_BF16);
    Packer P;
    P.fmt = c->fmt_conv;                                 // conv weights first
    size_t o_conv0 = P.add_f32(w->conv_w[0], 512 * 10);
    size_t o_gnw = P.add_f32(w->gn_w, 512), o_gnb = P.add_f32(w->gn_b, 512);
    size_t o_conv[7] = {0};
    for (int i = 1; i < 7; ++i) {
        // [o][c][j] -> [o][j*512 + c]: one output position's receptive field is then ONE contiguous
        // run of taps*512 channels-last activations (implicit GEMM with ldx = stride*512)
        // The 3-tap layers of the 16-bit modes are packed in the chunk-major K order every kernel walks them in (GemmArgs::kpat,
        // common.h tap3_offset): position o of a weight row holds the input element whose operand-row offset is tap3_offset(o)
        const int k = CK[i];
        const bool chunk_major = k == 3 && !f32 && c->fmt_conv != FMT_SPLIT;
        std::vector<float> tmp((size_t)512 * k * 512);
        for (int o = 0; o < 512; ++o)
            for (int cc = 0; cc < 512; ++cc)
                for (int j = 0; j < k; ++j) tmp[((size_t)o * k + j) * 512 + cc] = w->conv_w[i][((size_t)o * 512 + cc) * k + j];
        if (chunk_major) {
            std::vector<float> t2(tmp.size());
            for (int o = 0; o < 512; ++o)
                for (int pos = 0; pos < 1536; ++pos)
                    t2[(size_t)o * 1536 + pos] = tmp[(size_t)o * 1536 + tap3_offset(pos * 2) / 2];
            tmp.swap(t2);
        }
        o_conv[i] = f32 ? P.add_f32(tmp.data(), tmp.size()) : P.add_bf16(tmp.data(), tmp.size());
    }
    P.fmt = c->fmt;                                      // everything after the conv stack
    size_t o_fplw = P.add_f32(w->fp_ln_w, 512), o_fplb = P.add_f32(w->fp_ln_b, 512);
    size_t o_fpw = f32 ? P.add_f32(w->fp_w, 768 * 512) : P.add_bf16(w->fp_w, 768 * 512), o_fpb = P.add_f32(w->fp_b, 768);
    size_t o_posw;
    {
        // [768 = g*48+n][48 c][128 tap] -> [g][tap][64 n][56 c] (zero padded), 112-byte rows
        std::vector<float> tmp((size_t)16 * 128 * 64 * 56, 0.f);
        for (int g = 0; g < 16; ++g)
            for (int n = 0; n < 48; ++n)
                for (int cc = 0; cc < 48; ++cc)
                    for (int t = 0; t < 128; ++t)
                        tmp[(((size_t)g * 128 + t) * 64 + n) * 56 + cc] = w->pos_w[(((size_t)g * 48 + n) * 48 + cc) * 128 + t];
        if (!f32) o_posw = P.add_bf16(tmp.data(), tmp.size());
        else {
            // fp32 parity kernel: [g][tap][64 n][52 c] fp32, zero padded (208-byte rows, see fp32_path.hip)
            std::vector<float> t32((size_t)16 * 128 * 64 * 52, 0.f);
            for (int g = 0; g < 16; ++g)
                for (int n = 0; n < 48; ++n)
                    for (int cc = 0; cc < 48; ++cc)
                        for (int t = 0; t < 128; ++t)
                            t32[(((size_t)g * 128 + t) * 64 + n) * 52 + cc] = w->pos_w[(((size_t)g * 48 + n) * 48 + cc) * 128 + t];
            o_posw = P.add_f32(t32.data(), t32.size());
        }
    }
    size_t o_posb = P.add_f32(w->pos_b, 768);
    size_t o_elw = P.add_f32(w->enc_ln_w, 768), o_elb = P.add_f32(w->enc_ln_b, 768);
    struct LOff { size_t wqkv, bqkv, wo, bo, l1w, l1b, w1, b1, w2, b2, l2w, l2b; } lo[SYLBER_MAX_LAYERS];
    for (int l = 0; l < w->num_layers; ++l) {
        const SylberLayerWeights& lw = w->layers[l];
        std::vector<float> qkv((size_t)2304 * 768), bq(2304);
        memcpy(qkv.data(), lw.q_w, 768 * 768 * 4);
        memcpy(qkv.data() + 768 * 768, lw.k_w, 768 * 768 * 4);
        memcpy(qkv.data() + 2 * 768 * 768, lw.v_w, 768 * 768 * 4);
        memcpy(bq.data(), lw.q_b, 768 * 4); memcpy(bq.data() + 768, lw.k_b, 768 * 4); memcpy(bq.data() + 1536, lw.v_b, 768 * 4);
        lo[l].wqkv = f32 ? P.add_f32(qkv.data(), qkv.size()) : P.add_bf16(qkv.data(), qkv.size()); lo[l].bqkv = P.add_f32(bq.data(), 2304);
        lo[l].wo = f32 ? P.add_f32(lw.o_w, 768 * 768) : P.add_bf16(lw.o_w, 768 * 768); lo[l].bo = P.add_f32(lw.o_b, 768);
        lo[l].l1w = P.add_f32(lw.ln1_w, 768); lo[l].l1b = P.add_f32(lw.ln1_b, 768);
        lo[l].w1 = f32 ? P.add_f32(lw.ff1_w, (size_t)3072 * 768) : P.add_bf16(lw.ff1_w, (size_t)3072 * 768); lo[l].b1 = P.add_f32(lw.ff1_b, 3072);
        lo[l].w2 = f32 ? P.add_f32(lw.ff2_w, (size_t)768 * 3072) : P.add_bf16(lw.ff2_w, (size_t)768 * 3072); lo[l].b2 = P.add_f32(lw.ff2_b, 768);
        lo[l].l2w = P.add_f32(lw.ln2_w, 768); lo[l].l2b = P.add_f32(lw.ln2_b, 768);
    }
    c->wbytes = P.host.size();
    if (hipMalloc((void**)&c->wbase, c->wbytes) != hipSuccess) { delete c; syl_set_error("sylber_create", "hipMalloc(weights) failed"); return 1; }
    if (hipMemcpy(c->wbase, P.host.data(), c->wbytes, hipMemcpyHostToDevice) != hipSuccess) {
        hipFree(c->wbase); delete c; syl_set_error("sylber_create", "weight upload failed"); return 1;
    }
    char* b = c->wbase;
    c->conv0_w = (float*)(b + o_conv0); c->gn_w = (float*)(b + o_gnw); c->gn_b = (float*)(b + o_gnb);
    c->conv_w[0] = nullptr;
    for (int i = 1; i < 7; ++i) { c->conv_w[i] = (bf16_t*)(b + o_conv[i]); c->conv_w32[i] = (float*)(b + o_conv[i]); }
    c->fp_w32 = (float*)(b + o_fpw); c->pos_w32 = (float*)(b + o_posw);
    c->fp_ln_w = (float*)(b + o_fplw); c->fp_ln_b = (float*)(b + o_fplb);
    c->fp_w = (bf16_t*)(b + o_fpw); c->fp_b = (float*)(b + o_fpb);
    c->pos_w = (bf16_t*)(b + o_posw); c->pos_b = (float*)(b + o_posb);
    c->enc_ln_w = (float*)(b + o_elw); c->enc_ln_b = (float*)(b + o_elb);
    for (int l = 0; l < w->num_layers; ++l) {
        LayerDev& d = c->L[l];
        d.wqkv = (bf16_t*)(b + lo[l].wqkv); d.bqkv = (float*)(b + lo[l].bqkv);
        d.wo = (bf16_t*)(b + lo[l].wo); d.bo = (float*)(b + lo[l].bo);
        d.ln1w = (float*)(b + lo[l].l1w); d.ln1b = (float*)(b + lo[l].l1b);
        d.w1 = (bf16_t*)(b + lo[l].w1); d.b1 = (float*)(b + lo[l].b1);
        d.w2 = (bf16_t*)(b + lo[l].w2); d.b2 = (float*)(b + lo[l].b2);
        d.ln2w = (float*)(b + lo[l].l2w); d.ln2b = (float*)(b + lo[l].l2b);
        c->L32[l].wqkv = (float*)(b + lo[l].wqkv); c->L32[l].wo = (float*)(b + lo[l].wo);
        c->L32[l].w1 = (float*)(b + lo[l].w1); c->L32[l].w2 = (float*)(b + lo[l].w2);
    }
    if (precision == SYLBER_FP8) {
        // FFN weights once more as MXFP8 (e4m3 + one E8M0 scale per 32 input features), quantised on the device
        // from the fp32 originals with the same kernel the activations' op-level entry point uses
        const size_t per_layer = (size_t)3072 * 768 + (size_t)3072 * 24 + (size_t)768 * 3072 + (size_t)768 * 96 +
                                 (size_t)2304 * 768 + (size_t)2304 * 24 + (size_t)768 * 768 + (size_t)768 * 24;
        c->f8bytes = per_layer * w->num_layers;
        float* tmp = nullptr;
        if (hipMalloc((void**)&c->f8base, c->f8bytes) != hipSuccess || hipMalloc((void**)&tmp, (size_t)3072 * 768 * 4) != hipSuccess) {
            if (c->f8base) hipFree(c->f8base);
            hipFree(c->wbase); delete c; syl_set_error("sylber_create", "hipMalloc(fp8 weights) failed"); return 1;
        }
        int bad = 0;
        for (int l = 0; l < w->num_layers && !bad; ++l) {
            LayerDev& d = c->L[l];
            uint8_t* q = (uint8_t*)c->f8base + per_layer * l;
            d.w1q = q; d.w1s = d.w1q + (size_t)3072 * 768; d.w2q = d.w1s + (size_t)3072 * 24; d.w2s = d.w2q + (size_t)768 * 3072;
            d.wqkvq = d.w2s + (size_t)768 * 96; d.wqkvs = d.wqkvq + (size_t)2304 * 768;
            d.woq = d.wqkvs + (size_t)2304 * 24; d.wos = d.woq + (size_t)768 * 768;
            bad |= hipMemcpy(tmp, w->layers[l].o_w, (size_t)768 * 768 * 4, hipMemcpyHostToDevice) != hipSuccess;
            bad |= launch_mx_quant_rows(tmp, 768, d.woq, 768, d.wos, 768, 768, 768, nullptr);
            bad |= hipDeviceSynchronize() != hipSuccess;
            {
                const SylberLayerWeights& lw = w->layers[l];
                bad |= hipMemcpy(tmp, lw.q_w, (size_t)768 * 768 * 4, hipMemcpyHostToDevice) != hipSuccess;
                bad |= hipMemcpy(tmp + (size_t)768 * 768, lw.k_w, (size_t)768 * 768 * 4, hipMemcpyHostToDevice) != hipSuccess;
                bad |= hipMemcpy(tmp + (size_t)2 * 768 * 768, lw.v_w, (size_t)768 * 768 * 4, hipMemcpyHostToDevice) != hipSuccess;
                bad |= launch_mx_quant_rows(tmp, 768, d.wqkvq, 768, d.wqkvs, 2304, 2304, 768, nullptr);
                bad |= hipDeviceSynchronize() != hipSuccess;
            }
            bad |= hipMemcpy(tmp, w->layers[l].ff1_w, (size_t)3072 * 768 * 4, hipMemcpyHostToDevice) != hipSuccess;
            bad |= launch_mx_quant_rows(tmp, 768, d.w1q, 768, d.w1s, 3072, 3072, 768, nullptr);
            bad |= hipDeviceSynchronize() != hipSuccess;
            bad |= hipMemcpy(tmp, w->layers[l].ff2_w, (size_t)768 * 3072 * 4, hipMemcpyHostToDevice) != hipSuccess;
            bad |= launch_mx_quant_rows(tmp, 3072, d.w2q, 3072, d.w2s, 768, 768, 3072, nullptr);
            bad |= hipDeviceSynchronize() != hipSuccess;
        }
        hipFree(tmp);
        if (bad) { hipFree(c->f8base); hipFree(c->wbase); delete c; syl_set_error("sylber_create", "fp8 weight quantisation failed"); return 1; }
    }
    *out = c;
    return 0;
}

extern "C" void sylber_destroy(sylber_t c) {
    if (!c) return;
    DeviceGuard dg(c->device);
    if (c->wbase) hipFree(c->wbase);
    if (c->f8base) hipFree(c->f8base);
    if (c->ws) hipFree(c->ws);
    if (c->seg_scratch) hipFree(c->seg_scratch);
    if (c->audit_dev) hipFree(c->audit_dev);
    for (auto& g : c->graphs) if (g.exec) hipGraphExecDestroy(g.exec);
    for (auto e : c->ev_pool) hipEventDestroy(e);
    delete c;
}

// stages of the fp16 headroom audit (audit16 below)
enum { AUD_CONV0 = 0, AUD_CONV6 = 6, AUD_LN512 = 7, AUD_XPAD = 8, AUD_LN = 9, AUD_Q = 10, AUD_K = 11, AUD_V = 12, AUD_CTX = 13, AUD_FFN1 = 14, AUDIT_STAGES = 15 };
extern "C" int sylber_set_stop_stage(sylber_t c, int32_t stage) { if (!c) return 1; c->stop_stage = stage; return 0; }
extern "C" int sylber_set_option(sylber_t c, int32_t key, int32_t value) {
    if (!c) { syl_set_error("sylber_set_option", "null handle"); return 1; }
    switch (key) {
        case SYLBER_OPT_GEMM_TILE: c->opt_gemm_cfg = value < 0 ? 0 : value + 1; break;     // stored as id + 1, 0 = automatic
        case SYLBER_OPT_ATTN_QUERIES_PER_WAVE: c->opt_attn_qw = value == 32 ? 1 : (value == 64 ? 2 : 0); break;
        case SYLBER_OPT_GEMM_PERSISTENT: c->opt_gemm_persist = value; break;      // < 0: also keep the 256x256 kernel one tile per workgroup
        case SYLBER_OPT_FUSE_OUTPROJ_LN: c->opt_fuse_ln = value > 0 ? 1 : (value < 0 ? -1 : 0); break;
        case SYLBER_OPT_CONV0_VALU: c->opt_conv0_valu = value > 0 ? (value == 2 ? 2 : 1) : 0; break;     // (2: producer-only timing probe, experiments build)
        case SYLBER_OPT_RESLN_PREFETCH: c->opt_resln_pre = value < 0 ? -1 : (value <= 3 ? value : 0); break;
        case SYLBER_OPT_FP8_ATTENTION: c->opt_attn8 = value < 0 ? -1 : (value > 0 ? 1 : 0); break;
        case SYLBER_OPT_FP16_AUDIT: {
            GUARD_DEVICE(c->device);
            c->opt_audit16 = value > 0 ? 1 : 0;
            if (c->opt_audit16) {                                 // (re)start the audit from zero
                if (!c->audit_dev) HIP_TRY(hipMalloc((void**)&c->audit_dev, AUDIT_STAGES * 2 * sizeof(unsigned)));
                HIP_TRY(hipDeviceSynchronize());
                HIP_TRY(hipMemset(c->audit_dev, 0, AUDIT_STAGES * 2 * sizeof(unsigned)));
            }
            break;
        }
        case SYLBER_OPT_SEGMENT: c->opt_segment = value < 0 ? -1 : 0; break;
        case SYLBER_OPT_GEMM_MODEL: c->opt_gemm_model = value == 5 ? 5 : (value == 2 ? 2 : (value == 6 ? 6 : 0)); break;   // (6 = 0 without the lone-round rule: A/B)
        case SYLBER_OPT_GEMM_H192: c->opt_gemm_h192 = value < 0 ? -1 : 0; break;
        case SYLBER_OPT_GEMM_MFMA16: c->opt_gemm_mfma16 = value < 0 ? -1 : 0; break;
        case SYLBER_OPT_GEMM_TAIL: c->opt_gemm_tail = value < 0 ? -1 : (value > 0 ? value + 1 : 0); break;   // k > 0: tail tile id k (stored id + 1)
        default: syl_set_error("sylber_set_option", "unknown option key"); return 1;
    }
    if (c->graph_mode) { for (auto& g : c->graphs) if (g.exec) hipGraphExecDestroy(g.exec); c->graphs.clear(); }   // captured launches are stale
    return 0;
}
// enabling (or re-enabling) profiling resets the accumulated per-kernel times
extern "C" int sylber_set_profiling(sylber_t c, int32_t enable) {
    if (!c) return 1;
    const char* nm[1]; float ms[1];
    sylber_get_profile(c, nm, ms, 0);   // retire pending events
    c->prof_names.clear(); c->prof_ms.clear();
    c->profiling = enable;
    return 0;
}
extern "C" int64_t sylber_workspace_bytes(sylber_t c) { return c ? (int64_t)(c->ws_bytes + c->seg_scratch_floats * 4 + c->wbytes + c->f8bytes) : 0; }

// ------------------------------------------------------------------------------------------------
struct Plan {
    int B, Lmax, L[7], T, Tp, Tpv, R[7];
    size_t o_bufA, o_bufB, o_ln512, o_xf32, o_xpad, o_pre, o_stats, o_hbf16, o_q, o_k, o_vt, o_ctx, o_ffn, o_part, o_ss, o_valid,
        total;
    int nchunk;
    bool zero_all = false;        // fp32 parity plan: its own offsets, zero everything on a layout change
    // SYLBER_SPLIT16: every 16-bit buffer holds two half planes; element offsets of the lo planes (0 otherwise)
    long lo_bufA = 0, lo_bufB = 0, lo_ln512 = 0, lo_xpad = 0, lo_hbf = 0, lo_qk = 0, lo_vt = 0, lo_ctx = 0, lo_ffn = 0;
};

static void make_plan(int B, int Lmax, Plan& p, int planes = 1) {
    p.B = B; p.Lmax = Lmax;
    int n = Lmax;
    for (int i = 0; i < 7; ++i) { n = (n - CK[i]) / CS[i] + 1; p.L[i] = n; }
    p.T = p.L[6];
    p.Tp = padded_frames(Lmax);
    p.Tpv = (p.Tp + 63) & ~63;
    for (int i = 0; i < 7; ++i) p.R[i] = p.Tp << (6 - i);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t M = (size_t)B * p.Tp;
    const size_t P = (size_t)planes;
    p.o_bufA = take(((size_t)B * p.R[0] + 8) * 512 * 2 * P);
    p.o_bufB = take(((size_t)B * p.R[1] + 8) * 512 * 2 * P);
    p.o_ln512 = take(M * 512 * 2 * P);
    p.o_xf32 = take(M * 768 * 4);
    p.o_xpad = take((size_t)B * (p.Tp + 128) * 768 * 2 * P);
    p.o_pre = take(M * 768 * 4);
    p.o_stats = take(M * 2 * 4);
    p.o_hbf16 = take((M + 128) * 768 * 2 * P);
    if (planes == 2) {
        p.lo_bufA = (long)(((size_t)B * p.R[0] + 8) * 512); p.lo_bufB = (long)(((size_t)B * p.R[1] + 8) * 512);
        p.lo_ln512 = (long)(M * 512); p.lo_xpad = (long)((size_t)B * (p.Tp + 128) * 768); p.lo_hbf = (long)((M + 128) * 768);
        p.lo_qk = (long)(M * 768); p.lo_vt = (long)((size_t)B * 12 * 64 * p.Tpv); p.lo_ctx = (long)((M + 128) * 768);
        p.lo_ffn = (long)((M + 128) * 3072);
        p.zero_all = true;        // pad regions of both planes: zero the workspace on a layout change
    }
    // q, k, V^T and the attention context are dead by the time FFN1 writes its intermediate, and that is dead before the
    // next layer's q/k/v projection: the FFN intermediate ALIASES them, which keeps a layer's working set
    // (residual stream + bf16 copy + this region + weights = ~190 MB at 32 x 10 s) inside the 256 MB Infinity Cache
    const size_t attn_begin = off;
    p.o_q = take(M * 768 * 2 * P);
    p.o_k = take(M * 768 * 2 * P);
    p.o_vt = take((size_t)B * 12 * 64 * p.Tpv * 2 * P);
    p.o_ctx = take((M + 128) * 768 * 2 * P);
    p.o_ffn = attn_begin;
    {
        const size_t need = (M + 128) * 3072 * 2 * P;
        if (off - attn_begin < need) take(need - (off - attn_begin));
    }
    p.nchunk = (p.L[0] + 2047) / 2048;
    p.o_part = take((size_t)B * p.nchunk * 65 * 8);
    p.o_ss = take((size_t)B * 512 * 2 * 4);
    p.o_valid = take((size_t)B * 4);
    p.total = off;
}

static int ensure_workspace(sylber_ctx* c, const Plan& p, hipStream_t s) {
    if (p.total > c->ws_bytes) {
        HIP_TRY(hipStreamSynchronize(s));
        if (c->ws) HIP_TRY(hipFree(c->ws));
        c->ws = nullptr; c->ws_bytes = 0;
        HIP_TRY(hipMalloc((void**)&c->ws, p.total));
        c->ws_bytes = p.total;
        c->ws_B = 0;
    }
    if (c->ws_B != p.B || c->ws_Lmax != p.Lmax) {
        // layout changed: the regions that are read but never (fully) written must read as zeros -- the pos-conv
        // input's halo rows, the V^T key tail [Tp, Tpv) and the slack rows behind the GEMM operands (which only ever
        // feed rows beyond M, but must stay finite).  Everything else is written before it is read, so a ragged
        // serving loop (new Lmax per call) pays ~60 MB of memset instead of the whole 2 GB workspace.
        const size_t M = (size_t)p.B * p.Tp;
        if (p.zero_all) HIP_TRY(hipMemsetAsync(c->ws, 0, p.total, s));
        else {
            HIP_TRY(hipMemsetAsync(c->ws + p.o_xpad, 0, (size_t)p.B * (p.Tp + 128) * 768 * 2, s));
            HIP_TRY(hipMemsetAsync(c->ws + p.o_vt, 0, (size_t)p.B * 12 * 64 * p.Tpv * 2, s));
            HIP_TRY(hipMemsetAsync(c->ws + p.o_bufA + (size_t)p.B * p.R[0] * 512 * 2, 0, 8 * 512 * 2, s));
            HIP_TRY(hipMemsetAsync(c->ws + p.o_bufB + (size_t)p.B * p.R[1] * 512 * 2, 0, 8 * 512 * 2, s));
            HIP_TRY(hipMemsetAsync(c->ws + p.o_hbf16 + M * 768 * 2, 0, 128 * 768 * 2, s));
            HIP_TRY(hipMemsetAsync(c->ws + p.o_ctx + M * 768 * 2, 0, 128 * 768 * 2, s));
            HIP_TRY(hipMemsetAsync(c->ws + p.o_ffn + M * 3072 * 2, 0, (size_t)128 * 3072 * 2, s));
        }
        c->ws_B = p.B; c->ws_Lmax = p.Lmax;
    }
    return 0;
}

// valid frames per utterance -> device, without a host staging buffer: the values travel as kernel arguments (64 per
// launch), so there is no pageable hipMemcpyAsync (an implicit host synchronisation) and nothing whose lifetime
// the caller has to think about
struct ValidPack { int v[64]; };
__global__ void set_valid_kernel(int* __restrict__ dst, ValidPack p, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = p.v[threadIdx.x];
}
static int upload_valid(int* valid_dev, const int32_t* lengths_host, int B, int Lmax, hipStream_t s) {
    for (int b0 = 0; b0 < B; b0 += 64) {
        ValidPack pk;
        const int n = B - b0 < 64 ? B - b0 : 64;
        for (int i = 0; i < n; ++i) {
            const int len = lengths_host ? lengths_host[b0 + i] : Lmax;
            if (len > Lmax || len < 400) { syl_set_error("sylber_forward", "lengths must be in [400, Lmax]"); return 1; }
            pk.v[i] = sylber_num_frames(len);
        }
        hipLaunchKernelGGL(set_valid_kernel, dim3(1), dim3(64), 0, s, valid_dev + b0, pk, n);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- fp16 headroom audit --------------------------------------------------------------------------------------------
// IEEE half tops out at 65504; the fp16 modes SATURATE on conversion (H16<FMT_F16>::sat) instead of producing infinities, so a
// checkpoint whose activations outgrow the format is clamped silently.  With SYLBER_OPT_FP16_AUDIT on, every producer of a 16-bit
// activation buffer is followed by a scan of what it wrote: values AT +-65504 (0x7bff) are counted as saturated, the largest magnitude is
// kept as the stage's headroom figure.  Off (default): nothing is launched.  Synthetic weights have never come near the limit; real ones
// have never been seen by this library (the checkpoint is not obtainable offline) -- this is how a user finds out.
static const char* const AUDIT_NAMES[AUDIT_STAGES] = {"conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "ln512", "proj_xpad", "layernorm",
                                                      "q", "k", "v", "context", "ffn1"};
__global__ __launch_bounds__(256) void audit16_kernel(const unsigned short* __restrict__ buf, long rows, long cols, long pitch, unsigned* __restrict__ slot) {
    unsigned sat = 0, mx = 0;
    const long n = rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const unsigned a = buf[(i / cols) * pitch + i % cols] & 0x7fffu;
        if (a <= 0x7c00u) mx = a > mx ? a : mx;             // (NaN patterns are not magnitudes)
        sat += a == 0x7bffu;
    }
    for (int d = 32; d >= 1; d >>= 1) { sat += __shfl_xor(sat, d, 64); const unsigned o = __shfl_xor(mx, d, 64); mx = o > mx ? o : mx; }
    if ((threadIdx.x & 63) == 0) { if (sat) atomicAdd(slot, sat); atomicMax(slot + 1, mx); }
}
static int audit16(sylber_ctx* c, int stage, const void* buf, long rows, long cols, long pitch, hipStream_t s) {
    if (!c->opt_audit16 || !c->audit_dev) return 0;
    long blocks = (rows * cols + 256L * 16 - 1) / (256L * 16);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(audit16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned short*)buf, rows, cols, pitch, c->audit_dev + 2 * stage);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int sylber_get_fp16_audit(sylber_t c, const char** names, uint32_t* saturated, float* max_abs, int32_t cap) {
    if (!c) { syl_set_error("sylber_get_fp16_audit", "null handle"); return -1; }
    if (!c->audit_dev) return 0;
    GUARD_DEVICE(c->device);
    unsigned host[AUDIT_STAGES * 2];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host, c->audit_dev, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) {
        syl_set_error("sylber_get_fp16_audit", "device read failed"); return -1;
    }
    const int n = cap < AUDIT_STAGES ? cap : AUDIT_STAGES;
    for (int i = 0; i < n; ++i) { names[i] = AUDIT_NAMES[i]; saturated[i] = host[2 * i]; max_abs[i] = h2f_host((bf16_t)host[2 * i + 1]); }
    return n;
}

struct ProfScope {
    sylber_ctx* c; hipStream_t s; bool on;
    ProfScope(sylber_ctx* c_, hipStream_t s_, const char* name) : c(c_), s(s_), on(c_->profiling != 0) {
        if (!on) return;
        ProfEntry e; e.name = name;
        auto get = [&]() { hipEvent_t ev; if (c->ev_pool.empty()) { hipEventCreate(&ev); } else { ev = c->ev_pool.back(); c->ev_pool.pop_back(); } return ev; };
        e.e0 = get(); e.e1 = get();
        hipEventRecord(e.e0, s);
        c->prof.push_back(e);
    }
    ~ProfScope() { if (on) hipEventRecord(c->prof.back().e1, s); }
};

extern "C" int sylber_get_profile(sylber_t c, const char** names, float* ms, int32_t cap) {
    if (!c) return -1;
    if (!c->prof.empty()) {
        c->prof_names.clear(); c->prof_ms.clear();
        for (auto& e : c->prof) {
            hipEventSynchronize(e.e1);
            float t = 0.f;
            hipEventElapsedTime(&t, e.e0, e.e1);
            size_t i = 0;
            for (; i < c->prof_names.size(); ++i) if (c->prof_names[i] == e.name) break;
            if (i == c->prof_names.size()) { c->prof_names.push_back(e.name); c->prof_ms.push_back(0.f); }
            c->prof_ms[i] += t;
            c->ev_pool.push_back(e.e0); c->ev_pool.push_back(e.e1);
        }
        c->prof.clear();
    }
    int n = (int)c->prof_names.size();
    n = n < cap ? n : cap;
    for (int i = 0; i < n; ++i) { names[i] = c->prof_names[i].c_str(); ms[i] = c->prof_ms[i]; }
    return n;
}

#define RUN(name, call)                         \
    do {                                        \
        ProfScope _ps(c, s, name);              \
        if ((call) != 0) return 1;              \
    } while (0)

static int forward_f32(sylber_ctx* c, const float* wav_dev, const int32_t* lengths_host, int B, int Lmax, float* hidden_dev,
                       hipStream_t s);

// every kernel launch of the bf16 / fp8 forward, in stream order; nothing else (no allocation, copy or synchronisation),
// so the sequence can be replayed from a captured hipGraph
static int forward_launch(sylber_ctx* c, const Plan& p, const float* wav_dev, float* hidden_dev, hipStream_t s) {
    const int B = p.B, Lmax = p.Lmax;
    char* w = c->ws;
    bf16_t* bufA = (bf16_t*)(w + p.o_bufA); bf16_t* bufB = (bf16_t*)(w + p.o_bufB);
    bf16_t* ln512 = (bf16_t*)(w + p.o_ln512);
    float* xf32 = (float*)(w + p.o_xf32); bf16_t* xpad = (bf16_t*)(w + p.o_xpad);
    float* pre = (float*)(w + p.o_pre); float* stats = (float*)(w + p.o_stats); bf16_t* hbf = (bf16_t*)(w + p.o_hbf16);
    bf16_t* q = (bf16_t*)(w + p.o_q); bf16_t* k = (bf16_t*)(w + p.o_k); bf16_t* vt = (bf16_t*)(w + p.o_vt);
    bf16_t* ctx = (bf16_t*)(w + p.o_ctx); bf16_t* ffn = (bf16_t*)(w + p.o_ffn);
    double* part = (double*)(w + p.o_part); float* ss = (float*)(w + p.o_ss); int* valid = (int*)(w + p.o_valid);
    const int M = B * p.Tp;

    // ---- conv layer 0 + GroupNorm + GELU
    RUN("conv0_stats", launch_conv0_stats(wav_dev, B, Lmax, p.L[0], part, p.nchunk, s));
    RUN("conv0_finalize", launch_conv0_finalize(part, p.nchunk, c->conv0_w, c->gn_w, c->gn_b, B, p.L[0], ss, s));
    const bool split = c->precision == SYLBER_SPLIT16;      // hi / lo half planes, erf GELU (fp32-grade decisions)
    RUN("conv0_gn_gelu", launch_conv0_gn_gelu(wav_dev, B, Lmax, p.L[0], p.R[0], c->conv0_w, ss, bufA, 0, s, c->fmt_conv, p.lo_bufA, c->opt_conv0_valu));
    const bool aud_c = c->opt_audit16 && c->fmt_conv == FMT_F16, aud_e = c->opt_audit16 && c->fmt == FMT_F16;   // (the audit launches are never captured: graph mode is refused with it)
    if (aud_c && audit16(c, AUD_CONV0, bufA, (long)B * p.R[0], 512, 512, s)) return 1;
    // ---- conv layers 1..6 as implicit GEMM (ping-pong)
    bf16_t* src = bufA; bf16_t* dst = bufB;
    long src_lo = p.lo_bufA, dst_lo = p.lo_bufB;
    for (int i = 1; i < 7; ++i) {
        GemmArgs a = {};
        a.X = src; a.ldx = (long)CS[i] * 512; a.W = c->conv_w[i];
        a.M = B * p.R[i]; a.N = 512; a.K = CK[i] * 512; a.bias = nullptr; a.act = split ? ACT_GELU_ERF7 : ACT_GELU_FAST;
        a.out0 = dst; a.ld0 = 512; a.tune_cfg = c->opt_gemm_cfg; a.tune_persist = c->opt_gemm_persist; a.tune_tail = c->opt_gemm_tail; a.tune_h192 = c->opt_gemm_h192; a.tune_mfma16 = c->opt_gemm_mfma16; a.tune_model = c->opt_gemm_model;  a.fmt = c->fmt_conv;
        a.x_lo = src_lo; a.w_lo = (long)512 * CK[i] * 512; a.out_lo = dst_lo;
        a.kpat = (CK[i] == 3 && c->fmt_conv != FMT_SPLIT) ? 1 : 0;      // chunk-major K order (weights packed to match at create)
        static const char* nm[7] = {"", "gemm_conv1", "gemm_conv2", "gemm_conv3", "gemm_conv4", "gemm_conv5", "gemm_conv6"};
        RUN(nm[i], launch_gemm_bf16(EPI_BF16, a, s));
        if (aud_c && audit16(c, AUD_CONV0 + i, dst, (long)B * p.R[i], 512, 512, s)) return 1;
        bf16_t* t = src; src = dst; dst = t;
        const long tl = src_lo; src_lo = dst_lo; dst_lo = tl;
    }
    bf16_t* feats = src;   // [B*Tp][512]
    const long feats_lo = src_lo;
    if (c->stop_stage == 1) {
        RUN("copy_out", launch_bf16_to_f32_rows(feats, 512, hidden_dev, B, p.Tp, p.T, 512, s, c->fmt_conv, feats_lo));
        return 0;
    }
    // ---- feature projection: LN(512) -> Linear(512->768), zero padded frames
    {
        LnArgs a = {};
        a.in = feats; a.in_bf16 = 1; a.ld_in = 512; a.gamma = c->fp_ln_w; a.beta = c->fp_ln_b;
        a.out_bf16 = ln512; a.ld_bf16 = 512; a.M = M; a.D = 512; a.fmt = c->fmt; a.fmt_in = c->fmt_conv;
        a.in_lo = feats_lo; a.out_lo = p.lo_ln512;
        RUN("ln512", launch_layernorm(a, s));
        GemmArgs g = {};
        g.X = ln512; g.ldx = 512; g.W = c->fp_w; g.M = M; g.N = 768; g.K = 512; g.bias = c->fp_b;
        g.out0 = xf32; g.ld0 = 768; g.out1 = xpad; g.Tp = p.Tp; g.T = p.T; g.valid = valid; g.xpad_rows = p.Tp + 128; g.fmt = c->fmt;
        g.x_lo = p.lo_ln512; g.w_lo = (long)768 * 512; g.out_lo = p.lo_xpad;
        RUN("gemm_proj", launch_gemm_bf16(EPI_PROJ, g, s));
        if (aud_e && (audit16(c, AUD_LN512, ln512, M, 512, 512, s) || audit16(c, AUD_XPAD, xpad, (long)B * (p.Tp + 128), 768, 768, s))) return 1;
    }
    // ---- positional conv + residual, encoder LayerNorm
    RUN("posconv", launch_posconv(xpad, c->pos_w, c->pos_b, xf32, pre, B, p.Tp, split ? 2 : 1, s, c->fmt, p.lo_xpad, (long)16 * 128 * 64 * 56));
    // SYLBER_FP8: the FFN runs on MXFP8 operands; the LayerNorm in front of it then emits e4m3 + E8M0 block scales
    // instead of bf16 (into the same buffer), and FFN1 leaves its GELU output as MXFP8 for FFN2
    const bool f8 = c->precision == SYLBER_FP8;
    const long Mp = ((long)M + 255) & ~255L;            // row pitch of the activations' scale arrays
    uint8_t* h8 = (uint8_t*)hbf; uint8_t* h8s = h8 + (((size_t)M * 768 + 255) & ~(size_t)255);       // 24 Mp bytes
    uint8_t* ctx8 = (uint8_t*)ctx; uint8_t* ctx8s = ctx8 + (((size_t)M * 768 + 255) & ~(size_t)255);     // 24 Mp bytes
    uint8_t* ffn8 = (uint8_t*)ffn; uint8_t* ffn8s = ffn8 + (((size_t)M * 3072 + 255) & ~(size_t)255);  // 96 Mp bytes
    auto run_ln = [&](const float* gam, const float* bet, bool last, bool to_fp8 = false) -> int {
        LnArgs a = {};
        a.in = pre; a.in_bf16 = 0; a.ld_in = 768; a.gamma = gam; a.beta = bet; a.M = M; a.D = 768; a.fmt = c->fmt; a.fmt_in = c->fmt;
        if (last) { a.out_f32 = hidden_dev; a.ld_f32 = 768; a.Tp = p.Tp; a.T = p.T; }
        else if (to_fp8) { a.out_fp8 = h8; a.ld_fp8 = 768; a.out_scale = h8s; a.scale_rows = Mp; a.out_stats = stats; }
        else { a.out_bf16 = hbf; a.ld_bf16 = 768; a.out_stats = stats; a.out_lo = p.lo_hbf; }   // no fp32 copy: see EPI_F32_RESLN
        if (launch_layernorm(a, s)) return 1;
        if (aud_e && !last && !to_fp8) return audit16(c, AUD_LN, hbf, M, 768, 768, s);
        return 0;
    };
    RUN("layernorm", run_ln(c->enc_ln_w, c->enc_ln_b, c->stop_stage == 2, f8));
    if (c->stop_stage == 2) return 0;
    // the residual of every block is the previous LayerNorm's output; it is re-derived in the GEMM epilogue from
    // the pre-LN sum still sitting in `pre` (updated in place) + that LayerNorm's row statistics and affine
    const float* res_g = c->enc_ln_w; const float* res_b = c->enc_ln_b;
    // ---- encoder layers (post-LN)
    for (int l = 0; l < c->num_layers; ++l) {
        const LayerDev& d = c->L[l];
        const bool last = (l == c->num_layers - 1) || (c->stop_stage == 3 + l);
        bool fused_ln1 = false;
        // one launch for q, k and v (N = 2304): the q / k thirds leave head-major, the v third transposed (EPI_QK)
        // SYLBER_FP8, attention core on MXFP8 operands: the q / k / V^T regions hold e4m3 bytes in their first halves and the block
        // scales behind them (k: one 64-key tile of slack in between, read by the last key tile of an utterance and masked)
        uint8_t* q8 = (uint8_t*)q; uint8_t* k8 = (uint8_t*)k; uint8_t* v8 = (uint8_t*)vt;
        uint8_t* q8s = q8 + (size_t)M * 768; uint8_t* k8s = k8 + (size_t)M * 768 + 4096; uint8_t* v8s = v8 + (size_t)B * 768 * p.Tpv;
        bool attn8 = false;
        if (f8) {
            GemmF8Args g = {};
            g.g.M = M; g.g.N = 2304; g.g.K = 768; g.g.bias = d.bqkv; g.g.out0 = q; g.g.out1 = k; g.g.out2 = vt;
            g.g.Tp = p.Tp; g.g.Tpv = p.Tpv; g.g.T = p.T;
            g.X8 = h8; g.ldx8 = 768; g.XS = h8s; g.xs_rows = Mp; g.W8 = d.wqkvq; g.WS = d.wqkvs; g.ws_rows = 2304;
            // the attention core of the mode must not depend on the batch shape (one utterance, same hidden states alone or in a
            // batch): the q / k / v launch always runs on whole 256-row tiles -- M is padded up, the rows beyond the batch read
            // whatever follows the operand (inside the workspace: the buffer holds (M + 128) x 768 x 2 bytes) and are not stored
            GemmF8Args gq = g;
            gq.g.M = (int)Mp; gq.g.M_store = M;
            attn8 = c->opt_attn8 >= 0 && gemm_asm_f8_tile(EPI_QK8, gq) != 0;
            if (attn8) {
                gq.g.out0 = q8; gq.g.out1 = k8; gq.g.out2 = v8; gq.qs = q8s; gq.ks = k8s; gq.vs = v8s;
                RUN("gemm_qkv", launch_gemm_mxfp8(EPI_QK8, gq, s));
            } else
            RUN("gemm_qkv", launch_gemm_mxfp8(EPI_QK, g, s));
        } else {
            GemmArgs g = {};
            g.X = hbf; g.ldx = 768; g.W = d.wqkv; g.M = M; g.N = 2304; g.K = 768; g.bias = d.bqkv;
            g.out0 = q; g.out1 = k; g.out2 = vt; g.Tp = p.Tp; g.Tpv = p.Tpv; g.T = p.T;
            g.tune_cfg = c->opt_gemm_cfg; g.tune_persist = c->opt_gemm_persist; g.tune_tail = c->opt_gemm_tail; g.tune_h192 = c->opt_gemm_h192; g.tune_model = c->opt_gemm_model;  g.fmt = c->fmt;
            g.x_lo = p.lo_hbf; g.w_lo = (long)2304 * 768; g.out_lo = p.lo_qk; g.out2_lo = p.lo_vt;
            RUN("gemm_qkv", launch_gemm_bf16(EPI_QK, g, s));
            if (aud_e && (audit16(c, AUD_Q, q, M, 768, 768, s) || audit16(c, AUD_K, k, M, 768, 768, s) ||
                          audit16(c, AUD_V, vt, (long)B * 768, p.Tp, p.Tpv, s))) return 1;
        }
        if (f8) {
            if (attn8) RUN("attention", launch_attention_f8(q8, q8s, k8, k8s, v8, v8s, valid, ctx8, ctx8s, Mp, B, p.T, p.Tp, p.Tpv, s));
            else
            RUN("attention", launch_attention_f8out(q, k, vt, valid, ctx8, ctx8s, Mp, B, p.T, p.Tp, p.Tpv, c->opt_attn_qw, s));
            GemmF8Args o = {};
            o.g.M = M; o.g.N = 768; o.g.K = 768; o.g.bias = d.bo; o.g.out0 = pre; o.g.ld0 = 768; o.g.res = pre; o.g.ldres = 768;
            o.g.ln_stats = stats; o.g.ln_gamma = res_g; o.g.ln_beta = res_b;
            o.X8 = ctx8; o.ldx8 = 768; o.XS = ctx8s; o.xs_rows = Mp; o.W8 = d.woq; o.WS = d.wos; o.ws_rows = 768;
            RUN("gemm_out", launch_gemm_mxfp8(EPI_F32_RESLN, o, s));
        } else {
        RUN("attention", launch_attention(q, k, vt, valid, ctx, B, p.T, p.Tp, p.Tpv, c->opt_attn_qw, s, c->fmt, p.lo_qk, p.lo_vt, p.lo_ctx));
        if (aud_e && audit16(c, AUD_CTX, ctx, M, 768, 768, s)) return 1;
        GemmArgs o = {};
        o.X = ctx; o.ldx = 768; o.W = d.wo; o.M = M; o.N = 768; o.K = 768; o.bias = d.bo;
        o.out0 = pre; o.ld0 = 768; o.res = pre; o.ldres = 768; o.ln_stats = stats; o.ln_gamma = res_g; o.ln_beta = res_b;
        o.tune_cfg = c->opt_gemm_cfg; o.tune_persist = c->opt_gemm_persist; o.tune_tail = c->opt_gemm_tail; o.tune_h192 = c->opt_gemm_h192; o.tune_model = c->opt_gemm_model;  o.fmt = c->fmt; o.tune_pre = c->opt_resln_pre;
        o.x_lo = p.lo_ctx; o.w_lo = (long)768 * 768;
        // out-projection + LayerNorm 1 as ONE launch on full-row tiles (gemm_rowln.hip) where the batch fills the chip
        o.out1 = hbf; o.ln_stats_out = stats; o.ln_gamma_out = d.ln1w; o.ln_beta_out = d.ln1b;
        fused_ln1 = !split && c->opt_fuse_ln > 0 && gemm_rowln_applicable(o);      // measured: faster as a pair, slower with two batches in flight
        if (fused_ln1) RUN("gemm_out_ln", launch_gemm_rowln(o, s));
        else RUN("gemm_out", launch_gemm_bf16(EPI_F32_RESLN, o, s));
        }
        if (!fused_ln1) RUN("layernorm", run_ln(d.ln1w, d.ln1b, false, f8));
        if (f8) {
            GemmF8Args f1 = {};
            f1.g.M = M; f1.g.N = 3072; f1.g.K = 768; f1.g.bias = d.b1; f1.g.act = 1; f1.g.out0 = ffn8; f1.g.ld0 = 3072;
            f1.X8 = h8; f1.ldx8 = 768; f1.XS = h8s; f1.xs_rows = Mp; f1.W8 = d.w1q; f1.WS = d.w1s; f1.ws_rows = 3072; f1.out_scale = ffn8s; f1.os_rows = Mp;
            RUN("gemm_ffn1", launch_gemm_mxfp8(EPI_MXFP8, f1, s));
            GemmF8Args f2 = {};
            f2.g.M = M; f2.g.N = 768; f2.g.K = 3072; f2.g.bias = d.b2; f2.g.out0 = pre; f2.g.ld0 = 768; f2.g.res = pre; f2.g.ldres = 768;
            f2.g.ln_stats = stats; f2.g.ln_gamma = d.ln1w; f2.g.ln_beta = d.ln1b;
            f2.X8 = ffn8; f2.ldx8 = 3072; f2.XS = ffn8s; f2.xs_rows = Mp; f2.W8 = d.w2q; f2.WS = d.w2s; f2.ws_rows = 768;
            RUN("gemm_ffn2", launch_gemm_mxfp8(EPI_F32_RESLN, f2, s));
        } else {
        GemmArgs f1 = {};
        f1.X = hbf; f1.ldx = 768; f1.W = d.w1; f1.M = M; f1.N = 3072; f1.K = 768; f1.bias = d.b1; f1.act = split ? ACT_GELU_ERF7 : ACT_GELU_FAST;
        f1.out0 = ffn; f1.ld0 = 3072; f1.tune_cfg = c->opt_gemm_cfg; f1.tune_persist = c->opt_gemm_persist; f1.tune_tail = c->opt_gemm_tail; f1.tune_h192 = c->opt_gemm_h192; f1.tune_mfma16 = c->opt_gemm_mfma16; f1.tune_model = c->opt_gemm_model;  f1.fmt = c->fmt;
        f1.x_lo = p.lo_hbf; f1.w_lo = (long)3072 * 768; f1.out_lo = p.lo_ffn;
        RUN("gemm_ffn1", launch_gemm_bf16(EPI_BF16, f1, s));
        if (aud_e && audit16(c, AUD_FFN1, ffn, M, 3072, 3072, s)) return 1;
        GemmArgs f2 = {};
        f2.X = ffn; f2.ldx = 3072; f2.W = d.w2; f2.M = M; f2.N = 768; f2.K = 3072; f2.bias = d.b2;
        f2.out0 = pre; f2.ld0 = 768; f2.res = pre; f2.ldres = 768; f2.ln_stats = stats; f2.ln_gamma = d.ln1w; f2.ln_beta = d.ln1b;
        f2.tune_cfg = c->opt_gemm_cfg; f2.tune_persist = c->opt_gemm_persist; f2.tune_tail = c->opt_gemm_tail; f2.tune_h192 = c->opt_gemm_h192; f2.tune_model = c->opt_gemm_model;  f2.fmt = c->fmt; f2.tune_pre = c->opt_resln_pre;
        f2.x_lo = p.lo_ffn; f2.w_lo = (long)768 * 3072;
        RUN("gemm_ffn2", launch_gemm_bf16(EPI_F32_RESLN, f2, s));
        }
        RUN("layernorm", run_ln(d.ln2w, d.ln2b, last, f8));
        res_g = d.ln2w; res_b = d.ln2b;
        if (last) break;
    }
    return 0;
}


// Graph mode (sylber_set_graph_mode): ~110 launches per forward are launch-latency bound for short / single
// utterances (1.5 ms for one 3 s clip with 0.4 ms of kernel work).  The second call with the same
// (B, Lmax, input, output) captures the launch sequence on the caller's stream into a hipGraph; later calls replay it.
static void graphs_clear(sylber_ctx* c) {
    for (auto& g : c->graphs) if (g.exec) hipGraphExecDestroy(g.exec);
    c->graphs.clear();
}

extern "C" int sylber_set_graph_mode(sylber_t c, int32_t enable) {
    if (!c) return 1;
    c->graph_mode = enable != 0;
    if (!enable) graphs_clear(c);
    return 0;
}

extern "C" int sylber_forward(sylber_t c, const float* wav_dev, const int32_t* lengths_host, int32_t B, int32_t Lmax,
                              float* hidden_dev, void* stream) {
    if (!c || !wav_dev || !hidden_dev) { syl_set_error("sylber_forward", "null argument"); return 1; }
    if (c->precision == SYLBER_FP32) { GUARD_DEVICE(c->device); return forward_f32(c, wav_dev, lengths_host, B, Lmax, hidden_dev, (hipStream_t)stream); }
    if (B < 1 || Lmax < 400) { syl_set_error("sylber_forward", "need B >= 1 and at least 400 samples (one frame)"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    GUARD_DEVICE(c->device);
    Plan p;
    make_plan(B, Lmax, p, c->precision == SYLBER_SPLIT16 ? 2 : 1);
    char* ws_before = c->ws;
    if (ensure_workspace(c, p, s)) return 1;
    if (c->ws != ws_before) graphs_clear(c);           // captured graphs hold workspace addresses
    int* valid = (int*)(c->ws + p.o_valid);
    // valid frames per utterance (TP:664-689): conv-length formula of the number of valid samples
    if (upload_valid(valid, lengths_host, B, Lmax, s)) return 1;
    if (!c->graph_mode || c->profiling || c->opt_audit16 || s == nullptr) return forward_launch(c, p, wav_dev, hidden_dev, s);
    GraphEntry* e = nullptr;
    for (auto& g : c->graphs)
        if (g.B == B && g.Lmax == Lmax && g.stop_stage == c->stop_stage && g.in == wav_dev && g.out == hidden_dev) e = &g;
    if (e && e->exec) { e->stamp = ++c->graph_clock; HIP_TRY(hipGraphLaunch(e->exec, s)); return 0; }
    if (!e) {                                          // first sighting: run eagerly (also sets the kernels' attributes)
        if (c->graphs.size() >= 8) {                   // evict the least recently used entry
            size_t v = 0;
            for (size_t i = 1; i < c->graphs.size(); ++i) if (c->graphs[i].stamp < c->graphs[v].stamp) v = i;
            if (c->graphs[v].exec) hipGraphExecDestroy(c->graphs[v].exec);
            c->graphs.erase(c->graphs.begin() + v);
        }
        c->graphs.push_back({B, Lmax, c->stop_stage, wav_dev, hidden_dev, nullptr, ++c->graph_clock});
        return forward_launch(c, p, wav_dev, hidden_dev, s);
    }
    HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = forward_launch(c, p, wav_dev, hidden_dev, s);
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(s, &graph);
    if (rc != 0 || ec != hipSuccess || !graph) {
        if (graph) hipGraphDestroy(graph);
        if (rc == 0) syl_set_error("sylber_forward", "hipGraph capture failed");
        return 1;
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (ei != hipSuccess) { syl_set_error("sylber_forward", "hipGraphInstantiate failed"); return 1; }
    e->exec = exec; e->stamp = ++c->graph_clock;
    HIP_TRY(hipGraphLaunch(exec, s));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// fp32 parity mode: same sequence, every tensor fp32, own workspace plan
static int forward_f32(sylber_ctx* c, const float* wav_dev, const int32_t* lengths_host, int B, int Lmax, float* hidden_dev,
                       hipStream_t s) {
    if (B < 1 || Lmax < 400) { syl_set_error("sylber_forward", "need B >= 1 and at least 400 samples (one frame)"); return 1; }
    Plan p;
    make_plan(B, Lmax, p);
    const size_t M = (size_t)B * p.Tp;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_a = take(((size_t)B * p.R[0] + 8) * 512 * 4), o_b = take(((size_t)B * p.R[1] + 8) * 512 * 4);
    const size_t o_ln = take(M * 512 * 4), o_x = take(M * 768 * 4), o_xpad = take((size_t)B * (p.Tp + 128) * 768 * 4);
    const size_t o_pre = take(M * 768 * 4), o_h = take(M * 768 * 4), o_qkv = take(M * 2304 * 4), o_ctx = take(M * 768 * 4);
    const size_t o_ffn = take(M * 3072 * 4), o_part = take((size_t)B * p.nchunk * 65 * 8), o_ss = take((size_t)B * 512 * 2 * 4);
    const size_t o_valid = take((size_t)B * 4);
    Plan q = p; q.total = off; q.zero_all = true;
    if (ensure_workspace(c, q, s)) return 1;
    char* w = c->ws;
    float* bufA = (float*)(w + o_a); float* bufB = (float*)(w + o_b); float* ln512 = (float*)(w + o_ln);
    float* xf = (float*)(w + o_x); float* xpad = (float*)(w + o_xpad); float* pre = (float*)(w + o_pre); float* h = (float*)(w + o_h);
    float* qkv = (float*)(w + o_qkv); float* ctx = (float*)(w + o_ctx); float* ffn = (float*)(w + o_ffn);
    double* part = (double*)(w + o_part); float* ss = (float*)(w + o_ss); int* valid = (int*)(w + o_valid);
    if (upload_valid(valid, lengths_host, B, Lmax, s)) return 1;
    RUN("conv0_stats", launch_conv0_stats(wav_dev, B, Lmax, p.L[0], part, p.nchunk, s));
    RUN("conv0_finalize", launch_conv0_finalize(part, p.nchunk, c->conv0_w, c->gn_w, c->gn_b, B, p.L[0], ss, s));
    RUN("conv0_gn_gelu", launch_conv0_gn_gelu(wav_dev, B, Lmax, p.L[0], p.R[0], c->conv0_w, ss, bufA, 1, s));
    float* src = bufA; float* dst = bufB;
    for (int i = 1; i < 7; ++i) {
        GemmArgsF32 a = {};
        a.X = src; a.ldx = (long)CS[i] * 512; a.W = c->conv_w32[i]; a.M = B * p.R[i]; a.N = 512; a.K = CK[i] * 512; a.act = 1;
        a.out0 = dst; a.ld0 = 512; a.tiled = 1;
        RUN("gemm_f32", launch_gemm_f32(a, s));
        float* t = src; src = dst; dst = t;
    }
    float* feats = src;
    if (c->stop_stage == 1) {
        for (int b = 0; b < B; ++b)
            HIP_TRY(hipMemcpyAsync(hidden_dev + (size_t)b * p.T * 512, feats + (size_t)b * p.Tp * 512, (size_t)p.T * 512 * 4,
                                   hipMemcpyDeviceToDevice, s));
        return 0;
    }
    {
        LnArgs a = {};
        a.in = feats; a.in_bf16 = 0; a.ld_in = 512; a.gamma = c->fp_ln_w; a.beta = c->fp_ln_b; a.out_f32 = ln512; a.ld_f32 = 512;
        a.M = (int)M; a.D = 512;
        RUN("ln512", launch_layernorm(a, s));
        GemmArgsF32 g = {};
        g.X = ln512; g.ldx = 512; g.W = c->fp_w32; g.M = (int)M; g.N = 768; g.K = 512; g.bias = c->fp_b; g.out0 = xf; g.ld0 = 768;
        g.Tp = p.Tp; g.T = p.T; g.valid = valid; g.xpad = xpad; g.xpad_rows = p.Tp + 128; g.tiled = 1;
        RUN("gemm_f32", launch_gemm_f32(g, s));
    }
    RUN("posconv_f32", launch_posconv_f32(xpad, c->pos_w32, c->pos_b, xf, pre, B, p.Tp, s));
    auto run_ln = [&](const float* gam, const float* bet, bool last) -> int {
        LnArgs a = {};
        a.in = pre; a.in_bf16 = 0; a.ld_in = 768; a.gamma = gam; a.beta = bet; a.M = (int)M; a.D = 768;
        if (last) { a.out_f32 = hidden_dev; a.ld_f32 = 768; a.Tp = p.Tp; a.T = p.T; }
        else { a.out_f32 = h; a.ld_f32 = 768; }
        return launch_layernorm(a, s);
    };
    RUN("layernorm", run_ln(c->enc_ln_w, c->enc_ln_b, c->stop_stage == 2));
    if (c->stop_stage == 2) return 0;
    for (int l = 0; l < c->num_layers; ++l) {
        const LayerDev& d = c->L[l];
        const bool last = (l == c->num_layers - 1) || (c->stop_stage == 3 + l);
        GemmArgsF32 g = {};
        g.X = h; g.ldx = 768; g.W = c->L32[l].wqkv; g.M = (int)M; g.N = 2304; g.K = 768; g.bias = d.bqkv; g.out0 = qkv; g.ld0 = 2304; g.tiled = 1;
        RUN("gemm_f32", launch_gemm_f32(g, s));
        RUN("attention_f32", launch_attention_f32(qkv, qkv + 768, qkv + 1536, valid, ctx, B, p.T, p.Tp, s));
        GemmArgsF32 o = {};
        o.X = ctx; o.ldx = 768; o.W = c->L32[l].wo; o.M = (int)M; o.N = 768; o.K = 768; o.bias = d.bo; o.out0 = pre; o.ld0 = 768;
        o.res = h; o.ldres = 768; o.tiled = 1;
        RUN("gemm_f32", launch_gemm_f32(o, s));
        RUN("layernorm", run_ln(d.ln1w, d.ln1b, false));
        GemmArgsF32 f1 = {};
        f1.X = h; f1.ldx = 768; f1.W = c->L32[l].w1; f1.M = (int)M; f1.N = 3072; f1.K = 768; f1.bias = d.b1; f1.act = 1;
        f1.out0 = ffn; f1.ld0 = 3072; f1.tiled = 1;
        RUN("gemm_f32", launch_gemm_f32(f1, s));
        GemmArgsF32 f2 = {};
        f2.X = ffn; f2.ldx = 3072; f2.W = c->L32[l].w2; f2.M = (int)M; f2.N = 768; f2.K = 3072; f2.bias = d.b2; f2.out0 = pre; f2.ld0 = 768;
        f2.res = h; f2.ldres = 768; f2.tiled = 1;
        RUN("gemm_f32", launch_gemm_f32(f2, s));
        RUN("layernorm", run_ln(d.ln2w, d.ln2b, last));
        if (last) break;
    }
    return 0;
}

extern "C" int sylber_segment(sylber_t c, const float* hidden_dev, int32_t B, int32_t T, int32_t D, float norm_thr,
                              float merge_thr, int64_t* seg_dev, int32_t* nseg_dev, float* feat_dev, void* stream) {
    if (!c || !hidden_dev || !seg_dev || !nseg_dev) { syl_set_error("sylber_segment", "null argument"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    GUARD_DEVICE(c->device);
    // per-utterance slab of the wide path (frame norms, slot table, bookkeeping of long runs; grow-only)
    const size_t need = segment_scratch_floats(B, T, D);
    if (need > c->seg_scratch_floats) {
        HIP_TRY(hipStreamSynchronize(s));
        if (c->seg_scratch) HIP_TRY(hipFree(c->seg_scratch));
        c->seg_scratch = nullptr; c->seg_scratch_floats = 0;
        HIP_TRY(hipMalloc((void**)&c->seg_scratch, need * 4));
        c->seg_scratch_floats = need;
    }
    ProfScope ps(c, s, "segment");
    return launch_segment(hidden_dev, B, T, D, norm_thr, merge_thr, seg_dev, nseg_dev, feat_dev, c->seg_scratch, s, c->opt_segment);
}

// ------------------------------------------------------------------------------------------------
// single-op entry points for unit parity tests
struct TmpBuf {
    void* p = nullptr;
    ~TmpBuf() { if (p) hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes) == hipSuccess ? 0 : 1; }
};

// `tile` argument of the op-level entry points: -1 = automatic, else tile id + 1000 x (workgroups per CU; 9 = one workgroup per tile)
// + 100000 x t (tail policy of the launch, GemmArgs::tune_tail: t = 1 never split by rows, t >= 2 force a split with tail tile id t - 2)
static void decode_tile(int tile, GemmArgs& g) {
    if (tile >= 1000000) { g.tune_mfma16 = -1; tile -= 1000000; if (tile == 999) tile = -1; }   // tile + 1000000: the 16-bit-output role on the 32x32x16 kernels (GemmArgs::tune_mfma16); 1000999 = that with the automatic tile
    const int t = tile >= 100000 ? tile / 100000 : 0;
    if (tile >= 100000) tile %= 100000;
    g.tune_tail = t == 0 ? 0 : (t == 1 ? -1 : t - 1);
    g.tune_cfg = tile < 0 ? 0 : tile % 1000 + 1;
    g.tune_persist = tile >= 9000 ? -1 : (tile >= 1000 ? tile / 1000 : 0);
}

extern "C" int sylber_op_linear(const float* a_dev, const float* w_dev, const float* bias_dev, float* c_dev, int32_t M,
                                int32_t N, int32_t K, int32_t act, int32_t precision, int32_t tile, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (precision == SYLBER_FP8) {
        // both operands quantised to MXFP8 on the device, contraction on the block-scaled fp8 MFMA
        if (K % 128 != 0) { syl_set_error("sylber_op_linear", "fp8 needs K % 128 == 0"); return 1; }
        TmpBuf a8, as, w8, wsc;
        const long Mp = ((long)M + 255) & ~255L, Np = ((long)N + 255) & ~255L;
        if (a8.alloc((size_t)M * K) || as.alloc((size_t)Mp * (K / 32)) || w8.alloc((size_t)N * K) || wsc.alloc((size_t)Np * (K / 32))) {
            syl_set_error("sylber_op_linear", "alloc"); return 1;
        }
        HIP_TRY(hipMemsetAsync(as.p, 127, (size_t)Mp * (K / 32), s)); HIP_TRY(hipMemsetAsync(wsc.p, 127, (size_t)Np * (K / 32), s));
        if (launch_mx_quant_rows(a_dev, K, (uint8_t*)a8.p, K, (uint8_t*)as.p, Mp, M, K, s)) return 1;
        if (launch_mx_quant_rows(w_dev, K, (uint8_t*)w8.p, K, (uint8_t*)wsc.p, Np, N, K, s)) return 1;
        GemmF8Args g = {};
        g.g.M = M; g.g.N = N; g.g.K = K; g.g.bias = bias_dev; g.g.act = act; g.g.out0 = c_dev; g.g.ld0 = N;
        g.g.tune_cfg = tile < 0 ? 0 : tile + 1;
        g.X8 = (uint8_t*)a8.p; g.ldx8 = K; g.XS = (uint8_t*)as.p; g.xs_rows = Mp; g.W8 = (uint8_t*)w8.p; g.WS = (uint8_t*)wsc.p; g.ws_rows = Np;
        if (launch_gemm_mxfp8(EPI_F32, g, s)) return 1;
        HIP_TRY(hipStreamSynchronize(s));
        return 0;
    }
    if (precision == SYLBER_SPLIT16) {
        // both operands as hi / lo half planes, three MFMA passes into one fp32 accumulator
        TmpBuf ab, wb;
        const long xp = (((long)M + 128) * K + 255) & ~255L, wp = (((long)N + 128) * K + 255) & ~255L;
        if (ab.alloc((size_t)xp * 4) || wb.alloc((size_t)wp * 4)) { syl_set_error("sylber_op_linear", "alloc"); return 1; }
        HIP_TRY(hipMemsetAsync(ab.p, 0, (size_t)xp * 4, s)); HIP_TRY(hipMemsetAsync(wb.p, 0, (size_t)wp * 4, s));
        if (launch_f32_to_split16(a_dev, (bf16_t*)ab.p, xp, (size_t)M * K, s)) return 1;
        if (launch_f32_to_split16(w_dev, (bf16_t*)wb.p, wp, (size_t)N * K, s)) return 1;
        GemmArgs g = {};
        g.X = (bf16_t*)ab.p; g.ldx = K; g.W = (bf16_t*)wb.p; g.M = M; g.N = N; g.K = K; g.bias = bias_dev; g.act = act;
        g.out0 = c_dev; g.ld0 = N; g.fmt = FMT_SPLIT; g.x_lo = xp; g.w_lo = wp;
        decode_tile(tile, g);
        if (launch_gemm_bf16(EPI_F32, g, s)) return 1;
        HIP_TRY(hipStreamSynchronize(s));
        return 0;
    }
    if (precision != SYLBER_BF16) { syl_set_error("sylber_op_linear", "precision must be bf16, fp8 or split16"); return 1; }
    TmpBuf ab, wb;
    if (ab.alloc(((size_t)M + 128) * K * 2) || wb.alloc(((size_t)N + 128) * K * 2)) { syl_set_error("sylber_op_linear", "alloc"); return 1; }
    if (launch_f32_to_bf16(a_dev, (bf16_t*)ab.p, (size_t)M * K, s)) return 1;
    if (launch_f32_to_bf16(w_dev, (bf16_t*)wb.p, (size_t)N * K, s)) return 1;
    GemmArgs g = {};
    g.X = (bf16_t*)ab.p; g.ldx = K; g.W = (bf16_t*)wb.p; g.M = M; g.N = N; g.K = K; g.bias = bias_dev; g.act = act;
    g.out0 = c_dev; g.ld0 = N; decode_tile(tile, g);
    if (launch_gemm_bf16(EPI_F32, g, s)) return 1;
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

// the residual GEMM of an encoder block (attention out-projection, FFN2): pre[M,N] <- A W^T + bias + LayerNorm(pre) in place,
// the LayerNorm re-derived from the row statistics (mean, rstd) and affine the previous LayerNorm launch left (EPI_F32_RESLN)
extern "C" int sylber_op_linear_resln(const float* a_dev, const float* w_dev, const float* bias_dev, float* pre_dev,
                                      const float* stats_dev, const float* gamma_dev, const float* beta_dev, int32_t M, int32_t N,
                                      int32_t K, int32_t tile, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    TmpBuf ab, wb;
    if (ab.alloc(((size_t)M + 128) * K * 2) || wb.alloc(((size_t)N + 128) * K * 2)) { syl_set_error("sylber_op_linear_resln", "alloc"); return 1; }
    if (launch_f32_to_bf16(a_dev, (bf16_t*)ab.p, (size_t)M * K, s)) return 1;
    if (launch_f32_to_bf16(w_dev, (bf16_t*)wb.p, (size_t)N * K, s)) return 1;
    GemmArgs g = {};
    g.X = (bf16_t*)ab.p; g.ldx = K; g.W = (bf16_t*)wb.p; g.M = M; g.N = N; g.K = K; g.bias = bias_dev;
    g.out0 = pre_dev; g.ld0 = N; g.res = pre_dev; g.ldres = N; g.ln_stats = stats_dev; g.ln_gamma = gamma_dev; g.ln_beta = beta_dev;
    decode_tile(tile, g);
    if (launch_gemm_bf16(EPI_F32_RESLN, g, s)) return 1;
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

// one 3-tap stride-2 conv layer of the feature extractor as the 16-bit forward runs it (implicit GEMM over channels-last rows in
// the chunk-major K order, GELU, 16-bit out): x [R, 512] fp32 rows (R >= 2 M + 1), w [512 out][512 in][3] fp32 (torch Conv1d
// layout), y16 [M, 512] bf16 words, y[m] = gelu(sum_{t, c} w[:, c, t] x[2 m + t, c])
extern "C" int sylber_op_conv3(const float* x_dev, const float* w_host, uint16_t* y16_dev, int32_t R, int32_t M, int32_t tile, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (M < 1 || R < 2 * M + 1) { syl_set_error("sylber_op_conv3", "need R >= 2 M + 1 input rows"); return 1; }
    TmpBuf xb, wb;
    if (xb.alloc(((size_t)R + 130) * 512 * 2) || wb.alloc((size_t)512 * 1536 * 2)) { syl_set_error("sylber_op_conv3", "alloc"); return 1; }
    HIP_TRY(hipMemsetAsync(xb.p, 0, ((size_t)R + 130) * 512 * 2, s));
    if (launch_f32_to_bf16(x_dev, (bf16_t*)xb.p, (size_t)R * 512, s)) return 1;
    std::vector<bf16_t> wp((size_t)512 * 1536);
    for (int o = 0; o < 512; ++o)
        for (int pos = 0; pos < 1536; ++pos) {
            const int e = tap3_offset(pos * 2) / 2, t = e / 512, cc = e % 512;     // operand-row element = (tap t, channel cc)
            wp[(size_t)o * 1536 + pos] = f2bf(w_host[((size_t)o * 512 + cc) * 3 + t]);
        }
    HIP_TRY(hipMemcpyAsync(wb.p, wp.data(), wp.size() * 2, hipMemcpyHostToDevice, s));
    GemmArgs g = {};
    g.X = (bf16_t*)xb.p; g.ldx = 1024; g.W = (bf16_t*)wb.p; g.M = M; g.N = 512; g.K = 1536; g.act = ACT_GELU_FAST; g.kpat = 1;
    g.out0 = y16_dev; g.ld0 = 512; decode_tile(tile, g);
    if (launch_gemm_bf16(EPI_BF16, g, s)) return 1;
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

// the same GEMM with its 16-bit output epilogue (EPI_BF16: what the conv layers and FFN1 run): C16 = bf16 / fp16 words
extern "C" int sylber_op_linear16(const float* a_dev, const float* w_dev, const float* bias_dev, uint16_t* c16_dev, int32_t M,
                                  int32_t N, int32_t K, int32_t act, int32_t precision, int32_t tile, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (precision != SYLBER_BF16 && precision != SYLBER_FP16) { syl_set_error("sylber_op_linear16", "precision must be bf16 or fp16"); return 1; }
    if (precision == SYLBER_FP16) { syl_set_error("sylber_op_linear16", "fp16 operands are packed by sylber_create only"); return 1; }
    TmpBuf ab, wb;
    if (ab.alloc(((size_t)M + 128) * K * 2) || wb.alloc(((size_t)N + 128) * K * 2)) { syl_set_error("sylber_op_linear16", "alloc"); return 1; }
    if (launch_f32_to_bf16(a_dev, (bf16_t*)ab.p, (size_t)M * K, s)) return 1;
    if (launch_f32_to_bf16(w_dev, (bf16_t*)wb.p, (size_t)N * K, s)) return 1;
    GemmArgs g = {};
    g.X = (bf16_t*)ab.p; g.ldx = K; g.W = (bf16_t*)wb.p; g.M = M; g.N = N; g.K = K; g.bias = bias_dev; g.act = act;
    g.out0 = c16_dev; g.ld0 = N; decode_tile(tile, g);
    if (launch_gemm_bf16(EPI_BF16, g, s)) return 1;
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

extern "C" int sylber_op_mx_quantize(const float* x_dev, int32_t R, int32_t K, uint8_t* data_dev, uint8_t* scale_dev, void* stream) {
    if (!x_dev || !data_dev || !scale_dev) { syl_set_error("sylber_op_mx_quantize", "null argument"); return 1; }
    return launch_mx_quant_rows(x_dev, K, data_dev, K, scale_dev, R, R, K, (hipStream_t)stream);
}

extern "C" int sylber_op_layernorm(const float* x_dev, const float* res_dev, const float* g_dev, const float* b_dev,
                                   float* y_dev, int32_t M, int32_t D, void* stream) {
    LnArgs a = {};
    a.in = x_dev; a.in_bf16 = 0; a.ld_in = D; a.res = res_dev; a.ld_res = D; a.gamma = g_dev; a.beta = b_dev;
    a.out_f32 = y_dev; a.ld_f32 = D; a.M = M; a.D = D;
    return launch_layernorm(a, (hipStream_t)stream);
}

// q,k,v [B,T,768] f32 -> bf16 head-major q (x SYL_Q_SCALE = log2(e) / 8), k and key-permuted V^T
__global__ void pack_qkv_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                bf16_t* __restrict__ qo, bf16_t* __restrict__ ko, bf16_t* __restrict__ vto, int T, int Tp, int Tpv) {
    const int b = blockIdx.y, t = blockIdx.x;
    for (int c = threadIdx.x; c < 768; c += 256) {
        const int head = c >> 6, d = c & 63;
        const size_t src = ((size_t)b * T + t) * 768 + c;
        const size_t hm = (((size_t)b * 12 + head) * Tp + t) * 64 + d;
        qo[hm] = f2bf(q[src] * SYL_Q_SCALE);
        ko[hm] = f2bf(k[src]);
        const int pos = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1);
        vto[(((size_t)b * 12 + head) * 64 + d) * Tpv + pos] = f2bf(v[src]);
    }
}

// q,k [B,T,768] f32 -> MXFP8 head-major q (x0.125), k: e4m3 [B,H,Tp,64] + one E8M0 scale per 32 features [B,H,Tp,2].
// One 32-lane group per (token, head, 32-feature block).
__global__ void pack_qk_f8_kernel(const float* __restrict__ q, const float* __restrict__ k, uint8_t* __restrict__ q8, uint8_t* __restrict__ qs,
                                  uint8_t* __restrict__ k8, uint8_t* __restrict__ ks, int T, int Tp) {
    const int b = blockIdx.y, t = blockIdx.x;
    for (int c = threadIdx.x; c < 768; c += 256) {
        const int head = c >> 6, d = c & 63;
        const size_t src = ((size_t)b * T + t) * 768 + c;
        const size_t row = ((size_t)b * 12 + head) * Tp + t;
        const float v[2] = {q[src] * 0.125f, k[src]};
        uint8_t* dst[2] = {q8, k8};
        uint8_t* sc[2] = {qs, ks};
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            float amax = fabsf(v[w]);
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
            const unsigned e = mx_e8m0(amax);
            dst[w][row * 64 + d] = (uint8_t)(pack_fp8x4(v[w] * mx_inv_scale(e), 0.f, 0.f, 0.f) & 0xffu);
            if ((d & 31) == 0) sc[w][row * 2 + (d >> 5)] = (uint8_t)e;
        }
    }
}
// v [B,T,768] f32 -> V^T MXFP8: e4m3 [B,H,64,Tpv] (natural key order) + one E8M0 scale per 32 keys [B,H,64,Tpv/32]; keys >= T are zero.
// One thread per (feature, 32-key block).
__global__ void pack_vt_f8_kernel(const float* __restrict__ v, uint8_t* __restrict__ v8, uint8_t* __restrict__ vs, int T, int Tpv) {
    const int b = blockIdx.y, kb = blockIdx.x;           // 32-key block
    for (int c = threadIdx.x; c < 768; c += 256) {
        const int head = c >> 6, d = c & 63;
        float x[32];
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int t = kb * 32 + i;
            x[i] = t < T ? v[((size_t)b * T + t) * 768 + c] : 0.f;
            amax = fmaxf(amax, fabsf(x[i]));
        }
        const unsigned e = mx_e8m0(amax);
        const float inv = mx_inv_scale(e);
        const size_t row = ((size_t)b * 12 + head) * 64 + d;
        unsigned* dst = (unsigned*)(v8 + row * Tpv + kb * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = pack_fp8x4(x[4 * i] * inv, x[4 * i + 1] * inv, x[4 * i + 2] * inv, x[4 * i + 3] * inv);
        vs[row * (Tpv / 32) + kb] = (uint8_t)e;
    }
}

struct AttnF8Bufs {
    TmpBuf q8, qs, k8, ks, v8, vs, cb;
    int alloc(int B, int Tp, int Tpv) {
        const size_t n = (size_t)B * Tp * 768;
        // (k and its scales: one 64-key tile of slack, as in the bf16 path)
        return q8.alloc(n) || qs.alloc(n / 32) || k8.alloc(n + 64 * 64) || ks.alloc(n / 32 + 128) || v8.alloc((size_t)B * 768 * Tpv) ||
               vs.alloc((size_t)B * 768 * (Tpv / 32)) || cb.alloc(n * 2);
    }
    int clear(int B, int Tp, int Tpv, hipStream_t s) {
        const size_t n = (size_t)B * Tp * 768;
        HIP_TRY(hipMemsetAsync(q8.p, 0, n, s)); HIP_TRY(hipMemsetAsync(qs.p, 127, n / 32, s));
        HIP_TRY(hipMemsetAsync(k8.p, 0, n + 64 * 64, s)); HIP_TRY(hipMemsetAsync(ks.p, 127, n / 32 + 128, s));
        HIP_TRY(hipMemsetAsync(v8.p, 0, (size_t)B * 768 * Tpv, s)); HIP_TRY(hipMemsetAsync(vs.p, 127, (size_t)B * 768 * (Tpv / 32), s));
        HIP_TRY(hipMemsetAsync(cb.p, 0, n * 2, s));
        return 0;
    }
};

extern "C" int sylber_op_attention(const float* q_dev, const float* k_dev, const float* v_dev, const int32_t* valid_dev,
                                   float* o_dev, int32_t B, int32_t T, int32_t precision, int32_t queries_per_wave, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (precision != SYLBER_BF16 && precision != SYLBER_FP8) { syl_set_error("sylber_op_attention", "precision: SYLBER_BF16 or SYLBER_FP8"); return 1; }
    const int Tp = (T + 31) & ~31, Tpv = (Tp + 63) & ~63;
    if (precision == SYLBER_FP8) {
        AttnF8Bufs f;
        if (f.alloc(B, Tp, Tpv)) { syl_set_error("sylber_op_attention", "alloc"); return 1; }
        if (f.clear(B, Tp, Tpv, s)) return 1;
        hipLaunchKernelGGL(pack_qk_f8_kernel, dim3(T, B), dim3(256), 0, s, q_dev, k_dev, (uint8_t*)f.q8.p, (uint8_t*)f.qs.p, (uint8_t*)f.k8.p, (uint8_t*)f.ks.p, T, Tp);
        hipLaunchKernelGGL(pack_vt_f8_kernel, dim3(Tpv / 32, B), dim3(256), 0, s, v_dev, (uint8_t*)f.v8.p, (uint8_t*)f.vs.p, T, Tpv);
        if (launch_attention_f8((uint8_t*)f.q8.p, (uint8_t*)f.qs.p, (uint8_t*)f.k8.p, (uint8_t*)f.ks.p, (uint8_t*)f.v8.p, (uint8_t*)f.vs.p, valid_dev,
                                f.cb.p, nullptr, 0, B, T, Tp, Tpv, s)) return 1;
        if (launch_bf16_to_f32_rows((bf16_t*)f.cb.p, 768, o_dev, B, Tp, T, 768, s)) return 1;
        HIP_TRY(hipStreamSynchronize(s));
        return 0;
    }
    const int qw = queries_per_wave == 32 ? 1 : (queries_per_wave == 64 ? 2 : 0);
    TmpBuf qb, kb, vb, cb;
    const size_t n = (size_t)B * Tp * 768;
    // (k: one 64-key tile of slack -- the kernel's last K tile may start at Tp - 32 and reads 64 rows; the scores of rows >= Tp are masked)
    if (qb.alloc(n * 2) || kb.alloc(n * 2 + 64 * 64 * 2) || vb.alloc((size_t)B * 768 * Tpv * 2) || cb.alloc(n * 2)) { syl_set_error("sylber_op_attention", "alloc"); return 1; }
    HIP_TRY(hipMemsetAsync(qb.p, 0, n * 2, s)); HIP_TRY(hipMemsetAsync(kb.p, 0, n * 2 + 64 * 64 * 2, s));
    HIP_TRY(hipMemsetAsync(vb.p, 0, (size_t)B * 768 * Tpv * 2, s)); HIP_TRY(hipMemsetAsync(cb.p, 0, n * 2, s));
    hipLaunchKernelGGL(pack_qkv_kernel, dim3(T, B), dim3(256), 0, s, q_dev, k_dev, v_dev, (bf16_t*)qb.p, (bf16_t*)kb.p, (bf16_t*)vb.p, T, Tp, Tpv);
    if (launch_attention((bf16_t*)qb.p, (bf16_t*)kb.p, (bf16_t*)vb.p, valid_dev, (bf16_t*)cb.p, B, T, Tp, Tpv, qw, s)) return 1;
    if (launch_bf16_to_f32_rows((bf16_t*)cb.p, 768, o_dev, B, Tp, T, 768, s)) return 1;
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}


__global__ void fill_random_bf16(bf16_t* p, size_t n, unsigned seed);
// test aid: every byte of the handle's activation workspace becomes `byte` (0xFF: NaN patterns in every format), and the next forward
// re-runs the zeroing of the regions that are read without being written (as after a batch-shape change).  A forward that then
// returns the same bits as before reads nothing it has not written -- stale data can never leak into a result.
extern "C" int sylber_debug_poison_workspace(sylber_t c, int32_t byte) {
    if (!c) { syl_set_error("sylber_debug_poison_workspace", "null handle"); return 1; }
    GUARD_DEVICE(c->device);
    HIP_TRY(hipDeviceSynchronize());
    if (c->ws) HIP_TRY(hipMemset(c->ws, byte & 0xff, c->ws_bytes));
    if (c->seg_scratch) HIP_TRY(hipMemset(c->seg_scratch, byte & 0xff, c->seg_scratch_floats * 4));   // (long-utterance bookkeeping slab of sylber_segment)
    c->ws_B = 0; c->ws_Lmax = 0;
    if (c->graph_mode) { for (auto& g : c->graphs) if (g.exec) hipGraphExecDestroy(g.exec); c->graphs.clear(); }
    return 0;
}

// kernel-only timing of the attention core on random packed operands (development aid): precision SYLBER_BF16 or SYLBER_FP8
extern "C" int sylber_debug_attention_bench(int32_t B, int32_t T, int32_t precision, int32_t iters, float* ms_out) {
    const int Tp = (T + 31) & ~31, Tpv = (Tp + 63) & ~63;
    const size_t n = (size_t)B * Tp * 768;
    TmpBuf qin;
    if (qin.alloc((size_t)B * T * 768 * 4 * 3)) { syl_set_error("sylber_debug_attention_bench", "alloc"); return 1; }
    float* q = (float*)qin.p; float* k = q + (size_t)B * T * 768; float* v = k + (size_t)B * T * 768;
    const bool zero_data = iters < 0;         // iters < 0: all-zero operands (DVFS probe: the same instruction stream at lower switching power)
    if (zero_data) { iters = -iters; HIP_TRY(hipMemset(qin.p, 0, (size_t)B * T * 768 * 4 * 3)); }
    else
    {   // pseudo-random fp32 q, k, v in [-1, 1) via the bf16 filler (values irrelevant for timing beyond being finite)
        TmpBuf tmp;
        if (tmp.alloc((size_t)B * T * 768 * 3 * 2)) { syl_set_error("sylber_debug_attention_bench", "alloc"); return 1; }
        hipLaunchKernelGGL(fill_random_bf16, dim3(2048), dim3(256), 0, 0, (bf16_t*)tmp.p, (size_t)B * T * 768 * 3, 7u);
        if (launch_bf16_to_f32_rows((bf16_t*)tmp.p, 768, q, 3 * B, T, T, 768, 0)) return 1;
        HIP_TRY(hipDeviceSynchronize());
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int rc = 0;
    if (precision == SYLBER_FP8) {
        AttnF8Bufs f;
        if (f.alloc(B, Tp, Tpv) || f.clear(B, Tp, Tpv, 0)) { syl_set_error("sylber_debug_attention_bench", "alloc"); return 1; }
        hipLaunchKernelGGL(pack_qk_f8_kernel, dim3(T, B), dim3(256), 0, 0, q, k, (uint8_t*)f.q8.p, (uint8_t*)f.qs.p, (uint8_t*)f.k8.p, (uint8_t*)f.ks.p, T, Tp);
        hipLaunchKernelGGL(pack_vt_f8_kernel, dim3(Tpv / 32, B), dim3(256), 0, 0, v, (uint8_t*)f.v8.p, (uint8_t*)f.vs.p, T, Tpv);
        auto run = [&]() { return launch_attention_f8((uint8_t*)f.q8.p, (uint8_t*)f.qs.p, (uint8_t*)f.k8.p, (uint8_t*)f.ks.p, (uint8_t*)f.v8.p, (uint8_t*)f.vs.p,
                                                      nullptr, f.cb.p, nullptr, 0, B, T, Tp, Tpv, 0); };
        for (int i = 0; i < 3 && !rc; ++i) rc = run();
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters && !rc; ++i) rc = run();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    } else {
        TmpBuf qb, kb, vb, cb;
        if (qb.alloc(n * 2) || kb.alloc(n * 2 + 64 * 64 * 2) || vb.alloc((size_t)B * 768 * Tpv * 2) || cb.alloc(n * 2)) { syl_set_error("sylber_debug_attention_bench", "alloc"); return 1; }
        HIP_TRY(hipMemset(qb.p, 0, n * 2)); HIP_TRY(hipMemset(kb.p, 0, n * 2 + 64 * 64 * 2)); HIP_TRY(hipMemset(vb.p, 0, (size_t)B * 768 * Tpv * 2));
        hipLaunchKernelGGL(pack_qkv_kernel, dim3(T, B), dim3(256), 0, 0, q, k, v, (bf16_t*)qb.p, (bf16_t*)kb.p, (bf16_t*)vb.p, T, Tp, Tpv);
        // precision SYLBER_BF16: the default kernel (hand-scheduled key loop); 132 / 164: the compiler-scheduled kernels, 32 / 64 queries per wave
        const int qw = precision == 132 ? 1 : (precision == 164 ? 2 : (precision > 200 && precision < 220 ? precision - 100 : 0));   // 201..209: knock-out variants (experiments build)
        auto run = [&]() { return launch_attention((bf16_t*)qb.p, (bf16_t*)kb.p, (bf16_t*)vb.p, nullptr, (bf16_t*)cb.p, B, T, Tp, Tpv, qw, 0); };
        for (int i = 0; i < 3 && !rc; ++i) rc = run();
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters && !rc; ++i) rc = run();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / (iters > 0 ? iters : 1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// GEMM micro-benchmark (development aid): times `iters` launches of the bf16 GEMM on pseudo-random
// operands with HIP events.  cfg: -1 auto, 0 = 256x128, 1 = 128x192, 2 = 128x128 tiles.
__global__ void fill_random_bf16(bf16_t* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = f2bf(((float)(x & 0xffff) / 32768.0f - 1.0f) * 0.5f);
    }
}
__global__ void fill_random_fp8(uint8_t* p, size_t n, unsigned seed, int scale_bytes) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        unsigned b = x & 0xffu;
        if ((b & 0x7fu) == 0x7fu) b ^= 1u;                    // no NaN codes
        p[i] = scale_bytes ? (uint8_t)(124u + (b & 7u)) : (uint8_t)b;
    }
}

// MXFP8 leg of the micro-benchmark: cfg = 100 + tile config (0 = 128x192, 1 = 128x128, 2 = 256x256 8-wave, 3 = 256x192 8-wave); epi 0 = MXFP8 output (FFN1),
// 1 = fp32 output, 6 = fp32 residual + LayerNorm re-derivation (FFN2)
static int gemm_bench_f8(int M, int N, int K, int epi, int act, int cfg, int iters, float* ms_out) {
    TmpBuf xb, xs, wb, wsb, ob, os, bb, st;
    const long Mp = ((long)M + 255) & ~255L, Np = ((long)N + 255) & ~255L;
    if (xb.alloc((size_t)M * K) || xs.alloc((size_t)Mp * (K / 32)) || wb.alloc((size_t)N * K) || wsb.alloc((size_t)Np * (K / 32)) ||
        ob.alloc((size_t)M * N * 4) || os.alloc((size_t)Mp * (N / 32 + 2)) || bb.alloc((size_t)N * 4 * 3) || st.alloc((size_t)M * 8)) {
        syl_set_error("sylber_debug_gemm_bench", "alloc"); return 1;
    }
    hipLaunchKernelGGL(fill_random_fp8, dim3(2048), dim3(256), 0, 0, (uint8_t*)xb.p, (size_t)M * K, 1u, 0);
    hipLaunchKernelGGL(fill_random_fp8, dim3(2048), dim3(256), 0, 0, (uint8_t*)wb.p, (size_t)N * K, 2u, 0);
    hipLaunchKernelGGL(fill_random_fp8, dim3(256), dim3(256), 0, 0, (uint8_t*)xs.p, (size_t)Mp * (K / 32), 3u, 1);
    hipLaunchKernelGGL(fill_random_fp8, dim3(256), dim3(256), 0, 0, (uint8_t*)wsb.p, (size_t)Np * (K / 32), 4u, 1);
    HIP_TRY(hipMemset(bb.p, 0, (size_t)N * 12)); HIP_TRY(hipMemset(st.p, 0, (size_t)M * 8)); HIP_TRY(hipMemset(ob.p, 0, (size_t)M * N * 4));
    GemmF8Args g = {};
    g.g.M = M; g.g.N = N; g.g.K = K; g.g.bias = (float*)bb.p; g.g.act = act; g.g.out0 = ob.p; g.g.ld0 = N;
    g.g.res = (float*)ob.p; g.g.ldres = N; g.g.ln_stats = (float*)st.p; g.g.ln_gamma = (float*)bb.p + N; g.g.ln_beta = (float*)bb.p + 2 * N;
    g.X8 = (uint8_t*)xb.p; g.ldx8 = K; g.XS = (uint8_t*)xs.p; g.xs_rows = Mp; g.W8 = (uint8_t*)wb.p; g.WS = (uint8_t*)wsb.p; g.ws_rows = Np;
    g.out_scale = (uint8_t*)os.p; g.os_rows = Mp;
    const int e = epi == 0 ? EPI_MXFP8 : (epi == 6 ? EPI_F32_RESLN : EPI_F32);
    g.g.tune_cfg = cfg < 0 ? 0 : cfg + 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int rc = 0;
    for (int i = 0; i < 3 && !rc; ++i) rc = launch_gemm_mxfp8(e, g, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters && !rc; ++i) rc = launch_gemm_mxfp8(e, g, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return rc;
}

static int gemm_bench_impl(int32_t M, int32_t N, int32_t K, int32_t ldx, int32_t epi, int32_t act, int32_t cfg, int32_t iters,
                           float* ms_out, unsigned long long* g_gemm_trace_out) {
    if (cfg >= 100 && cfg < 200) return gemm_bench_f8(M, N, K, epi, act, cfg - 100, iters, ms_out);
    const bool no_h192 = act >= 10000;                    // act + 10000: the cost model without the 192-row tiles (A/B)
    act %= 10000;
    const int tail_code = act / 100;                      // act + 100 t: tail policy of the launch (GemmArgs::tune_tail)
    act %= 100;
    const bool legacy16 = cfg >= 1000000;                 // cfg + 1000000: the 16-bit-output role on the 32x32x16 kernels (GemmArgs::tune_mfma16 = -1)
    if (legacy16) cfg -= 1000000;
    const bool kpat = cfg >= 350000;                      // cfg + 400000: the 3-tap conv layers' chunk-major K order (K = 1536, ldx = 1024)
    if (kpat) cfg -= 400000;
    const bool cold = cfg >= 150000;                      // cfg + 200000: operands flushed out of the caches before every launch
    if (cold) cfg -= 200000;
    TmpBuf xb, wb, ob, rb, bb;
    const size_t xn = (size_t)(M + 8) * ldx + K, wn = (size_t)N * K;
    if (xb.alloc(xn * 2) || wb.alloc(wn * 2) || ob.alloc((size_t)M * N * 4) || rb.alloc((size_t)M * N * 4) || bb.alloc((size_t)N * 4)) {
        syl_set_error("sylber_debug_gemm_bench", "alloc"); return 1;
    }
    hipLaunchKernelGGL(fill_random_bf16, dim3(2048), dim3(256), 0, 0, (bf16_t*)xb.p, xn, 1u);
    hipLaunchKernelGGL(fill_random_bf16, dim3(2048), dim3(256), 0, 0, (bf16_t*)wb.p, wn, 2u);
    HIP_TRY(hipMemset(rb.p, 0, (size_t)M * N * 4)); HIP_TRY(hipMemset(bb.p, 0, (size_t)N * 4));
    GemmArgs g = {};
    g.X = (bf16_t*)xb.p; g.ldx = ldx; g.W = (bf16_t*)wb.p; g.M = M; g.N = N; g.K = K; g.bias = (float*)bb.p; g.act = act;
    g.out0 = ob.p; g.ld0 = N; g.res = (float*)rb.p; g.ldres = N; g.kpat = kpat ? 1 : 0;
    TmpBuf lnb;
    if (epi == EPI_F32_RESLN) {
        if (lnb.alloc((size_t)M * 8 + (size_t)N * 8)) { syl_set_error("sylber_debug_gemm_bench", "alloc"); return 1; }
        HIP_TRY(hipMemset(lnb.p, 0, (size_t)M * 8 + (size_t)N * 8));
        g.ln_stats = (float*)lnb.p; g.ln_gamma = (float*)lnb.p + (size_t)M * 2; g.ln_beta = g.ln_gamma + N;
    }
    g.Tp = 512; g.Tpv = 512; g.T = 499;               // EPI_QK: rows = (utterance, frame) at this pitch
    TmpBuf qkb;
    if (epi == EPI_QK) {
        if (qkb.alloc((size_t)(M + 512) * 768 * 2 * 3)) { syl_set_error("sylber_debug_gemm_bench", "alloc"); return 1; }
        g.out0 = qkb.p; g.out1 = (char*)qkb.p + (size_t)(M + 512) * 768 * 2; g.out2 = (char*)qkb.p + (size_t)(M + 512) * 768 * 4;
    }
    g.tune_cfg = cfg < 0 ? 0 : (cfg % 1000) + 1;
    g.tune_persist = cfg >= 9000 ? -1 : (cfg >= 1000 ? cfg / 1000 : 0);    // cfg = persist * 1000 + tile (9000 + tile: persist = -1)
    g.tune_h192 = no_h192 ? -1 : 0; g.tune_model = no_h192 ? 5 : 0;
    g.tune_mfma16 = legacy16 ? -1 : 0;
    g.tune_tail = tail_code == 0 ? 0 : (tail_code == 1 ? -1 : tail_code - 1);   // act + 100 t: t = 1 never split, t >= 2 force tail tile id t - 2
    TmpBuf trb;
    if (g_gemm_trace_out) {
        if (trb.alloc(20 * 8)) { syl_set_error("sylber_debug_gemm_bench", "alloc"); return 1; }
        HIP_TRY(hipMemset(trb.p, 0, 20 * 8));
        g.trace = (unsigned long long*)trb.p;
    }
    struct TraceFetch {
        TmpBuf& b; unsigned long long* dst;
        ~TraceFetch() { if (dst && b.p) { hipDeviceSynchronize(); hipMemcpy(dst, b.p, 20 * 8, hipMemcpyDeviceToHost); } }
    } trace_fetch{trb, g_gemm_trace_out};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int rc = 0;
    for (int i = 0; i < 3 && !rc; ++i) rc = launch_gemm_bf16(epi, g, 0);
    if (cold) {
        // every timed launch finds its operands COLD: 1 GiB is written between the launches (the 256 MB memory-side cache
        // and the L2s hold none of X / W / res afterwards), each launch timed by its own event pair
        TmpBuf flush;
        if (flush.alloc((size_t)1 << 30)) { syl_set_error("sylber_debug_gemm_bench", "alloc"); return 1; }
        float tot = 0.f;
        for (int i = 0; i < iters && !rc; ++i) {
            hipMemsetAsync(flush.p, i & 0xff, (size_t)1 << 30, 0);
            hipEventRecord(e0, 0);
            rc = launch_gemm_bf16(epi, g, 0);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms1 = 0.f;
            hipEventElapsedTime(&ms1, e0, e1);
            tot += ms1;
        }
        *ms_out = tot / iters;
        hipEventDestroy(e0); hipEventDestroy(e1);
        return rc;
    }
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters && !rc; ++i) rc = launch_gemm_bf16(epi, g, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return rc;
}

// which tile the cost model of launch_gemm_bf16 picks (host arithmetic only; no device is touched): epi / act as GemmEpi / GemmAct, fmt 0 bf16 / 1 fp16,
// model = SYLBER_OPT_GEMM_MODEL (0: the handle owns the chip, 5: shares it), kpat = 1 for the 3-tap conv K order
extern "C" int sylber_debug_gemm_pick(int32_t M, int32_t N, int32_t K, int32_t epi, int32_t act, int32_t fmt, int32_t model, int32_t kpat) {
    GemmArgs g = {};
    g.M = M; g.N = N; g.K = K; g.act = act; g.fmt = fmt; g.tune_model = model % 100; g.kpat = kpat;
    g.tune_mfma16 = model >= 100 ? -1 : 0;               // model + 100: the 16-bit-output role on the 32x32x16 kernels
    return gemm_pick_tile(epi, g);
}

extern "C" int sylber_debug_gemm_bench(int32_t M, int32_t N, int32_t K, int32_t ldx, int32_t epi, int32_t act, int32_t cfg,
                                       int32_t iters, float* ms_out) {
    return gemm_bench_impl(M, N, K, ldx, epi, act, cfg, iters, ms_out, nullptr);
}

// the trace instantiation of the 8-wave kernel (tile id 30: s_memtime stamps around the phases of the K loop): one
// launch series, then the 2 x 10 cycle counters of workgroup 0's waves 0 (group 0) and 4 (group 1):
// [0] sum A (loop top -> fragments landed), [1] sum barrier after A, [2] sum B (MFMA + DMA issue), [3] sum barrier after B,
// [4] K loop total, [5] steps, [6] cost of one stamp, [7] epilogue, [8] prologue, [9] tile total
extern "C" int sylber_debug_gemm_trace(int32_t M, int32_t N, int32_t K, int32_t ldx, int32_t epi, int32_t act,
                                       unsigned long long* out20, float* ms_out) {
    if (!out20 || !ms_out) { syl_set_error("sylber_debug_gemm_trace", "null argument"); return 1; }
    // act >= 100: the trace instantiation of the UNSTAGGERED kernel (tile id 41; plain bf16 epilogue only)
    if (act >= 200) return gemm_bench_impl(M, N, K, ldx, 0, 1, 98, 3, ms_out, out20);   // asm tile 97 with phase stamps (SYLBER_EXPERIMENTS builds)
    if (act >= 100) return gemm_bench_impl(M, N, K, ldx, 0, 0, 9041, 3, ms_out, out20);
    return gemm_bench_impl(M, N, K, ldx, epi, act, 9030, 3, ms_out, out20);
}
