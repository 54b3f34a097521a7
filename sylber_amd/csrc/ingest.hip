// File ingest on the device: interleaved PCM -> float32 [channels, N] at 16 kHz,
// per-file normalised.  Stands where the reference does (sylber/model/sylber.py:83-86)
//     wav, sr = torchaudio.load(file)
//     if sr != 16000: wav = torchaudio.transforms.Resample(sr, 16000)(wav)
//     wav = (wav - wav.mean()) / wav.std()
//
// torchaudio is a third-party dependency that is absent from the image (requirements pin 2.4.1), so the resampler
// restates its published algorithm ("sinc_interp_hann", lowpass_filter_width 6, rolloff 0.99; kernel computed in
// float64 and cast to float32 because transforms.Resample is built with dtype=None):
//     orig, new = sr_in / g, 16000 / g (g = gcd);  base = min(orig, new) * 0.99;  width = ceil(6 * orig / base)
//     h[p][k] = sinc(pi t) * cos^2(pi t / 12) * base / orig,  t = clamp(((k - width) / orig - p / new) * base, -6, 6)
//     y[f * new + p] = sum_k xpad[f * orig + k] * h[p][k],  k in [0, 2 width + orig),  xpad = zero-pad(x, width, width + orig)
//     output length ceil(new * N / orig)
// Outside |t| < 6 the window is cos^2(pi/2) ~ 4e-33 (not exactly zero in floating point); those taps are skipped
// here (they cannot change a float32 result unless the whole sum is < 1e-25), so each output costs ~2 width + 1
// multiply-adds instead of 2 width + orig.  Products of two floats are exact in double and the taps are added in
// ascending order in double, so the result is independent of the launch shape and reproducible on the host.
//
// All of this is HBM-trivial byte work (0.6 MB per 10 s clip): one thread per output sample, coalesced reads of
// neighbouring input samples, the per-phase filter rows from L2; no LDS, no MFMA.
#include "kernels.h"
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

#define ING_PARTIALS 1024

struct IngestTable {
    int orig = 1, neu = 1, width = 0, cmax = 0;
    std::vector<float> h;        // [neu][cmax] support taps, zero-filled
    std::vector<int> klo;        // [neu] first support tap (index into the 2*width+orig tap axis)
};

static std::mutex g_tab_mu;
static std::map<int, IngestTable*> g_tabs;     // never freed: async H2D copies read from them

static int igcd(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

// a rate is supported when its polyphase table is small: 16000 / gcd phases x (2 width + sr / gcd) taps <= 2^26 floats (every common rate:
// 8 / 11.025 / 22.05 / 24 / 32 / 44.1 / 48 / 96 kHz are <= 1e5).  A corrupt header (or a rate coprime to 16000) would otherwise ask for
// tens of GB of host memory here -- and a C++ exception must not cross the C ABI.
static bool rate_supported(int sr_in) {
    if (sr_in < 1) return false;
    if (sr_in == 16000) return true;
    const int g = igcd(sr_in, 16000);
    const long orig = sr_in / g, neu = 16000 / g;
    const double base = (double)(orig < neu ? orig : neu) * 0.99;
    const long width = (long)std::ceil(6.0 * orig / base);
    return neu * (2 * width + orig) <= (1L << 26);
}

static const IngestTable* get_table(int sr_in) {
    std::lock_guard<std::mutex> lk(g_tab_mu);
    auto it = g_tabs.find(sr_in);
    if (it != g_tabs.end()) return it->second;
    IngestTable* t = new IngestTable;
    const int g = igcd(sr_in, 16000);
    t->orig = sr_in / g; t->neu = 16000 / g;
    const double L = 6.0, rolloff = 0.99;
    const double base = (double)(t->orig < t->neu ? t->orig : t->neu) * rolloff;
    t->width = (int)std::ceil(L * t->orig / base);
    const int taps = 2 * t->width + t->orig;
    const double scale = base / t->orig;
    std::vector<float> full((size_t)t->neu * taps);
    std::vector<int> lo(t->neu, taps), hi(t->neu, -1);
    for (int p = 0; p < t->neu; ++p)
        for (int k = 0; k < taps; ++k) {
            double tt = ((double)(-p) / t->neu + (double)(k - t->width) / t->orig) * base;
            const bool inside = tt > -L && tt < L;
            tt = tt < -L ? -L : (tt > L ? L : tt);
            const double c = std::cos(tt * M_PI / L / 2.0);
            const double window = c * c;
            const double tp = tt * M_PI;
            const double v = (tp == 0.0 ? 1.0 : std::sin(tp) / tp) * (window * scale);
            full[(size_t)p * taps + k] = (float)v;
            if (inside) { if (k < lo[p]) lo[p] = k; if (k > hi[p]) hi[p] = k; }
        }
    for (int p = 0; p < t->neu; ++p) t->cmax = std::max(t->cmax, hi[p] - lo[p] + 1);
    t->h.assign((size_t)t->neu * t->cmax, 0.f);
    t->klo = lo;
    for (int p = 0; p < t->neu; ++p)
        for (int k = lo[p]; k <= hi[p]; ++k) t->h[(size_t)p * t->cmax + (k - lo[p])] = full[(size_t)p * taps + k];
    g_tabs[sr_in] = t;
    return t;
}

// torchaudio.load scaling: int16 / 2^15, int32 / 2^31, 24-bit / 2^23, uint8 (x - 128) / 128; IEEE-float files
// (width -4 = float32, -8 = float64) are taken as they are (float64 rounded to float32)
__device__ __forceinline__ float pcm_sample(const unsigned char* __restrict__ pcm, int width, long idx) {
    if (width == -4) return ((const float*)pcm)[idx];
    if (width == -8) return (float)((const double*)pcm)[idx];
    if (width == 2) return (float)((const short*)pcm)[idx] * (1.0f / 32768.0f);
    if (width == 4) return (float)((const int*)pcm)[idx] * (1.0f / 2147483648.0f);
    if (width == 1) return ((float)pcm[idx] - 128.0f) * (1.0f / 128.0f);
    const unsigned char* b = pcm + idx * 3;
    const int v = (int)((unsigned)b[0] | ((unsigned)b[1] << 8) | ((unsigned)(signed char)b[2] << 16));
    return (float)v * (1.0f / 8388608.0f);
}

__global__ __launch_bounds__(256) void pcm_decode_kernel(const unsigned char* __restrict__ pcm, int width, int C, long N,
                                                         float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // frame
    if (i >= N) return;
    for (int c = 0; c < C; ++c) out[(long)c * N + i] = pcm_sample(pcm, width, i * C + c);
}

__global__ __launch_bounds__(256) void pcm_resample_kernel(const unsigned char* __restrict__ pcm, int width, int C, long N,
                                                           const float* __restrict__ h, const int* __restrict__ klo, int orig,
                                                           int neu, int fwidth, int cmax, long Nout, float* __restrict__ out) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;     // output sample
    const int c = blockIdx.y;
    if (j >= Nout) return;
    const long f = j / neu;
    const int p = (int)(j - f * neu);
    const long i0 = f * orig + klo[p] - fwidth;               // input frame of the first support tap
    const float* hp = h + (long)p * cmax;
    double acc = 0.0;
    for (int k = 0; k < cmax; ++k) {
        const long i = i0 + k;
        const float x = (i >= 0 && i < N) ? pcm_sample(pcm, width, i * C + c) : 0.f;
        acc += (double)x * (double)hp[k];
    }
    out[(long)c * Nout + j] = (float)acc;
}

// ---- (x - mean) / std with the unbiased std over ALL elements of the file (sylber.py:86) -------------------------
// fixed-shape two-stage reductions in double: G partial sums over contiguous chunks, combined in index order by
// every consumer, so the result does not depend on scheduling.
__device__ __forceinline__ double block_sum_f64(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__device__ __forceinline__ double ordered_total(const double* __restrict__ part, int G) {
    double s = 0.0;
    for (int g = 0; g < G; ++g) s += part[g];
    return s;
}

__global__ __launch_bounds__(256) void ingest_sum_kernel(const float* __restrict__ x, long n, long chunk, double* __restrict__ part) {
    __shared__ double sh[4];
    const long b0 = (long)blockIdx.x * chunk, b1 = b0 + chunk < n ? b0 + chunk : n;
    double s = 0.0;
    for (long i = b0 + threadIdx.x; i < b1; i += 256) s += (double)x[i];
    s = block_sum_f64(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void ingest_sqdev_kernel(const float* __restrict__ x, long n, long chunk, int G,
                                                           const double* __restrict__ part_sum, double* __restrict__ part_sq) {
    __shared__ double sh[4];
    const double mean = ordered_total(part_sum, G) / (double)n;
    const long b0 = (long)blockIdx.x * chunk, b1 = b0 + chunk < n ? b0 + chunk : n;
    double s = 0.0;
    for (long i = b0 + threadIdx.x; i < b1; i += 256) { const double d = (double)x[i] - mean; s += d * d; }
    s = block_sum_f64(s, sh);
    if (threadIdx.x == 0) part_sq[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void ingest_normalize_kernel(float* __restrict__ x, long n, int G, const double* __restrict__ part_sum,
                                                               const double* __restrict__ part_sq) {
    const float mean = (float)(ordered_total(part_sum, G) / (double)n);
    const float sd = (float)sqrt(ordered_total(part_sq, G) / (double)(n - 1));     // n == 1 -> NaN, like torch
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = (x[i] - mean) / sd;
}

extern "C" int64_t sylber_ingest_num_frames(int64_t frames_in, int32_t sr_in) {
    if (!rate_supported(sr_in) || frames_in < 0) return -1;
    if (sr_in == 16000) return frames_in;
    const int g = igcd(sr_in, 16000);
    const int64_t orig = sr_in / g, neu = 16000 / g;
    return (neu * frames_in + orig - 1) / orig;
}

extern "C" int64_t sylber_ingest_workspace_bytes(int32_t sr_in) {
    if (!rate_supported(sr_in)) return -1;
    size_t bytes = 2 * ING_PARTIALS * sizeof(double);
    if (sr_in != 16000) {
        const IngestTable* t = get_table(sr_in);
        bytes += (size_t)t->neu * t->cmax * 4 + (size_t)t->neu * 4 + 512;
    }
    return (int64_t)bytes;
}

extern "C" int sylber_ingest(const void* pcm_dev, int32_t sample_width, int32_t channels, int64_t frames_in, int32_t sr_in,
                             int32_t normalize, float* wav_out_dev, void* workspace_dev, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!pcm_dev || !wav_out_dev || !workspace_dev) { syl_set_error("sylber_ingest", "null pointer"); return 1; }
    if (!((sample_width >= 1 && sample_width <= 4) || sample_width == -4 || sample_width == -8)) {
        syl_set_error("sylber_ingest", "sample width must be 1..4 bytes (integer PCM) or -4 / -8 (IEEE float32 / float64)"); return 1;
    }
    if (channels < 1 || channels > 65535 || frames_in < 1 || sr_in < 1) { syl_set_error("sylber_ingest", "bad channels / frames / rate"); return 1; }
    if (!rate_supported(sr_in)) { syl_set_error("sylber_ingest", "unsupported sample rate (its resampling table to 16 kHz would exceed 2^26 taps)"); return 1; }
    const unsigned char* pcm = (const unsigned char*)pcm_dev;
    double* part_sum = (double*)workspace_dev;
    double* part_sq = part_sum + ING_PARTIALS;
    const long Nout = (long)sylber_ingest_num_frames(frames_in, sr_in);
    if (sr_in == 16000) {
        hipLaunchKernelGGL(pcm_decode_kernel, dim3((unsigned)((frames_in + 255) / 256)), dim3(256), 0, s, pcm, sample_width, channels,
                           (long)frames_in, wav_out_dev);
    } else {
        const IngestTable* t = get_table(sr_in);
        float* h_dev = (float*)((char*)workspace_dev + 2 * ING_PARTIALS * sizeof(double));
        int* klo_dev = (int*)(h_dev + (size_t)t->neu * t->cmax);
        HIP_TRY(hipMemcpyAsync(h_dev, t->h.data(), t->h.size() * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(klo_dev, t->klo.data(), t->klo.size() * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(pcm_resample_kernel, dim3((unsigned)((Nout + 255) / 256), channels), dim3(256), 0, s, pcm, sample_width,
                           channels, (long)frames_in, h_dev, klo_dev, t->orig, t->neu, t->width, t->cmax, Nout, wav_out_dev);
    }
    HIP_TRY(hipGetLastError());
    if (normalize) {
        const long n = Nout * channels;
        long chunk = (n + ING_PARTIALS - 1) / ING_PARTIALS;
        chunk = (chunk + 255) / 256 * 256;
        const int G = (int)((n + chunk - 1) / chunk);
        hipLaunchKernelGGL(ingest_sum_kernel, dim3(G), dim3(256), 0, s, wav_out_dev, n, chunk, part_sum);
        hipLaunchKernelGGL(ingest_sqdev_kernel, dim3(G), dim3(256), 0, s, wav_out_dev, n, chunk, G, part_sum, part_sq);
        hipLaunchKernelGGL(ingest_normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, wav_out_dev, n, G, part_sum, part_sq);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}
