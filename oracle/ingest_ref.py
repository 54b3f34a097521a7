"""TEST ORACLE (not product code) for the file-ingest row N1: CPU restatement of sylber/model/sylber.py:83-86

    wav, sr = torchaudio.load(file)                                   # PCM -> float32 [C, N] in [-1, 1)
    if sr != 16000: wav = torchaudio.transforms.Resample(sr, 16000)(wav)
    wav = (wav - wav.mean()) / wav.std()

PARITY UNPINNED for the resampler: torchaudio (requirements.txt pins 2.4.1) is a third-party dependency that is
absent from this image and from /root/reference, so its output cannot be generated here.  This file restates the
published algorithm of ``torchaudio.functional.resample`` / ``transforms.Resample`` with the defaults the reference
uses (resampling_method "sinc_interp_hann", lowpass_filter_width 6, rolloff 0.99, dtype None => the filter bank is
computed in float64 and cast to float32):

    g = gcd(sr_in, 16000); orig = sr_in // g; new = 16000 // g
    base = min(orig, new) * rolloff;  width = ceil(lowpass_filter_width * orig / base)
    idx  = arange(-width, width + orig) / orig
    t    = (arange(0, -new, -1)[:, None] / new + idx[None, :]) * base, clamped to [-6, 6]
    h    = where(t == 0, 1, sin(pi t) / (pi t)) * cos(pi t / 6 / 2) ** 2 * (base / orig)          -> float32 [new, 2 width + orig]
    y    = conv1d(pad(x, (width, width + orig)), h, stride=orig), interleave the `new` phases, cut to ceil(new * N / orig)

The decode scaling and the normalisation ARE pinned: tests compare them with torch CPU ops (the same ops the
reference executes) on the golden sample.  Accumulation: float32 products summed in float64 in ascending tap order
(exact products, so the GPU kernel reproduces this bit for bit when it keeps the same order)."""
from __future__ import annotations

import math

import numpy as np

LOWPASS_FILTER_WIDTH = 6
ROLLOFF = 0.99
TARGET_RATE = 16000


def decode_pcm(raw: np.ndarray, sample_width: int, channels: int) -> np.ndarray:
    """interleaved little-endian PCM bytes -> float32 [C, N] scaled like torchaudio.load (sylber.py:83)"""
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    if sample_width == -4:                                 # WAVE_FORMAT_IEEE_FLOAT: torchaudio.load returns the samples as stored
        x = raw.view("<f4").astype(np.float32)
    elif sample_width == -8:
        x = raw.view("<f8").astype(np.float32)
    elif sample_width == 2:
        x = raw.view("<i2").astype(np.float32) * np.float32(1.0 / 32768.0)
    elif sample_width == 4:
        x = raw.view("<i4").astype(np.float32) * np.float32(1.0 / 2147483648.0)
    elif sample_width == 1:
        x = (raw.astype(np.float32) - np.float32(128.0)) * np.float32(1.0 / 128.0)
    elif sample_width == 3:
        b = raw.reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = v - ((v & 0x800000) << 1)
        x = v.astype(np.float32) * np.float32(1.0 / 8388608.0)
    else:
        raise ValueError("sample width")
    return np.ascontiguousarray(x.reshape(-1, channels).T)


def sinc_kernel(sr_in: int):
    """float32 filter bank [new, 2*width+orig] (+ orig, new, width), computed in float64 like transforms.Resample"""
    g = math.gcd(int(sr_in), TARGET_RATE)
    orig, new = int(sr_in) // g, TARGET_RATE // g
    base = min(orig, new) * ROLLOFF
    width = math.ceil(LOWPASS_FILTER_WIDTH * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t_raw = t * base
    t = np.clip(t_raw, -LOWPASS_FILTER_WIDTH, LOWPASS_FILTER_WIDTH)
    window = np.cos(t * math.pi / LOWPASS_FILTER_WIDTH / 2) ** 2
    tp = t * math.pi
    scale = base / orig
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(tp == 0, 1.0, np.sin(tp) / np.where(tp == 0, 1.0, tp))
    k = k * (window * scale)
    support = np.abs(t_raw) < LOWPASS_FILTER_WIDTH
    return k.astype(np.float32), support, orig, new, width


def num_frames_16k(n: int, sr_in: int) -> int:
    g = math.gcd(int(sr_in), TARGET_RATE)
    orig, new = int(sr_in) // g, TARGET_RATE // g
    return -((-new * int(n)) // orig)


def resample_to_16k(x: np.ndarray, sr_in: int, support_only: bool = False) -> np.ndarray:
    """x float32 [C, N] -> float32 [C, ceil(new N / orig)].  ``support_only`` drops the taps whose window argument
    is clamped (values ~1e-33) exactly like the GPU kernel does; the default keeps torchaudio's full filter."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if int(sr_in) == TARGET_RATE:
        return x.copy()
    h, support, orig, new, width = sinc_kernel(sr_in)
    C, N = x.shape
    taps = h.shape[1]
    n_out = num_frames_16k(N, sr_in)
    frames = -(-n_out // new)
    xp = np.zeros((C, width + (frames - 1) * orig + taps + 1), dtype=np.float64)
    xp[:, width:width + N] = x
    y = np.zeros((C, frames, new), dtype=np.float64)
    base_idx = np.arange(frames) * orig
    for p in range(new):
        acc = np.zeros((C, frames), dtype=np.float64)
        for k in range(taps):                       # ascending taps, float64 accumulation of exact products
            if support_only and not support[p, k]:
                continue
            acc += xp[:, base_idx + k] * np.float64(h[p, k])
        y[:, :, p] = acc
    return y.reshape(C, frames * new)[:, :n_out].astype(np.float32)


def normalize(x: np.ndarray) -> np.ndarray:
    """(wav - wav.mean()) / wav.std() over all elements, unbiased (sylber.py:86); statistics in float64, rounded to
    float32 scalars, elementwise float32 arithmetic"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    x64 = x.astype(np.float64)
    mean = np.float32(x64.mean())
    sd = np.float32(np.sqrt(((x64 - x64.mean()) ** 2).sum() / (x.size - 1)))
    return (x - mean) / sd


def ingest(raw: np.ndarray, sample_width: int, channels: int, sr_in: int, do_normalize: bool = True,
           support_only: bool = False) -> np.ndarray:
    y = resample_to_16k(decode_pcm(raw, sample_width, channels), sr_in, support_only)
    return normalize(y) if do_normalize else y
