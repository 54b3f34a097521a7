"""CPU tier, row N1 (file ingest, sylber/model/sylber.py:83-86): the oracle's decode scaling and normalisation
against the torch CPU ops the reference executes, and the properties of the restated torchaudio resampler
(its output cannot be generated here: torchaudio is absent from the image -> resampler parity is unpinned)."""
import os
import wave

import numpy as np
import pytest
import torch

from oracle import ingest_ref as R


def test_decode_matches_reference_scaling(golden_dir):
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    pcm = g["sample_pcm"]                                          # int16 samples of samples/sample.wav
    x = R.decode_pcm(pcm.view(np.uint8), 2, 1)
    assert x.shape == (1, pcm.size) and x.dtype == np.float32
    assert np.array_equal(x[0], pcm.astype(np.float32) / 32768.0)  # torchaudio.load normalisation of int16
    # 8-bit unsigned, 24-bit and 32-bit signed, two channels interleaved
    rng = np.random.default_rng(0)
    u8 = rng.integers(0, 256, 64, dtype=np.uint8)
    assert np.array_equal(R.decode_pcm(u8, 1, 2), ((u8.astype(np.float32) - 128) / 128).reshape(-1, 2).T)
    i32 = rng.integers(-2 ** 31, 2 ** 31, 64, dtype=np.int64).astype("<i4")
    assert np.array_equal(R.decode_pcm(i32.view(np.uint8), 4, 2), (i32.astype(np.float32) / 2147483648.0).reshape(-1, 2).T)
    i24 = rng.integers(-2 ** 23, 2 ** 23, 64, dtype=np.int64)
    b = np.stack([(i24 & 255), (i24 >> 8) & 255, (i24 >> 16) & 255], 1).astype(np.uint8).reshape(-1)
    assert np.array_equal(R.decode_pcm(b, 3, 1)[0], (i24.astype(np.float32) / 8388608.0))


def test_normalize_matches_torch_ops(golden_dir):
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    x = R.decode_pcm(g["sample_pcm"].view(np.uint8), 2, 1)
    t = torch.from_numpy(x)
    ref = ((t - t.mean()) / t.std()).numpy()                      # the very ops of sylber.py:86
    got = R.normalize(x)
    # float32 statistics of torch differ from the float64-derived ones by at most an ulp of the scalar
    assert np.abs(got - ref).max() <= 4e-6 * np.abs(ref).max()
    stereo = np.stack([x[0, :20000], x[0, 20000:40000]])
    ts = torch.from_numpy(stereo)
    assert np.abs(R.normalize(stereo) - ((ts - ts.mean()) / ts.std()).numpy()).max() <= 4e-6 * 8


@pytest.mark.parametrize("sr", [8000, 11025, 22050, 32000, 44100, 48000])
def test_resampler_properties(sr):
    h, support, orig, new, width = R.sinc_kernel(sr)
    assert h.shape == (new, 2 * width + orig) and h.dtype == np.float32
    # taps with a clamped window argument are exactly zero in float32: skipping them (GPU kernel) is exact
    assert np.all(h[~support] == 0.0)
    # unit DC gain up to the passband ripple of the windowed sinc
    assert np.all(np.abs(h.astype(np.float64).sum(1) - 1.0) < 2e-3)
    n = sr // 2
    assert R.num_frames_16k(n, sr) == int(np.ceil(16000 * n / sr - 1e-9))
    t = np.arange(n) / sr
    x = np.sin(2 * np.pi * 300.0 * t).astype(np.float32)[None]
    y = R.resample_to_16k(x, sr)
    assert y.shape == (1, R.num_frames_16k(n, sr)) and y.dtype == np.float32
    ref = np.sin(2 * np.pi * 300.0 * np.arange(y.shape[1]) / 16000.0)
    assert np.abs(y[0, 200:-200] - ref[200:-200]).max() < 2e-3
    assert np.array_equal(y, R.resample_to_16k(x, sr, support_only=True))


def test_resampler_is_linear_and_shift_consistent():
    rng = np.random.default_rng(3)
    a = rng.standard_normal((1, 4410)).astype(np.float32)
    b = rng.standard_normal((1, 4410)).astype(np.float32)
    ya, yb, yab = R.resample_to_16k(a, 44100), R.resample_to_16k(b, 44100), R.resample_to_16k(a + b, 44100)
    assert np.abs(yab - (ya + yb)).max() < 1e-5
    # a shift by `orig` input samples is a shift by `new` output samples (441 -> 160)
    z = np.concatenate([np.zeros((1, 441), np.float32), a], 1)
    yz = R.resample_to_16k(z, 44100)
    assert np.array_equal(yz[0, 160:160 + ya.shape[1] - 1], ya[0, :-1])
    # 16 kHz passes through untouched
    assert np.array_equal(R.resample_to_16k(a, 16000), a)


def test_read_pcm_header_only(tmp_path):
    from sylber_amd.ingest import read_pcm
    p = str(tmp_path / "x.wav")
    data = (np.arange(200, dtype=np.int16) - 100)
    with wave.open(p, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(22050); w.writeframes(data.tobytes())
    pcm = read_pcm(p)
    assert (pcm.sample_rate, pcm.channels, pcm.sample_width, pcm.frames) == (22050, 2, 2, 100)
    assert np.array_equal(pcm.data.view("<i2"), data)


def test_resampler_independent_cross_check_scipy():
    """torchaudio is absent (resampler parity stays UNPINNED); as an independent check of the restated algorithm the
    oracle's 44.1 kHz -> 16 kHz and 48 kHz -> 16 kHz outputs are compared with scipy.signal.resample_poly (a different
    polyphase low-pass design) on band-limited material: the two agree in the interior to well below the -40 dB that
    any filter-shape difference in the transition band could explain, the sample count is ceil(16000 N / sr), and a
    16 kHz input passes through unchanged."""
    from scipy.signal import resample_poly
    rng = np.random.default_rng(1)
    for sr, up, down in [(44100, 160, 441), (48000, 1, 3), (22050, 320, 441)]:
        n = sr // 2
        t = np.arange(n) / sr
        x = sum(a * np.sin(2 * np.pi * f * t + p) for a, f, p in
                zip(rng.uniform(0.2, 1.0, 6), rng.uniform(80.0, 5500.0, 6), rng.uniform(0, 6.28, 6))).astype(np.float32)
        y = R.resample_to_16k(x[None], sr)[0]
        assert y.shape[0] == -(-16000 * n // sr)
        z = resample_poly(x.astype(np.float64), up, down)[: y.shape[0]]
        core = slice(200, y.shape[0] - 200)
        err = np.sqrt(((y[core] - z[core]) ** 2).mean() / (z[core] ** 2).mean())
        assert err < 5e-3, (sr, err)
    same = rng.standard_normal(4000).astype(np.float32)
    assert np.array_equal(R.resample_to_16k(same[None], 16000)[0], same)
