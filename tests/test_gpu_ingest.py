"""GPU tier, row N1: ``sylber_ingest`` (decode + resample to 16 kHz + normalise in HIP, through the C-ABI) against
the CPU oracle — bit-exact for decode and resampling (exact float32 products summed in float64 in the same order),
one float32 ulp of the statistics for the normalisation — and the file entry point of the drop-in Segmenter."""
import wave

import numpy as np
import pytest
import torch

from oracle import ingest_ref as R

pytestmark = pytest.mark.gpu


def _pcm(raw, sr, ch, width):
    from sylber_amd.ingest import PcmFile
    raw = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
    return PcmFile(raw, sr, ch, width, raw.size // (ch * abs(width)))


def _signal(n, ch, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n)[:, None] / 8000.0
    x = 0.4 * np.sin(2 * np.pi * (200 + 50 * np.arange(ch))[None, :] * t) + 0.1 * rng.standard_normal((n, ch)) + 0.05
    return np.clip(x, -0.99, 0.99)


@pytest.mark.parametrize("sr,ch,width", [(16000, 1, 2), (16000, 2, 1), (8000, 1, 2), (22050, 2, 2), (44100, 1, 2),
                                         (48000, 2, 4), (11025, 1, 3), (32000, 1, 1), (44100, 1, 3),
                                         (16000, 1, -4), (44100, 2, -4), (48000, 1, -8)])
def test_ingest_matches_oracle(sr, ch, width):
    from sylber_amd.ingest import ingest_pcm, num_frames_16k
    n = 3000 if sr != 16000 else 5000
    x = _signal(n, ch, sr + ch)
    if width == -4:
        raw = x.astype("<f4")                                     # WAVE_FORMAT_IEEE_FLOAT
    elif width == -8:
        raw = x.astype("<f8")
    elif width == 2:
        raw = np.round(x * 32767).astype("<i2")
    elif width == 4:
        raw = np.round(x * (2 ** 31 - 1)).astype("<i4")
    elif width == 1:
        raw = np.round(x * 127 + 128).astype(np.uint8)
    else:
        v = np.round(x * (2 ** 23 - 1)).astype(np.int64).reshape(-1)
        raw = np.stack([v & 255, (v >> 8) & 255, (v >> 16) & 255], 1).astype(np.uint8)
    pcm = _pcm(raw, sr, ch, width)
    assert pcm.frames == n
    exp = R.ingest(pcm.data, width, ch, sr, do_normalize=False)
    got = ingest_pcm(pcm, "cuda:0", normalize=False).cpu().numpy()
    assert got.shape == exp.shape == (ch, num_frames_16k(n, sr)) and got.dtype == np.float32
    assert np.array_equal(got, exp)                               # bit-exact
    expn = R.normalize(exp)
    gotn = ingest_pcm(pcm, "cuda:0", normalize=True).cpu().numpy()
    assert np.abs(gotn - expn).max() <= 2.5e-7 * np.abs(expn).max() + 1e-7   # <= 1 ulp of mean / std
    assert abs(float(gotn.astype(np.float64).mean())) < 1e-6 and abs(float(gotn.astype(np.float64).std(ddof=1)) - 1) < 1e-6
    t = torch.from_numpy(exp)
    ref = ((t - t.mean()) / t.std()).numpy()                     # the reference's own torch ops (sylber.py:86)
    assert np.abs(gotn - ref).max() <= 4e-6 * np.abs(ref).max()


def test_ingest_is_deterministic_and_large():
    from sylber_amd.ingest import ingest_pcm
    rng = np.random.default_rng(5)
    raw = rng.integers(-20000, 20000, size=2 * 441000, dtype=np.int64).astype("<i2")   # 10 s stereo at 44.1 kHz
    pcm = _pcm(raw, 44100, 2, 2)
    a = ingest_pcm(pcm, "cuda:0").cpu().numpy()
    b = ingest_pcm(pcm, "cuda:0").cpu().numpy()
    assert a.shape == (2, 160000) and np.array_equal(a, b)
    assert abs(a.astype(np.float64).mean()) < 1e-6 and abs(a.astype(np.float64).std(ddof=1) - 1) < 1e-6


def test_ingest_errors():
    from sylber_amd import _lib
    from sylber_amd.ingest import ingest_pcm
    with pytest.raises(_lib.SylberHipError):
        ingest_pcm(_pcm(np.zeros(16, np.uint8), 16000, 1, 2)._replace(sample_width=5), "cuda:0")
    lib = _lib.load()
    assert lib.sylber_ingest_num_frames(441000, 44100) == 160000
    assert lib.sylber_ingest_num_frames(100001, 44100) == -(-160 * 100001 // 441)
    assert lib.sylber_ingest_num_frames(7, 16000) == 7 and lib.sylber_ingest_num_frames(5, 0) == -1


def test_segmenter_accepts_non_16k_files(tmp_path):
    """the reference resamples any file to 16 kHz (sylber.py:84-85); the same clip stored at 16 kHz and at 48 kHz
    (sample-repeated = a legitimate 48 kHz file) must give the same number of frames and close hidden states"""
    from sylber_amd import Segmenter
    from sylber_amd.synth import syllable_wave
    from sylber_amd.weights import synthetic_state_dict
    S = Segmenter(model_ckpt=synthetic_state_dict(0))
    x = syllable_wave(24000, 11)[0].numpy()
    x = x / np.abs(x).max() * 0.8
    p16, p48 = str(tmp_path / "a16.wav"), str(tmp_path / "a48.wav")
    t48 = np.arange(72000) / 48000.0
    x48 = np.interp(t48, np.arange(24000) / 16000.0, x)
    for p, sig, sr in ((p16, x, 16000), (p48, x48, 48000)):
        with wave.open(p, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes(np.round(sig * 32767).astype("<i2").tobytes())
    a = S(p16, in_second=False)
    b = S([p48, p16], in_second=False)
    assert a["hidden_states"].shape == b[0]["hidden_states"].shape == b[1]["hidden_states"].shape
    assert np.isfinite(b[0]["hidden_states"]).all()
    # the batch row of the 16 kHz file reproduces the single-file call (same length => no padding difference)
    assert np.array_equal(a["segments"], b[1]["segments"])


def test_float_and_extensible_wav_headers(tmp_path):
    """read_pcm parses WAVE_FORMAT_IEEE_FLOAT and WAVE_FORMAT_EXTENSIBLE headers (torchaudio.load reads both); a
    non-RIFF file is refused with a clear error"""
    import struct
    from sylber_amd.ingest import ingest_file, read_pcm
    x = _signal(4000, 1, 5).astype("<f4")

    def riff(fmt, data):
        body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 3) + b"abc\0" + b"data" + struct.pack("<I", len(data)) + data
        return b"RIFF" + struct.pack("<I", len(body)) + body

    p1 = tmp_path / "f32.wav"
    p1.write_bytes(riff(struct.pack("<HHIIHH", 3, 1, 16000, 16000 * 4, 4, 32), x.tobytes()))
    pcm = read_pcm(str(p1))
    assert (pcm.sample_rate, pcm.channels, pcm.sample_width, pcm.frames) == (16000, 1, -4, 4000)
    got = ingest_file(str(p1), "cuda:0", normalize=False).cpu().numpy()
    assert np.array_equal(got[0], x[:, 0])
    sub = struct.pack("<H", 1) + bytes.fromhex("000000001000800000aa00389b71")
    i16 = np.round(x * 32767).astype("<i2")
    p2 = tmp_path / "ext.wav"
    p2.write_bytes(riff(struct.pack("<HHIIHH", 0xFFFE, 1, 16000, 32000, 2, 16) + struct.pack("<HHI", 22, 16, 4) + sub, i16.tobytes()))
    pcm2 = read_pcm(str(p2))
    assert (pcm2.sample_width, pcm2.frames) == (2, 4000)
    p3 = tmp_path / "x.flac"
    p3.write_bytes(b"fLaC" + b"\0" * 64)
    with pytest.raises(ValueError):
        read_pcm(str(p3))
