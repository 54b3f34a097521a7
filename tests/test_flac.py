"""csrc/flac_host.hip (host-side FLAC decode of row N1's file ingest) against streams made by tests/flac_enc.py -- an encoder written independently of the
decoder from the published format, since the build image holds no FLAC file, codec or encoder (parity with torchaudio: unpinned, like the resampler's).  CPU tier:
the entry points touch no device."""
import ctypes
import hashlib

import numpy as np
import pytest

from tests import flac_enc


@pytest.fixture(scope="module")
def lib():
    from sylber_amd import _lib
    return _lib.load()


def decode(lib, blob, expect_error=None):
    from sylber_amd import _lib
    raw = np.frombuffer(blob, dtype=np.uint8).copy()
    sr, nch, bps, frames, got = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
    p = raw.ctypes.data_as(ctypes.c_void_p)
    rc = lib.sylber_flac_info(p, raw.size, ctypes.byref(sr), ctypes.byref(nch), ctypes.byref(bps), ctypes.byref(frames))
    if rc != 0:
        assert expect_error and any(e in lib.sylber_last_error().decode() for e in expect_error.split("|")), lib.sylber_last_error().decode()
        return None
    n = frames.value
    if n == 0:
        rc = lib.sylber_flac_decode(p, raw.size, None, 0, ctypes.byref(got))
        if rc == 0:
            n = got.value
    out = np.full((max(n, 1), nch.value), -7, dtype=np.int32)
    rc = lib.sylber_flac_decode(p, raw.size, out.ctypes.data_as(ctypes.c_void_p), n, ctypes.byref(got))
    if expect_error:
        assert rc != 0 and any(e in lib.sylber_last_error().decode() for e in expect_error.split("|")), (rc, lib.sylber_last_error().decode())
        return None
    _lib.check(rc, "sylber_flac_decode")
    assert got.value == n
    return out[:n], sr.value, bps.value


def signal(n, nch, bps, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    amp = (1 << (bps - 1)) * 0.35
    x = np.stack([amp * (np.sin(2 * np.pi * (0.003 + 0.002 * c) * t) + 0.3 * np.sin(2 * np.pi * 0.031 * t + c)) + amp * 0.02 * rng.standard_normal(n)
                  for c in range(nch)], 1)
    if nch == 2:
        x[:, 1] = 0.8 * x[:, 0] + 0.2 * x[:, 1]                # correlated channels: what the side codings exist for
    return np.clip(np.round(x), -(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int64)


def test_fixed_predictors_partitions_and_block_sizes(lib):
    """FIXED orders 0-4 and every partition order the block allows, coded and explicit block sizes, a short last block; 16-bit mono and stereo"""
    for nch in (1, 2):
        pcm = signal(4096 * 3 + 1000, nch, 16, 1 + nch)
        for bs, explicit in ((4096, False), (1152, False), (1000, False), (256, True), (4000, True)):
            def choose(f, c, s, bs=bs):
                po = [p for p in range(0, 6) if len(s) % (1 << p) == 0 and (len(s) >> p) >= 4]
                return {"kind": "fixed", "order": min((f + c) % 5, len(s)), "porder": po[(f + c) % len(po)]}
            blob = flac_enc.encode(pcm, 16000, 16, bs, choose, explicit=explicit)
            out, sr, bps = decode(lib, blob)
            assert sr == 16000 and bps == 16 and np.array_equal(out, pcm), (nch, bs)


def test_lpc_orders_precisions_methods_and_escapes(lib):
    """LPC orders 1-32 with 5-15-bit coefficients, residual coding method 1 (5-bit Rice parameters), an escape (raw) partition, forced shifts"""
    pcm = signal(4096 * 2, 2, 16, 11)
    orders = [1, 2, 3, 8, 12, 16, 24, 32]
    for method in (0, 1):
        def choose(f, c, s):
            o = orders[(2 * f + c) % len(orders)]
            return {"kind": "lpc", "order": o, "precision": 5 + ((3 * f + c) % 11), "porder": (f + c) % 4, "method": method, "escape": 1 if (f + c) % 3 == 0 and (f + c) % 4 else None}
        blob = flac_enc.encode(pcm, 44100, 16, 1024, choose)
        out, sr, _ = decode(lib, blob)
        assert sr == 44100 and np.array_equal(out, pcm), method
    blob = flac_enc.encode(pcm, 16000, 16, 2048, lambda f, c, s: {"kind": "lpc", "order": 4, "precision": 14, "shift": f % 3, "porder": 3})
    assert np.array_equal(decode(lib, blob)[0], pcm)


def test_stereo_assignments_and_sample_sizes(lib):
    """independent / left-side / side-right / mid-side per frame; 8, 12, 16, 20, 24 and 32 bits per sample (a 33-bit side channel), odd sizes through STREAMINFO"""
    for bps in (8, 12, 16, 20, 24, 32, 13, 7):
        pcm = signal(3000, 2, bps, 20 + bps)
        pcm[5] = [(1 << (bps - 1)) - 1, -(1 << (bps - 1))]     # the extremes: the side channel needs its extra bit
        pcm[6] = [-(1 << (bps - 1)), (1 << (bps - 1)) - 1]
        kinds = ["verbatim", "fixed", "lpc"]

        def choose(f, c, s, bps=bps):
            if bps == 32:                                        # residuals are 32-bit quantities: a real encoder codes such blocks verbatim or with a low order
                return {"kind": "verbatim"} if f == 0 else {"kind": "fixed", "order": 1, "porder": 2, "method": 1}
            return {"kind": kinds[(f + c) % len(kinds)], "order": 2, "precision": 10, "porder": 2, "method": 1 if bps > 16 else 0}
        blob = flac_enc.encode(pcm, 48000, bps, 500, choose, assignment=lambda f: (0x1, 8, 9, 10)[f % 4], explicit=bps in (13, 7))
        out, sr, b = decode(lib, blob)
        assert b == bps and sr == 48000 and np.array_equal(out, pcm), bps


def test_constant_verbatim_wasted_bits_and_multichannel(lib):
    pcm = signal(2048, 3, 16, 31)
    pcm[:, 1] = 1234                                             # a constant channel
    pcm[:, 2] = (pcm[:, 2] >> 4) << 4                            # four wasted bits
    spec = [lambda s: {"kind": "verbatim"}, lambda s: {"kind": "constant"}, lambda s: {"kind": "fixed", "order": 3, "porder": 1, "wasted": 4}]
    blob = flac_enc.encode(pcm, 22050, 16, 1024, lambda f, c, s: spec[c](s))
    assert np.array_equal(decode(lib, blob)[0], pcm)
    silent = np.zeros((5000, 1), dtype=np.int64)
    blob = flac_enc.encode(silent, 16000, 16, 4096, lambda f, c, s: {"kind": "constant"})
    assert np.array_equal(decode(lib, blob)[0], silent)
    eight = signal(1500, 8, 16, 32)
    assert np.array_equal(decode(lib, flac_enc.encode(eight, 16000, 16, 512))[0], eight)


def test_container_variants(lib):
    """an ID3v2 tag in front, PADDING / VORBIS_COMMENT / SEEKTABLE blocks, variable blocking (sample numbers in the frame headers), a stream whose encoder
    knew neither its length nor the MD5, many frames (multi-byte coded numbers)"""
    pcm = signal(9000, 1, 16, 41)
    extra = [(1, bytes(100)), (4, b"\x04\x00\x00\x00test\x00\x00\x00\x00"), (3, bytes(18))]
    blob = flac_enc.encode(pcm, 16000, 16, 1024, extra_blocks=extra, id3=300)
    assert np.array_equal(decode(lib, blob)[0], pcm)
    blob = flac_enc.encode(pcm, 16000, 16, 777, variable=True, explicit=True)
    assert np.array_equal(decode(lib, blob)[0], pcm)
    blob = flac_enc.encode(pcm, 16000, 16, 1024, known_length=False, with_md5=False)
    assert np.array_equal(decode(lib, blob)[0], pcm)
    long = signal(16 * 2200, 1, 16, 42)
    assert np.array_equal(decode(lib, flac_enc.encode(long, 8000, 16, 16))[0], long)        # 2200 frames: two- and three-byte frame numbers
    assert np.array_equal(decode(lib, flac_enc.encode(long, 96000, 16, 16, variable=True))[0], long)


def test_corruption_is_detected(lib):
    pcm = signal(6000, 2, 16, 51)
    blob = bytearray(flac_enc.encode(pcm, 16000, 16, 1024))
    bad = bytearray(blob); bad[len(bad) // 2] ^= 0x10
    decode(lib, bytes(bad), expect_error="CRC")
    bad = bytearray(blob); bad[4 + 4 + 18 + 3] ^= 0xff              # the encoder's MD5 in STREAMINFO
    decode(lib, bytes(bad), expect_error="MD5")
    decode(lib, bytes(blob[:len(blob) - 700]), expect_error="stream ends|truncated|CRC")
    decode(lib, b"RIFF" + bytes(blob[4:]), expect_error="fLaC")
    decode(lib, bytes(blob[:20]), expect_error="truncated")
    # a frame that decodes cleanly but to other samples (header and frame CRCs recomputed): only the MD5 can tell
    other = flac_enc.encode(pcm[::-1].copy(), 16000, 16, 1024)
    forged = bytes(blob[:42]) + other[42:]
    decode(lib, forged, expect_error="MD5")
    assert hashlib.md5(b"").hexdigest()                          # (hashlib's MD5 is the independent implementation the encoder uses)


def test_fuzzed_streams_never_crash(lib):
    """a file is untrusted input: 1500 random corruptions (bit flips, byte runs, truncations, spliced headers) of valid streams either decode or fail with an error
    -- no crash, no write beyond the caller's buffer (guard rows stay intact)"""
    rng = np.random.default_rng(61)
    seeds = [flac_enc.encode(signal(3000, 2, 16, 60), 16000, 16, 512, lambda f, c, s: {"kind": ("fixed", "lpc", "verbatim")[(f + c) % 3], "order": 3, "precision": 9, "porder": 2},
                             assignment=lambda f: (1, 8, 9, 10)[f % 4]),
             flac_enc.encode(signal(2000, 1, 24, 61), 44100, 24, 400, lambda f, c, s: {"kind": "lpc", "order": 12, "precision": 15, "porder": 0, "method": 1, "escape": 0}, explicit=True, variable=True)]
    sr, nch, bps, frames, got = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
    ok = 0
    for it in range(1500):
        b = bytearray(seeds[it % 2])
        kind = it % 5
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            i = int(rng.integers(0, len(b) - 8)); b[i:i + 8] = rng.integers(0, 256, 8, dtype=np.uint8).tobytes()
        elif kind == 2:
            b = b[:int(rng.integers(0, len(b)))]
        elif kind == 3:
            i = int(rng.integers(42, len(b) - 4)); b[i:i + 2] = b"\xff\xf8"           # a false sync code
        else:
            i = int(rng.integers(4, 42)); b[i] = int(rng.integers(0, 256))            # STREAMINFO fields
        raw = np.frombuffer(bytes(b), dtype=np.uint8).copy() if len(b) else np.zeros(1, np.uint8)
        p = raw.ctypes.data_as(ctypes.c_void_p)
        if lib.sylber_flac_info(p, len(b), ctypes.byref(sr), ctypes.byref(nch), ctypes.byref(bps), ctypes.byref(frames)) != 0:
            continue
        cap = 4000
        out = np.full((cap + 2, max(nch.value, 1)), 0x5a5a5a5a, dtype=np.int32)
        rc = lib.sylber_flac_decode(p, len(b), out.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(got))
        assert (out[cap:] == 0x5a5a5a5a).all()
        ok += rc == 0
    assert ok < 1500                                             # (the corruptions do get caught; a few hit padding or unused fields and decode)


@pytest.mark.gpu
def test_flac_file_through_the_segmenter(tmp_path):
    """Segmenter(wav_file=...) on a .flac returns what it returns on the .wav holding the same 16-bit audio (the decoded samples reach the device as the same
    int16 PCM), mono and stereo, 16 kHz and a rate that is resampled; a 24-bit FLAC equals its 24-bit WAV"""
    import struct
    import torch
    from sylber_amd import Segmenter
    from sylber_amd.synth import syllable_wave
    from sylber_amd.weights import synthetic_state_dict
    S = Segmenter(model_ckpt=synthetic_state_dict(0))

    def wav_bytes(pcm, sr, bps):
        n, nch = pcm.shape
        w = bps // 8
        data = b"".join(int(x).to_bytes(w, "little", signed=True) for x in pcm.reshape(-1))
        fmt = struct.pack("<HHIIHH", 1, nch, sr, sr * nch * w, nch * w, bps)
        return b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", len(data)) + data
    for sr, nch, bps in ((16000, 1, 16), (22050, 2, 16), (16000, 1, 24)):
        x = torch.cat([syllable_wave(int(sr * 2.5), 70 + c) for c in range(nch)], 0).numpy().T
        pcm = np.clip(np.round(x / np.abs(x).max() * 0.7 * (1 << (bps - 1))), -(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int64)
        fw, ff = tmp_path / ("a_%d_%d_%d.wav" % (sr, nch, bps)), tmp_path / ("a_%d_%d_%d.flac" % (sr, nch, bps))
        fw.write_bytes(wav_bytes(pcm, sr, bps))
        ff.write_bytes(flac_enc.encode(pcm, sr, bps, 4096, lambda f, c, s: {"kind": "lpc", "order": 8, "precision": 12, "porder": 3 if len(s) % 8 == 0 else 0, "method": 1 if bps > 16 else 0},
                                       assignment=lambda f: 10))
        a, b = S(wav_file=[str(fw)]), S(wav_file=[str(ff)])
        assert len(a) == len(b) == nch
        for oa, ob in zip(a, b):
            assert np.array_equal(oa["hidden_states"], ob["hidden_states"]) and np.array_equal(oa["segments"], ob["segments"])
    with pytest.raises(ValueError):
        (tmp_path / "x.ogg").write_bytes(b"OggS" + bytes(100))
        S(wav_file=str(tmp_path / "x.ogg"))
