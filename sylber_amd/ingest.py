"""Device-side file ingest: PCM ``.wav`` -> float32 ``[channels, N]`` at 16 kHz,
per-file normalised, without host arithmetic.  Mirrors sylber/model/sylber.py:83-86
(``torchaudio.load`` -> ``transforms.Resample(sr, 16000)`` -> ``(wav - wav.mean()) / wav.std()``); the kernels
are in csrc/ingest.hip behind ``sylber_ingest`` of include/sylber_hip.h."""
from __future__ import annotations

import ctypes
import struct
from typing import NamedTuple

import numpy as np
import torch

from . import _lib


class PcmFile(NamedTuple):
    data: np.ndarray      # uint8, the raw little-endian interleaved data chunk
    sample_rate: int
    channels: int
    sample_width: int     # bytes per sample (1 = unsigned, 2/3/4 = signed); -4 / -8 = IEEE float32 / float64
    frames: int


def read_pcm(path: str) -> PcmFile:
    """RIFF/WAVE header parse + one read of the data chunk; no sample arithmetic on the host.  Integer PCM
    (WAVE_FORMAT_PCM, 8/16/24/32 bit), IEEE float (WAVE_FORMAT_IEEE_FLOAT, 32/64 bit) and WAVE_FORMAT_EXTENSIBLE
    wrappers of either -- what ``torchaudio.load`` reads from a ``.wav`` (sylber.py:83).  FLAC goes through ``read_flac``; other compressed
    containers (mp3 / ogg, which torchaudio decodes through ffmpeg / sox) are NOT supported: ValueError."""
    with open(str(path), "rb") as f:
        head = f.read(12)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise ValueError("%s is neither a RIFF/WAVE nor a FLAC file (other containers are not supported: decode them yourself and pass wav= tensors)" % path)
        fmt = None
        raw = None
        while True:
            ch = f.read(8)
            if len(ch) < 8:
                break
            cid, size = ch[:4], struct.unpack("<I", ch[4:])[0]
            if cid == b"fmt ":
                fmt = f.read(size)
            elif cid == b"data":
                raw = f.read(size)                       # a truncated file simply yields fewer bytes
                break
            else:
                f.seek(size, 1)
            if size & 1:
                f.seek(1, 1)                             # chunks are word aligned
    if fmt is None or raw is None or len(fmt) < 16:
        raise ValueError("%s has no fmt / data chunk" % path)
    tag, nch, sr, _, _, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:                 # WAVE_FORMAT_EXTENSIBLE: the real tag opens the SubFormat GUID
        tag = struct.unpack("<H", fmt[24:26])[0]
    if tag == 1 and bits in (8, 16, 24, 32):
        width = bits // 8
    elif tag == 3 and bits in (32, 64):
        width = -(bits // 8)                             # negative = IEEE float (include/sylber_hip.h)
    else:
        raise ValueError("%s: unsupported WAVE format tag %d with %d bits per sample" % (path, tag, bits))
    bps = abs(width) * nch
    n = len(raw) // bps
    if n < 1 or nch < 1:
        raise ValueError("%s holds no audio frames" % path)
    return PcmFile(np.frombuffer(raw, dtype=np.uint8, count=n * bps).copy(), int(sr), int(nch), int(width), int(n))


def read_flac(path: str) -> PcmFile:
    """FLAC -> the PcmFile the device path takes: the bitstream is decoded on the host by the library (csrc/flac_host.hip: frame CRCs and the encoder's MD5 of the
    audio are verified), the samples are handed over as 16-bit (up to 16 bits per sample) or 32-bit PCM scaled to full range -- the integer -> float conversion,
    resampling and normalisation stay on the device, and the floats are the ones ``torchaudio.load`` returns for the file (sample / 2^(bits - 1))."""
    lib = _lib.load()
    raw = np.fromfile(str(path), dtype=np.uint8)
    sr, nch, bps, frames = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
    _lib.check(lib.sylber_flac_info(raw.ctypes.data_as(ctypes.c_void_p), raw.size, ctypes.byref(sr), ctypes.byref(nch), ctypes.byref(bps), ctypes.byref(frames)),
               "sylber_flac_info(%s)" % path)
    n = int(frames.value)
    got = ctypes.c_int64()
    if n == 0:                                               # length unknown to the encoder: one verifying pass to count
        _lib.check(lib.sylber_flac_decode(raw.ctypes.data_as(ctypes.c_void_p), raw.size, None, 0, ctypes.byref(got)), "sylber_flac_decode(%s)" % path)
        n = int(got.value)
    if n < 1:
        raise ValueError("%s holds no audio frames" % path)
    if n > raw.size * 8192:                                   # (a block of 65535 constant samples takes ~14 bytes: no stream holds more per byte)
        raise ValueError("%s: STREAMINFO claims %d frames in %d bytes" % (path, n, raw.size))
    pcm = np.empty((n, nch.value), dtype=np.int32)
    _lib.check(lib.sylber_flac_decode(raw.ctypes.data_as(ctypes.c_void_p), raw.size, pcm.ctypes.data_as(ctypes.c_void_p), n, ctypes.byref(got)),
               "sylber_flac_decode(%s)" % path)
    if bps.value <= 16:
        data, width = (pcm << (16 - bps.value)).astype("<i2"), 2
    else:
        data, width = (pcm.astype(np.int64) << (32 - bps.value)).astype("<i4"), 4
    return PcmFile(np.frombuffer(data.tobytes(), dtype=np.uint8).copy(), int(sr.value), int(nch.value), width, n)


def read_audio(path: str) -> PcmFile:
    """RIFF/WAVE or FLAC, by the file's first bytes (an ID3v2 tag may precede a FLAC stream); anything else: ValueError naming the format problem"""
    with open(str(path), "rb") as f:
        head = f.read(4)
    if head == b"fLaC" or head[:3] == b"ID3":
        return read_flac(path)
    return read_pcm(path)


def num_frames_16k(frames: int, sample_rate: int) -> int:
    return int(_lib.load().sylber_ingest_num_frames(int(frames), int(sample_rate)))


def ingest_pcm(pcm: PcmFile, device, normalize: bool = True) -> torch.Tensor:
    """raw PCM -> device tensor ``[channels, N16k]`` float32 (decode, resample to 16 kHz, normalise on the GPU)."""
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise _lib.SylberHipError("no MI355X visible to PyTorch-ROCm; the HIP path has no CPU fallback")
    dev = torch.device(device)
    if not (0 < pcm.sample_rate < 2 ** 31):
        raise ValueError("sample rate %d out of range" % pcm.sample_rate)
    n_out = num_frames_16k(pcm.frames, pcm.sample_rate)
    if n_out < 1:
        raise ValueError("sample rate %d Hz is not supported (no compact polyphase table to 16 kHz)" % pcm.sample_rate)
    raw = torch.from_numpy(np.ascontiguousarray(pcm.data)).to(dev)
    out = torch.empty(pcm.channels, n_out, dtype=torch.float32, device=dev)
    ws = torch.empty(int(lib.sylber_ingest_workspace_bytes(pcm.sample_rate)) // 8 + 1, dtype=torch.float64, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    with torch.cuda.device(dev):
        _lib.check(lib.sylber_ingest(ctypes.c_void_p(raw.data_ptr()), pcm.sample_width, pcm.channels, pcm.frames,
                                     pcm.sample_rate, 1 if normalize else 0, ctypes.c_void_p(out.data_ptr()),
                                     ctypes.c_void_p(ws.data_ptr()), stream), "sylber_ingest")
    # raw / ws were used on torch's current stream only, so the caching allocator may recycle them stream-ordered
    return out


def ingest_file(path: str, device, normalize: bool = True) -> torch.Tensor:
    return ingest_pcm(read_audio(path), device, normalize)
