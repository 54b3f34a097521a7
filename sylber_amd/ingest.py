"""Device-side file ingest: PCM ``.wav`` -> float32 ``[channels, N]`` at 16 kHz,
per-file normalised, without host arithmetic.  Mirrors sylber/model/sylber.py:83-86
(``torchaudio.load`` -> ``transforms.Resample(sr, 16000)`` -> ``(wav - wav.mean()) / wav.std()``); the kernels
are in csrc/ingest.hip behind ``sylber_ingest`` of include/sylber_hip.h."""
from __future__ import annotations

import ctypes
import wave as _wave
from typing import NamedTuple

import numpy as np
import torch

from . import _lib


class PcmFile(NamedTuple):
    data: np.ndarray      # uint8, the raw little-endian interleaved data chunk
    sample_rate: int
    channels: int
    sample_width: int     # bytes per sample (1 = unsigned, 2/3/4 = signed)
    frames: int


def read_pcm(path: str) -> PcmFile:
    """Header parse + one read of the data chunk (stdlib ``wave``: integer PCM only); no sample arithmetic."""
    with _wave.open(str(path), "rb") as w:
        sr, nch, width, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if width not in (1, 2, 3, 4):
        raise ValueError("unsupported sample width %d" % width)
    n = len(raw) // (width * nch)            # a truncated file reports more frames than it holds
    if n < 1:
        raise ValueError("%s holds no audio frames" % path)
    return PcmFile(np.frombuffer(raw, dtype=np.uint8, count=n * width * nch).copy(), int(sr), int(nch), int(width), int(n))


def num_frames_16k(frames: int, sample_rate: int) -> int:
    return int(_lib.load().sylber_ingest_num_frames(int(frames), int(sample_rate)))


def ingest_pcm(pcm: PcmFile, device, normalize: bool = True) -> torch.Tensor:
    """raw PCM -> device tensor ``[channels, N16k]`` float32 (decode, resample to 16 kHz, normalise on the GPU)."""
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise _lib.SylberHipError("no MI355X visible to PyTorch-ROCm; the HIP path has no CPU fallback")
    dev = torch.device(device)
    n_out = num_frames_16k(pcm.frames, pcm.sample_rate)
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
    return ingest_pcm(read_pcm(path), device, normalize)
