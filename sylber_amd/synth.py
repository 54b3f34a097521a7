"""Synthetic 16 kHz inputs (there is no network for datasets): random-noise batches for the
throughput benchmark (BASELINE.json configs[1..3]) and a seeded "syllable-like" waveform
generator (bursts of harmonic sound separated by near-silence) that makes the segmenter emit
several segments per second, used by parity tests and golden fixtures."""
from __future__ import annotations

import math

import torch


def noise_batch(batch: int, samples: int, seed: int = 0) -> torch.Tensor:
    """cfg2/cfg3 generator: ``randn(B, N)`` — zero-mean/unit-std like the output of the
    per-file normalisation at sylber/model/sylber.py:86."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, samples, generator=g, dtype=torch.float32)


def syllable_wave(samples: int, seed: int = 0, rate: int = 16000) -> torch.Tensor:
    """[1, samples] float32, normalised to zero mean / unit std (the `wav=` entry point of the
    reference applies no normalisation itself, sylber.py:88-91)."""
    g = torch.Generator().manual_seed(seed)

    def u(lo, hi):
        return lo + (hi - lo) * torch.rand((), generator=g).item()

    t = torch.arange(samples, dtype=torch.float32) / rate
    x = torch.zeros(samples)
    pos = int(u(0.0, 0.08) * rate)
    while pos < samples:
        dur = int(u(0.09, 0.32) * rate)
        end = min(samples, pos + dur)
        n = end - pos
        if n > 32:
            f0 = u(90.0, 260.0)
            seg = torch.zeros(n)
            tt = t[:n]
            nh = int(u(3, 9))
            for h in range(1, nh + 1):
                amp = u(0.2, 1.0) / h ** u(0.3, 1.2)
                seg += amp * torch.sin(2 * math.pi * f0 * h * tt + u(0, 6.28))
            # formant-ish noise band
            noise = torch.randn(n, generator=g)
            k = int(u(2, 24))
            noise = torch.nn.functional.avg_pool1d(noise[None, None], k, 1, padding=k // 2)[0, 0, :n]
            seg += u(0.05, 0.6) * noise
            env = torch.sin(math.pi * torch.arange(n) / n) ** u(0.3, 1.0)
            x[pos:end] += u(0.4, 1.2) * env * seg
        pos = end
        if u(0, 1) < 0.45:
            pos += int(u(0.03, 0.25) * rate)
    x += 0.004 * torch.randn(samples, generator=g)
    x = (x - x.mean()) / x.std()
    return x[None, :]
