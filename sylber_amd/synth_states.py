"""Seeded synthetic hidden-state sequences [T,768] that look like SYLBER output to the
segmenter: piecewise-constant "syllable" directions with within-syllable jitter, silence gaps
with small norms, and knobs that push norms / cosines close to the thresholds.  Used by the
get_segment parity tests and golden fixtures (inputs are stored as seeds + this generator)."""
from __future__ import annotations

import numpy as np


def syllable_states(T: int, seed: int, dim: int = 768, mode: str = "normal") -> np.ndarray:
    rng = np.random.default_rng(seed)
    out = np.zeros((T, dim), dtype=np.float32)
    t = 0
    prev = None
    while t < T:
        r = rng.random()
        if mode == "silence" or (mode != "allspeech" and r < 0.2):
            n = int(rng.integers(1, 12))
            amp = rng.uniform(0.1, 0.6) if mode != "edge" else rng.uniform(2.4, 2.8) / np.sqrt(dim) * 1.0
            blk = rng.standard_normal((n, dim)) * (amp if mode == "edge" else amp / np.sqrt(dim) * 3.0)
        else:
            n = int(rng.integers(1, 20)) if mode != "long" else int(rng.integers(60, 400))
            if mode == "degenerate":
                n = 1
            centre = rng.standard_normal(dim)
            if prev is not None and rng.random() < 0.5:
                # correlated with the previous syllable: cosine near the merge threshold
                mix = rng.uniform(0.6, 0.95)
                centre = mix * prev + np.sqrt(max(1 - mix * mix, 0)) * centre
            centre /= np.linalg.norm(centre)
            prev = centre
            jitter = rng.uniform(0.1, 0.75)
            amp = rng.uniform(2.7, 9.0) if mode != "edge" else rng.uniform(2.5, 2.7)
            drift = rng.standard_normal(dim) / np.sqrt(dim)
            ramp = np.linspace(0, rng.uniform(0, 1.0), n)[:, None]
            blk = centre[None, :] + ramp * drift[None, :] + jitter * rng.standard_normal((n, dim)) / np.sqrt(dim)
            blk *= amp * rng.uniform(0.85, 1.15, size=(n, 1))
        n = min(n, T - t)
        out[t:t + n] = blk[:n].astype(np.float32)
        t += n
    return out
