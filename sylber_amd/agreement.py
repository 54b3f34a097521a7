"""How often does a fast-precision encoder reproduce the EXACT segment tables of the fp32 path?

``get_segment`` takes discontinuous decisions (norm >= 2.6, cosine >= 0.8, argmax), so a hidden-state error of a
few 1e-3 flips the frames that sit within that distance of a threshold.  The fp32 parity mode of this library
(every contraction on the exact-fp32 MFMA) reproduces the reference's tables bit for bit on every golden vector
(tests/test_gpu_fp32_parity.py), so it serves as the truth here; the segmenter kernel itself is bit-exact, i.e. every
disagreement is an encoder-precision effect.  Used by ``bench.py --agreement-clips N`` and tools/parity_report.py."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from .synth import syllable_wave


def segment_agreement(sd, engine, n_clips: int, clip_samples: int = 160000, device: str = "cuda", batch: int = 32,
                      norm_threshold: float = 2.6, merge_threshold: float = 0.8, truth=None) -> Dict[str, object]:
    from .segmenter import HubertEncoderHIP
    own_truth = truth is None
    if own_truth:
        truth = HubertEncoderHIP(sd, device=device, precision="fp32")
    tables_same = clips = 0
    b_truth = b_hit = b_got = 0
    rel_num = rel_den = 0.0
    for c0 in range(0, n_clips, batch):
        n = min(batch, n_clips - c0)
        wav = torch.cat([syllable_wave(clip_samples, 9000 + c0 + i) for i in range(n)], 0).to(device).contiguous()
        h_ref = truth.forward(wav, None)
        h = engine.forward(wav, None)
        d = (h.double() - h_ref.double())
        rel_num += float((d * d).sum())
        rel_den += float((h_ref.double() ** 2).sum())
        s_ref, n_ref, _ = truth.segment(h_ref, norm_threshold, merge_threshold, with_features=False)
        s_got, n_got, _ = engine.segment(h, norm_threshold, merge_threshold, with_features=False)
        s_ref, n_ref, s_got, n_got = s_ref.cpu().numpy(), n_ref.cpu().numpy(), s_got.cpu().numpy(), n_got.cpu().numpy()
        for i in range(n):
            a, b = s_ref[i, : n_ref[i]], s_got[i, : n_got[i]]
            clips += 1
            tables_same += int(a.shape == b.shape and np.array_equal(a, b))
            ra, rb = set(a.reshape(-1).tolist()), set(b.reshape(-1).tolist())
            b_truth += len(ra)
            b_got += len(rb)
            b_hit += len(ra & rb)
    if own_truth:
        del truth
    return {"clips": clips, "clip_seconds": clip_samples / 16000.0,
            "tables_identical": tables_same, "tables_identical_frac": round(tables_same / max(clips, 1), 4),
            "boundaries_fp32": b_truth, "boundaries_found": b_hit,
            "boundary_recall": round(b_hit / max(b_truth, 1), 5), "boundary_precision": round(b_hit / max(b_got, 1), 5),
            "hidden_rel_rms_vs_fp32": float(np.sqrt(rel_num / max(rel_den, 1e-300))),
            "truth": "this library's fp32 parity mode (bit-identical to the reference on every golden)",
            "input": "sylber_amd.synth.syllable_wave(seed 9000 + i)"}
