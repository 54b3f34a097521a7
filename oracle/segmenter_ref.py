"""ORACLE (test infrastructure): CPU restatement of ``Segmenter.__call__`` for tensor inputs
(sylber/model/sylber.py:88-138): pad to the batch max, frame mask, HuBERT-9L forward
(oracle/hubert_ref.py), per-utterance get_segment + mean-pool (oracle/segment_ref.c).
This is what bench.py times as ``cpu_baseline`` (kind "port") and what smoke() checks against."""
from __future__ import annotations

from typing import Dict, List, Sequence, Union

import numpy as np
import torch

from . import hubert_ref, segment_oracle


class SegmenterRef:
    def __init__(self, state_dict: Dict[str, torch.Tensor], encoding_layer: int = 9, merge_threshold: float = 0.8,
                 norm_threshold: float = 2.6):
        self.sd = {k: v.float() for k, v in state_dict.items()}
        self.encoding_layer = encoding_layer
        self.merge_threshold = merge_threshold
        self.norm_threshold = norm_threshold

    def encode(self, wavs: Sequence[torch.Tensor]) -> np.ndarray:
        lengths = [int(w.shape[1]) for w in wavs]
        lmax = max(lengths)
        batch = torch.zeros(len(wavs), lmax)
        for i, w in enumerate(wavs):
            batch[i, : lengths[i]] = w[0]
        # sylber.py:99-115 builds an attention mask for every utterance (all ones when unpadded)
        with torch.no_grad():
            h = hubert_ref.forward(self.sd, batch, lengths, num_layers=self.encoding_layer)["hidden"]
        return h.numpy()

    def __call__(self, wav: Union[torch.Tensor, List[torch.Tensor]], in_second: bool = True):
        is_batch = isinstance(wav, list)
        wavs = wav if is_batch else [wav]
        hidden = self.encode(wavs)
        outs = []
        for states in hidden:
            seg = segment_oracle.get_segment(states, self.norm_threshold, self.merge_threshold)
            outs.append({
                "segments": seg * 1.0 / 50 if in_second else seg,
                "segment_features": segment_oracle.mean_pool(states, seg) if len(seg) > 0 else np.array([]),
                "hidden_states": states,
            })
        return outs if is_batch else outs[0]
