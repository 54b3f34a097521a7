"""TEST ORACLE (not product code) for rows N3 / N4.

N4 ``km_indices`` restates KMQuantizer.get_indices (sylber/model/quantizer.py:98-111).  The look-up itself lives in
vector_quantize_pytorch (pinned 1.17.8 in requirements.txt, ABSENT from the image -> PARITY UNPINNED for it): its
EuclideanCodebook computes ``dist = -cdist(x, embed)`` and takes ``argmax`` — i.e. the nearest centroid in L2, first
index on ties.  Restated here in float64.

N3 ``mlp_forward`` / ``resynth_front`` restate ``RFF`` / ``MLP`` (sylber/model/segment_synthesis.py:17-53) and
``resynthesize`` lines 103-140 with the torch ops the reference uses; ``mlp_forward`` is pinned against the reference's
own ``MLP`` class by tests/golden/mlp_front.npz (tools/gen_golden.py imports it in the survey container)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import segment_oracle


def km_indices(token: np.ndarray, centroids: np.ndarray, normalize: bool = False) -> np.ndarray:
    x = np.asarray(token, np.float32)
    if normalize:                                                  # quantizer.py:104-105
        x = x / np.sqrt((x.astype(np.float32) ** 2).sum(-1, dtype=np.float32) + np.float32(1e-8))[..., None] * np.float32(6)
    x64, c64 = x.astype(np.float64), np.asarray(centroids, np.float64)
    d2 = (x64 ** 2).sum(-1)[:, None] - 2.0 * x64 @ c64.T + (c64 ** 2).sum(-1)[None, :]
    return d2.argmin(-1), d2


def mlp_forward(sd, x: torch.Tensor) -> torch.Tensor:
    """MLP.forward (segment_synthesis.py:52-53) in eval mode (dropout = identity)"""
    n = max(int(k.split(".")[1]) for k in sd if k.startswith("mlp.")) // 2
    for i in range(n):
        x = F.linear(x, sd["mlp.%d.weight" % (2 * i)], sd["mlp.%d.bias" % (2 * i)])                       # :42
        p = "mlp.%d." % (2 * i + 1)
        x2 = F.linear(F.relu(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), sd[p + "linear2.weight"], sd[p + "linear2.bias"])  # :28
        x = F.layer_norm(x + x2, (x.shape[-1],), sd[p + "norm.weight"], sd[p + "norm.bias"])              # :29-30
    return F.linear(x, sd["mlp.%d.weight" % (2 * n)], sd["mlp.%d.bias" % (2 * n)])                        # :46


def resynth_front(sd, hidden: torch.Tensor, normthreshold: float, merge_threshold: float = 0.8, centroids=None,
                  normalize: bool = False):
    """segment_synthesis.py:106-139: returns (input, averaged_target_hidden_states, segments).  ``centroids`` [K,768]
    switches on the quantizer branch (:121-125): every segment mean is replaced by the codebook entry of its nearest
    centroid (``get_indices`` :98-111 -> ``get_output_from_indices``)."""
    norms = ((hidden ** 2).sum(-1) + 1e-8) ** .5                                                            # :110
    segments = [segment_oracle.get_segment(s.numpy(), normthreshold, merge_threshold) for s in hidden]      # :112
    avg = torch.zeros_like(hidden)                                                                          # :115
    for b in range(len(hidden)):
        for s, e in segments[b].reshape(-1, 2):
            ft = hidden[b][s:e].mean(0)                                                                     # :121
            if centroids is not None:                                                                       # :122-125
                idx, _ = km_indices(ft.numpy()[None], centroids, normalize)
                ft = torch.from_numpy(np.asarray(centroids, np.float32)[idx[0]])
            avg[b][s:e] = ft                                                                                # :126
    inp = mlp_forward(sd, avg)                                                                              # :138
    inp[norms < normthreshold] = 0.0                                                                        # :139
    return inp, avg, segments


def resynth_front_features(sd, features: torch.Tensor) -> torch.Tensor:
    """the ``features is not None`` branch, segment_synthesis.py:135-139: no 1e-8 under the root, threshold 1e-4"""
    norms = ((features ** 2).sum(-1)) ** .5                                                                 # :136
    inp = mlp_forward(sd, features)                                                                         # :138
    inp[norms < 1e-4] = 0.0                                                                                 # :137,139
    return inp
