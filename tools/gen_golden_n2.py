"""Golden vector for row N2 (survey container only): the REFERENCE's own ``Sylber.segment``
(sylber/model/sylber.py:208-247) on a ragged two-clip batch with the seeded weights, stored in
tests/golden/sylber_segment.npz (inputs by seed, outputs: segments, zero-padded avg_fts, a hidden-state checksum).
Contains no reference code; never runs on the GPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sylber_amd.synth import syllable_wave                # noqa: E402
from sylber_amd.weights import synthetic_state_dict      # noqa: E402
from tools import ref_shim                                # noqa: E402

LENS = [30000, 22000]
SEEDS = [901, 902]


def main():
    ref, _, cfg_dir = ref_shim.load()
    torch.manual_seed(0)
    model = ref.Sylber(speech_upstream=cfg_dir, encoding_layer=9).eval()
    sd = synthetic_state_dict(0)
    missing = model.speech_model.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if "masked_spec_embed" not in k], missing
    batch = torch.zeros(2, max(LENS))
    mask = torch.zeros(2, max(LENS), dtype=torch.long)
    for i, (n, s) in enumerate(zip(LENS, SEEDS)):
        batch[i, :n] = syllable_wave(n, s)[0]
        mask[i, :n] = 1
    with torch.no_grad():
        feats, segments, avg_fts = model.segment(input_values=batch, attention_mask=mask, mergethreshold=0.8, normthreshold=2.6)
    out = {"lens": np.array(LENS), "seeds": np.array(SEEDS), "avg_fts": avg_fts.numpy(),
           "hidden_first_last": feats[:, [0, -1]].numpy(), "hidden_abs_mean": np.float64(feats.abs().mean())}
    for i, sg in enumerate(segments):
        out["segments%d" % i] = np.asarray(sg)
    print("segments per clip:", [len(s) for s in segments], "avg_fts", tuple(avg_fts.shape))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sylber_segment.npz"), **out)
    print("wrote tests/golden/sylber_segment.npz")


if __name__ == "__main__":
    main()
