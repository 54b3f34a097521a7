"""GPU tier, fp32 parity mode (precision="fp32": exact-fp32 MFMA contractions, erf-GELU): hidden states
within 1e-4 max-abs of the reference's fp32 CPU goldens and — because that is far inside the decision
margins of the (perturbation-checked) golden fixtures — END-TO-END segment tables bit-identical to the
reference's (sylber/model/sylber.py:63-138 run on the CPU in tools/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch

from sylber_amd.synth import syllable_wave
from sylber_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
FP32_TOL = 1e-4


@pytest.fixture(scope="module")
def sd():
    return synthetic_state_dict(0)


@pytest.fixture(scope="module")
def S32(sd):
    from sylber_amd import Segmenter
    return Segmenter(model_ckpt=sd, precision="fp32")


def test_stages_fp32(S32, golden_dir):
    g = np.load(os.path.join(golden_dir, "encoder_stages.npz"))
    enc = S32.speech_model
    wav = torch.from_numpy(g["wav"]).cuda()
    lengths = [int(x) for x in g["lengths"]]
    conv = enc.forward(wav, lengths, stop_stage=1).cpu().numpy()
    assert np.abs(conv - g["conv6"].transpose(0, 2, 1)).max() < FP32_TOL
    assert np.abs(enc.forward(wav, lengths, stop_stage=2).cpu().numpy() - g["enc_in"]).max() < FP32_TOL
    for l, key in [(0, "layer0"), (4, "layer4")]:
        assert np.abs(enc.forward(wav, lengths, stop_stage=3 + l).cpu().numpy() - g[key]).max() < FP32_TOL, key
    assert np.abs(enc.forward(wav, lengths).cpu().numpy() - g["layer8"]).max() < FP32_TOL


def test_e2e_segments_bit_identical_to_reference(S32, golden_dir):
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    x = torch.from_numpy(g["sample_pcm"].astype(np.float32) / 32768.0)[None]
    x = (x - x.mean()) / x.std()
    out = S32(wav=x, in_second=False)
    assert np.abs(out["hidden_states"] - g["sample_hidden"]).max() < FP32_TOL
    assert out["segments"].dtype == np.int64 and np.array_equal(out["segments"], g["sample_segments"])
    assert np.abs(out["segment_features"] - g["sample_features"]).max() < FP32_TOL
    assert np.array_equal(S32(wav=x, in_second=True)["segments"], g["sample_segments_sec"])
    wl = [syllable_wave(int(n), int(s)) for n, s in zip(g["batch_lengths"], g["batch_seeds"])]
    outs = S32(wav=wl, in_second=False)
    for i, o in enumerate(outs):
        assert np.abs(o["hidden_states"] - g[f"batch{i}_hidden"]).max() < FP32_TOL
        assert np.array_equal(o["segments"], g[f"batch{i}_segments"])
        assert np.abs(o["segment_features"] - g[f"batch{i}_features"]).max() < FP32_TOL


def test_sylber_segment_golden_bit_identical_segments(golden_dir):
    """row N2 against the reference's own ``Sylber.segment`` output (tests/golden/sylber_segment.npz): in the fp32
    parity mode the tensor-native ``Segmenter.segment`` returns bit-identical segment tables and the zero-padded
    averaged features within fp32 accumulation noise"""
    import os
    import torch
    from sylber_amd import Segmenter
    from sylber_amd.synth import syllable_wave
    from sylber_amd.weights import synthetic_state_dict
    g = np.load(os.path.join(golden_dir, "sylber_segment.npz"))
    lens = [int(x) for x in g["lens"]]
    batch = torch.zeros(2, max(lens))
    mask = torch.zeros(2, max(lens), dtype=torch.long)
    for i, (n, s) in enumerate(zip(lens, g["seeds"])):
        batch[i, :n] = syllable_wave(n, int(s))[0]
        mask[i, :n] = 1
    S = Segmenter(model_ckpt=synthetic_state_dict(0), precision="fp32")
    feats, segments, avg_fts = S.segment(input_values=batch, attention_mask=mask, mergethreshold=0.8, normthreshold=2.6)
    assert np.abs(feats[:, [0, -1]].cpu().numpy() - g["hidden_first_last"]).max() < 1e-4
    for i in range(2):
        assert np.array_equal(segments[i], g["segments%d" % i])
    a = avg_fts.cpu().numpy()
    assert a.shape == g["avg_fts"].shape
    assert np.abs(a - g["avg_fts"]).max() < 1e-4
    assert np.array_equal(a == 0.0, g["avg_fts"] == 0.0)
