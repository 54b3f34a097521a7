"""GPU tier, precision="split16" (every MFMA operand a hi / lo pair of IEEE halves, three fp16 MFMA passes per
contraction into one fp32 accumulator, erf GELU): an fp32-GRADE mode at about three times the fp16 step.  Bars: every
stage within the fp32 parity mode's tolerance of the REFERENCE's goldens, end-to-end segment tables bit-identical to the
reference's on every golden (sylber/model/sylber.py:63-138 run on the CPU by tools/gen_golden.py), and on synthetic clips
the tables of this library's fp32 mode (north_star: "segment boundaries bit-identical to the reference")."""
import os

import numpy as np
import pytest
import torch

from sylber_amd.synth import syllable_wave
from sylber_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
SPLIT_TOL = 1e-4          # the fp32 parity mode's tolerance (tests/test_gpu_fp32_parity.py)


@pytest.fixture(scope="module")
def sd():
    return synthetic_state_dict(0)


@pytest.fixture(scope="module")
def S(sd):
    from sylber_amd import Segmenter
    return Segmenter(model_ckpt=sd, precision="split16")


def test_stages_split16(S, golden_dir):
    g = np.load(os.path.join(golden_dir, "encoder_stages.npz"))
    enc = S.speech_model
    wav = torch.from_numpy(g["wav"]).cuda()
    lengths = [int(x) for x in g["lengths"]]
    conv = enc.forward(wav, lengths, stop_stage=1).cpu().numpy()
    assert np.abs(conv - g["conv6"].transpose(0, 2, 1)).max() < SPLIT_TOL
    assert np.abs(enc.forward(wav, lengths, stop_stage=2).cpu().numpy() - g["enc_in"]).max() < SPLIT_TOL
    for l, key in [(0, "layer0"), (4, "layer4")]:
        assert np.abs(enc.forward(wav, lengths, stop_stage=3 + l).cpu().numpy() - g[key]).max() < SPLIT_TOL, key
    assert np.abs(enc.forward(wav, lengths).cpu().numpy() - g["layer8"]).max() < SPLIT_TOL


def test_e2e_segments_bit_identical_to_reference(S, golden_dir):
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    x = torch.from_numpy(g["sample_pcm"].astype(np.float32) / 32768.0)[None]
    x = (x - x.mean()) / x.std()
    out = S(wav=x, in_second=False)
    assert np.abs(out["hidden_states"] - g["sample_hidden"]).max() < SPLIT_TOL
    assert out["segments"].dtype == np.int64 and np.array_equal(out["segments"], g["sample_segments"])
    assert np.abs(out["segment_features"] - g["sample_features"]).max() < SPLIT_TOL
    wl = [syllable_wave(int(n), int(s)) for n, s in zip(g["batch_lengths"], g["batch_seeds"])]
    outs = S(wav=wl, in_second=False)
    for i, o in enumerate(outs):
        assert np.abs(o["hidden_states"] - g[f"batch{i}_hidden"]).max() < SPLIT_TOL
        assert np.array_equal(o["segments"], g[f"batch{i}_segments"])
        assert np.abs(o["segment_features"] - g[f"batch{i}_features"]).max() < SPLIT_TOL


def test_tables_as_the_fp32_mode_on_synthetic_clips(S, sd):
    """64 synthetic 10 s clips (the generator of bench.py --agreement-clips): the segment tables of the exact fp32 mode,
    clip for clip (the 16-bit modes reproduce 15 % (bf16) / 74 % (fp16) of them)"""
    from sylber_amd.agreement import segment_agreement
    r = segment_agreement(sd, S.speech_model, 64)
    assert r["hidden_rel_rms_vs_fp32"] < 2e-5, r
    assert r["tables_identical"] >= 63, r
    assert r["boundary_recall"] > 0.9995 and r["boundary_precision"] > 0.9995, r


def test_full_batch_reproducible_and_batch_independent(S):
    """32 x 10 s (the persistent 256x256 kernel with its doubled store count at the seams, all tile shapes): bitwise
    reproducible, rows independent of their batch mates"""
    from sylber_amd.synth import noise_batch
    enc = S.speech_model
    x = noise_batch(32, 160000, seed=3).cuda()
    out = enc.forward(x)
    assert bool(torch.isfinite(out).all())
    for _ in range(2):
        assert torch.equal(enc.forward(x), out)
    assert torch.equal(enc.forward(x[8:12].contiguous()), out[8:12])
