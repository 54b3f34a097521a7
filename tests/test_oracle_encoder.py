"""Oracle pinning, CPU only: oracle/hubert_ref.py against per-stage golden activations produced by
the reference's HubertModel (sylber/model/sylber.py:122) in tools/gen_golden.py."""
import os

import numpy as np
import pytest
import torch

from oracle import hubert_ref
from oracle.segmenter_ref import SegmenterRef
from sylber_amd.synth import syllable_wave
from sylber_amd.weights import synthetic_state_dict, expected_shapes

FP32_TOL = 2e-5  # fp32 run-to-run floor of the reference is ~4e-6


@pytest.fixture(scope="module")
def sd():
    return synthetic_state_dict(0)


def test_schema_matches_appendix_a(sd):
    shapes = expected_shapes()
    assert set(sd) == set(shapes)
    assert sum(v.numel() for v in sd.values()) == 73108096
    for k, v in sd.items():
        assert tuple(v.shape) == shapes[k]


def test_frame_count_formula():
    assert hubert_ref.conv_out_lengths(160000) == [31999, 15999, 7999, 3999, 1999, 999, 499]
    assert hubert_ref.num_frames(46080) == 143
    assert hubert_ref.num_frames(960000) == 2999


def test_flops_match_survey():
    f = hubert_ref.flops_per_clip(160000)
    assert abs(f["total"] / 1e9 - 124.65) < 0.05
    assert abs(f["layer"] / 1e9 - 7.829) < 0.01


def test_stages_match_reference_goldens(sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "encoder_stages.npz"))
    out = hubert_ref.forward(sd, torch.from_numpy(g["wav"]), [int(x) for x in g["lengths"]], collect=True)
    for k in ("conv6", "enc_in", "layer0", "layer4", "layer8"):
        assert np.abs(out[k].numpy() - g[k]).max() < FP32_TOL, k


def test_e2e_matches_reference_goldens(sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    S = SegmenterRef(sd)
    x = torch.from_numpy(g["sample_pcm"].astype(np.float32) / 32768.0)[None]
    x = (x - x.mean()) / x.std()
    out = S(x, in_second=False)
    assert out["hidden_states"].shape == (143, 768)
    assert np.abs(out["hidden_states"] - g["sample_hidden"]).max() < FP32_TOL
    assert np.array_equal(out["segments"], g["sample_segments"])
    assert np.abs(out["segment_features"] - g["sample_features"]).max() < FP32_TOL
    sec = S(x, in_second=True)["segments"]
    assert sec.dtype == np.float64 and np.array_equal(sec, g["sample_segments_sec"])
    wl = [syllable_wave(int(n), int(s)) for n, s in zip(g["batch_lengths"], g["batch_seeds"])]
    outs = S(wl, in_second=False)
    for i, r in enumerate(outs):
        assert np.abs(r["hidden_states"] - g[f"batch{i}_hidden"]).max() < FP32_TOL
        assert np.array_equal(r["segments"], g[f"batch{i}_segments"])


def test_sylber_segment_golden_matches_oracle(golden_dir):
    """row N2: the reference's own ``Sylber.segment`` (sylber.py:208-247) on a ragged two-clip batch
    (tools/gen_golden_n2.py) vs the oracle path (hubert_ref forward + C get_segment + mean-pool)"""
    import os
    import numpy as np
    import torch
    from oracle import hubert_ref, segment_oracle
    from sylber_amd.synth import syllable_wave
    from sylber_amd.weights import synthetic_state_dict
    g = np.load(os.path.join(golden_dir, "sylber_segment.npz"))
    lens = [int(x) for x in g["lens"]]
    batch = torch.zeros(2, max(lens))
    for i, (n, s) in enumerate(zip(lens, g["seeds"])):
        batch[i, :n] = syllable_wave(n, int(s))[0]
    h = hubert_ref.forward(synthetic_state_dict(0), batch, lens)["hidden"].numpy()
    assert np.abs(h[:, [0, -1]] - g["hidden_first_last"]).max() < 2e-5
    for i in range(2):
        seg = segment_oracle.get_segment(h[i], 2.6, 0.8)
        assert np.array_equal(seg, g["segments%d" % i])
        pooled = segment_oracle.mean_pool(h[i], seg)
        # the reference pools with torch's mean here (sylber.py:235), numpy-order in Segmenter.__call__: last-ulp apart
        assert np.abs(pooled - g["avg_fts"][i, : len(seg)]).max() < 2e-5
        assert np.all(g["avg_fts"][i, len(seg):] == 0.0)                  # pad_sequence zero padding (:243)
