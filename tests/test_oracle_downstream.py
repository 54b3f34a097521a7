"""CPU tier, rows N3 / N4: the oracle's conditioner MLP against the golden produced by the reference's own ``MLP``
class (tools/gen_golden_mlp.py), and the nearest-centroid restatement against brute force."""
import os

import numpy as np
import torch

from oracle import downstream_ref as R
from sylber_amd.weights import synthetic_mlp_state_dict


def test_mlp_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "mlp_front.npz"))
    sd = synthetic_mlp_state_dict(0)
    y = R.mlp_forward(sd, torch.from_numpy(g["x"])).numpy()
    assert y.shape == g["y"].shape == (24, 256)
    assert np.abs(y - g["y"]).max() <= 1e-5
    assert float(g["oracle_vs_reference_max_abs"]) < 1e-5
    # key layout of MLP.state_dict() for 768 -> [512, 512] -> 256 (sylber_configs/sylber_resynthesis.yaml)
    assert sd["mlp.0.weight"].shape == (512, 768) and sd["mlp.1.linear1.weight"].shape == (512, 512)
    assert sd["mlp.3.norm.weight"].shape == (512,) and sd["mlp.4.weight"].shape == (256, 512)


def test_km_indices_brute_force():
    rng = np.random.default_rng(0)
    c = rng.standard_normal((37, 48)).astype(np.float32)
    x = rng.standard_normal((50, 48)).astype(np.float32)
    idx, d2 = R.km_indices(x, c)
    brute = np.array([np.argmin([np.sum((xi.astype(np.float64) - cj) ** 2) for cj in c]) for xi in x])
    assert np.array_equal(idx, brute)
    x[3] = c[11]
    assert R.km_indices(x, c)[0][3] == 11
    # normalisation (quantizer.py:104-105) maps every token onto the radius-6 sphere first
    idxn, _ = R.km_indices(x * 100.0, c, normalize=True)
    xs = x / np.linalg.norm(x, axis=-1, keepdims=True) * 6
    assert np.array_equal(idxn, R.km_indices(xs.astype(np.float32), c)[0])


def test_resynth_front_semantics():
    from sylber_amd.synth_states import syllable_states
    sd = synthetic_mlp_state_dict(1)
    h = torch.from_numpy(np.stack([syllable_states(60, 3), syllable_states(60, 4)]))
    inp, avg, segs = R.resynth_front(sd, h, 2.6, 0.8)
    assert inp.shape == (2, 60, 256) and avg.shape == h.shape
    norms = np.sqrt((h.numpy().astype(np.float64) ** 2).sum(-1))
    assert np.all(inp.numpy()[norms < 2.59] == 0.0)
    for b in range(2):
        covered = np.zeros(60, bool)
        for s, e in segs[b].reshape(-1, 2):
            covered[s:e] = True
            assert np.allclose(avg[b, s:e].numpy(), h[b, s:e].mean(0).numpy()[None], atol=1e-6)
        assert np.all(avg[b].numpy()[~covered] == 0.0)


def test_km_indices_independent_cross_check_cdist():
    """vector_quantize_pytorch is absent (N4 parity stays UNPINNED); EuclideanCodebook quantises with
    ``dist = -torch.cdist(x, embed); embed_ind = dist.argmax(-1)`` -- evaluated here with torch.cdist itself and compared
    with the oracle's float64 arg-min: identical indices except on numerical ties"""
    import torch
    from oracle import downstream_ref as R
    rng = np.random.default_rng(4)
    c = rng.standard_normal((2000, 768)).astype(np.float32)
    x = (c[rng.integers(0, 2000, 400)] + 0.8 * rng.standard_normal((400, 768))).astype(np.float32)
    for normalize in (False, True):
        xi = x
        if normalize:
            xi = x / np.sqrt((x ** 2).sum(-1) + np.float32(1e-8))[:, None] * np.float32(6)
        ref = (-torch.cdist(torch.from_numpy(xi)[None], torch.from_numpy(c)[None]))[0].argmax(-1).numpy()
        got, d2 = R.km_indices(x, c, normalize)
        diff = np.nonzero(ref != got)[0]
        for r in diff:
            assert abs(d2[r, ref[r]] - d2[r, got[r]]) <= 1e-4 * abs(d2[r, got[r]])
        assert len(diff) <= 2


def resynth_golden_inputs():
    """the seeded inputs tools/gen_golden_resynth.py fed to the reference's own SegmentSynthesis.resynthesize"""
    from sylber_amd.synth_states import syllable_states
    h = torch.from_numpy(np.stack([syllable_states(120, 3), syllable_states(120, 4), syllable_states(120, 5, mode="silence"),
                                   syllable_states(120, 6, mode="edge")]))
    cent = (np.random.default_rng(3).standard_normal((500, 768)) * 0.25).astype(np.float32)
    g = torch.Generator().manual_seed(5)
    f = torch.randn(2, 37, 768, generator=g)
    f[0, 3] = 0.0
    f[1, 10] = 5e-6
    f[1, 11] = 3e-6
    return h, cent, f


def test_resynth_front_matches_the_references_own_resynthesize(golden_dir):
    """row N3's ORCHESTRATION pinned (VERDICT r4 missing #4): tests/golden/resynth_front.npz holds what the reference's own
    ``SegmentSynthesis.resynthesize`` (segment_synthesis.py:103-146, run unmodified by tools/gen_golden_resynth.py) returned for
    seeded hidden states -- segment tables, the conditioning input with its silence mask, the quantiser hook's broadcast, the
    ``features=`` branch -- and the oracle restatement reproduces all of it"""
    import os
    g = np.load(os.path.join(golden_dir, "resynth_front.npz"))
    sd = synthetic_mlp_state_dict(1)
    h, cent, f = resynth_golden_inputs()
    inp, _, segs = R.resynth_front(sd, h, 2.6, 0.8)
    assert np.array_equal(np.array([len(s.reshape(-1, 2)) for s in segs], np.int32), g["nseg"])
    assert np.array_equal(np.concatenate([s.reshape(-1, 2) for s in segs], 0), g["segments"])
    assert np.array_equal(inp.numpy() == 0.0, g["cond"] == 0.0) and np.abs(inp.numpy() - g["cond"]).max() < 1e-5
    inq, _, _ = R.resynth_front(sd, h, 2.6, 0.8, centroids=cent)
    assert np.abs(inq.numpy() - g["cond_quantized"]).max() < 1e-5
    assert np.abs(R.resynth_front_features(sd, f).numpy() - g["cond_features"]).max() < 1e-5
    assert float(g["oracle_vs_reference_max_abs"]) < 1e-5
