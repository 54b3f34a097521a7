"""GPU tier, rows N3 / N4 through the C-ABI: k-means tokenisation and the resynthesis conditioner front half on the
device-resident outputs of the segmenter, against the CPU oracle (which is pinned to the reference's MLP class)."""
import os

import numpy as np
import pytest
import torch

from oracle import downstream_ref as R
from sylber_amd.synth_states import syllable_states
from sylber_amd.weights import synthetic_mlp_state_dict, synthetic_state_dict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,K,normalize", [(1, 64, False), (77, 1000, False), (300, 5000, True), (50, 1002, False)])
def test_km_assign_matches_oracle(n, K, normalize):
    from sylber_amd.downstream import KMQuantizer
    rng = np.random.default_rng(n + K)
    c = rng.standard_normal((K, 768)).astype(np.float32)
    x = (c[rng.integers(0, K, n)] + 0.7 * rng.standard_normal((n, 768))).astype(np.float32)
    x[0] = c[K - 1]                                              # exact hit on the last centroid
    q = KMQuantizer(c, normalize=normalize)
    idx = q.get_indices(torch.from_numpy(x)[None]).cpu().numpy()  # [1, n, 1] like the reference's (B, L, d) tokens
    assert idx.shape == (1, n, 1) and idx.dtype == np.int64
    exp, d2 = R.km_indices(x, c, normalize)
    got = idx[0, :, 0]
    same = got == exp
    # fp32 contraction vs the float64 oracle: a different index is only acceptable on a numerical tie
    for r in np.nonzero(~same)[0]:
        assert abs(d2[r, got[r]] - d2[r, exp[r]]) <= 1e-4 * abs(d2[r, exp[r]]) + 1e-4
    assert same.mean() > 0.99
    if not normalize:
        assert got[0] == K - 1
    dec = q.decode(torch.from_numpy(idx)).cpu().numpy()
    assert dec.shape == (1, n, 768) and np.array_equal(dec[0], c[got])
    assert np.array_equal(q.decode(torch.tensor([[-5]])).cpu().numpy()[0], c[0])      # clip(0), quantizer.py:129


def test_conditioner_mlp_matches_reference_golden(golden_dir):
    """the MLP alone, fed through the front half as 24 one-frame 'segments' of one utterance"""
    from sylber_amd.downstream import SegmentConditioner
    g = np.load(os.path.join(golden_dir, "mlp_front.npz"))
    x = torch.from_numpy(g["x"]).cuda()                         # [24, 768]
    T = x.shape[0]
    cond = SegmentConditioner(synthetic_mlp_state_dict(0))
    hidden = (x * 0 + 10.0)[None].contiguous()                  # loud frames: nothing masked
    seg = torch.stack([torch.arange(T), torch.arange(T) + 1], -1)[None].to("cuda", torch.int64).contiguous()
    nseg = torch.tensor([T], dtype=torch.int32, device="cuda")
    out, avg = cond(hidden, seg, nseg, x[None].contiguous(), normthreshold=2.6)
    assert out.shape == (1, T, 256)
    assert np.abs(out[0].cpu().numpy() - g["y"]).max() <= 2e-4 * np.abs(g["y"]).max()
    assert np.array_equal(avg[0].cpu().numpy(), g["x"])


def test_resynth_front_end_to_end():
    from sylber_amd import HubertEncoderHIP
    from sylber_amd.downstream import SegmentConditioner
    msd = synthetic_mlp_state_dict(1)
    enc = HubertEncoderHIP(synthetic_state_dict(0))
    cond = SegmentConditioner({"input_model." + k: v for k, v in msd.items()})      # checkpoint-style prefix
    h = np.stack([syllable_states(120, 3), syllable_states(120, 4), syllable_states(120, 5, mode="silence")])
    hd = torch.from_numpy(h).cuda()
    seg, nseg, feats = enc.segment(hd, 2.6, 0.8)
    inp, avg = cond(hd, seg, nseg, feats, normthreshold=2.6)
    exp_inp, exp_avg, exp_segs = R.resynth_front(msd, torch.from_numpy(h), 2.6, 0.8)
    for b in range(3):
        n = int(nseg[b])
        assert np.array_equal(seg[b, :n].cpu().numpy().reshape(-1, 2), exp_segs[b].reshape(-1, 2))
    assert int(nseg[2]) == 0
    # averaged states: numpy-order mean (segmenter) vs torch mean (reference line :121) differ in the last ulp
    assert np.abs(avg.cpu().numpy() - exp_avg.numpy()).max() <= 1e-5
    got, exp = inp.cpu().numpy(), exp_inp.numpy()
    assert got.shape == exp.shape == (3, 120, 256)
    assert np.array_equal(got == 0.0, exp == 0.0)                # identical silence mask (no frame within an ulp of 2.6 here)
    assert np.abs(got - exp).max() <= 2e-4 * np.abs(exp).max()
    # a smaller segment-slot budget than frames gives the same answer
    inp2, _ = cond(hd, seg, nseg, feats, normthreshold=2.6, max_segments=int(nseg.max()) + 3)
    assert torch.equal(inp, inp2)


def test_resynth_front_features_branch():
    """segment_synthesis.py:135-139: caller-supplied frame features; silent = norm WITHOUT the 1e-8 below 1e-4, so an
    all-zero frame IS masked here (the hidden-state branch, with its 1e-8, would keep it)"""
    from sylber_amd.downstream import SegmentConditioner
    msd = synthetic_mlp_state_dict(2)
    cond = SegmentConditioner(msd)
    g = torch.Generator().manual_seed(5)
    f = torch.randn(2, 37, 768, generator=g)
    f[0, 3] = 0.0                                                # silent frame
    f[1, 10] = 5e-6                                              # norm 1.4e-4: just above the threshold
    f[1, 11] = 3e-6                                              # norm 8.3e-5: below
    got = cond.from_features(f.cuda()).cpu().numpy()
    exp = R.resynth_front_features(msd, f).numpy()
    assert got.shape == exp.shape == (2, 37, 256)
    assert np.array_equal(got == 0.0, exp == 0.0)
    assert (got[0, 3] == 0).all() and (got[1, 11] == 0).all() and (got[1, 10] != 0).any()
    assert np.abs(got - exp).max() <= 2e-4 * np.abs(exp).max()


@pytest.mark.parametrize("normalize", [False, True])
def test_resynth_front_with_quantizer_hook(normalize):
    """segment_synthesis.py:121-125: segment means replaced by their nearest codebook entry before the broadcast"""
    from sylber_amd import HubertEncoderHIP
    from sylber_amd.downstream import KMQuantizer, SegmentConditioner
    msd = synthetic_mlp_state_dict(1)
    enc = HubertEncoderHIP(synthetic_state_dict(0, num_layers=1), num_layers=1)
    cond = SegmentConditioner(msd)
    h = np.stack([syllable_states(150, 13), syllable_states(150, 14)])
    rng = np.random.default_rng(3)
    cent = (rng.standard_normal((500, 768)) * (0.25 if not normalize else 6.0 / np.sqrt(768))).astype(np.float32)
    hd = torch.from_numpy(h).cuda()
    seg, nseg, feats = enc.segment(hd, 2.6, 0.8)
    q = KMQuantizer(cent, normalize=normalize)
    inp, avg = cond(hd, seg, nseg, feats, normthreshold=2.6, quantizer=q)
    exp_inp, exp_avg, _ = R.resynth_front(msd, torch.from_numpy(h), 2.6, 0.8, centroids=cent, normalize=normalize)
    got_avg, e_avg = avg.cpu().numpy(), exp_avg.numpy()
    # every frame carries either zeros or one codebook row; the rows agree with the float64 nearest-centroid oracle
    # except on numerical ties (none expected with 500 random centroids)
    assert np.array_equal(got_avg, e_avg)
    got, exp = inp.cpu().numpy(), exp_inp.numpy()
    assert np.array_equal(got == 0.0, exp == 0.0)
    assert np.abs(got - exp).max() <= 2e-4 * np.abs(exp).max()


def test_resynth_front_against_the_references_own_resynthesize(golden_dir):
    """row N3 end to end against REFERENCE OUTPUT (tests/golden/resynth_front.npz: the reference's own
    SegmentSynthesis.resynthesize on seeded hidden states, tools/gen_golden_resynth.py): segment tables bit-exact, the conditioning
    input with the reference's silence mask, the quantiser hook's broadcast, and the ``features=`` branch"""
    from test_oracle_downstream import resynth_golden_inputs         # (the seeded inputs of the golden: tests/ is on sys.path under pytest)
    from sylber_amd import HubertEncoderHIP
    from sylber_amd.downstream import KMQuantizer, SegmentConditioner
    g = np.load(os.path.join(golden_dir, "resynth_front.npz"))
    msd = synthetic_mlp_state_dict(1)
    h, cent, f = resynth_golden_inputs()
    enc = HubertEncoderHIP(synthetic_state_dict(0, num_layers=1), num_layers=1)
    cond = SegmentConditioner(msd)
    hd = h.cuda()
    seg, nseg, feats = enc.segment(hd, 2.6, 0.8)
    assert np.array_equal(nseg.cpu().numpy(), g["nseg"])
    tables = np.concatenate([seg[b, :int(nseg[b])].cpu().numpy().reshape(-1, 2) for b in range(4)], 0)
    assert np.array_equal(tables, g["segments"])
    inp, _ = cond(hd, seg, nseg, feats, normthreshold=2.6)
    got = inp.cpu().numpy()
    assert np.array_equal(got == 0.0, g["cond"] == 0.0)
    assert np.abs(got - g["cond"]).max() <= 2e-4 * np.abs(g["cond"]).max()
    inq, _ = cond(hd, seg, nseg, feats, normthreshold=2.6, quantizer=KMQuantizer(cent))
    gq = inq.cpu().numpy()
    assert np.array_equal(gq == 0.0, g["cond_quantized"] == 0.0)
    assert np.abs(gq - g["cond_quantized"]).max() <= 2e-4 * np.abs(g["cond_quantized"]).max()
    gf = cond.from_features(f.cuda()).cpu().numpy()
    assert np.array_equal(gf == 0.0, g["cond_features"] == 0.0)
    assert np.abs(gf - g["cond_features"]).max() <= 2e-4 * np.abs(g["cond_features"]).max()
