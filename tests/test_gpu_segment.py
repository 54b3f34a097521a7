"""GPU tier: the HIP segmentation kernel (sylber_segment through the C-ABI) must be BIT-EXACT
against the oracle (oracle/segment_ref.c) and the golden vectors produced by the reference's
get_segment (sylber/utils/segment_utils.py:72-131) — segment indices and pooled features."""
import os

import numpy as np
import pytest
import torch

from oracle import segment_oracle
from sylber_amd.synth_states import syllable_states

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["wide", "per_utterance"])
def enc(request):
    """both implementations of sylber_segment: the wide path (round 6: frame norms / one workgroup per run of speech frames / compaction /
    pooling on all CUs) and the one-workgroup-per-utterance kernel of rounds 1-5 (SYLBER_OPT_SEGMENT = -1) -- every case below is
    checked bit for bit on each"""
    from sylber_amd import HubertEncoderHIP
    from sylber_amd.weights import synthetic_state_dict
    e = HubertEncoderHIP(synthetic_state_dict(0, num_layers=1), num_layers=1)
    e.set_option(9, 0 if request.param == "wide" else -1)
    return e


def _run(enc, states_list, nt, mt):
    x = torch.from_numpy(np.stack(states_list)).cuda()
    seg, nseg, feats = enc.segment(x, nt, mt)
    torch.cuda.synchronize()
    return seg.cpu().numpy(), nseg.cpu().numpy(), feats.cpu().numpy()


def test_golden_cases_bit_exact(enc, golden_dir):
    cases = np.load(os.path.join(golden_dir, "segment_cases.npz"))
    off = cases["offsets"]
    # group by (T, thresholds) so that every launch is a real batch
    groups = {}
    for i in range(len(cases["T"])):
        key = (int(cases["T"][i]), float(cases["norm_thr"][i]), float(cases["merge_thr"][i]))
        groups.setdefault(key, []).append(i)
    checked = 0
    for (T, nt, mt), idx in groups.items():
        sts = [syllable_states(T, int(cases["seed"][i]), mode=str(cases["mode"][i])) for i in idx]
        seg, nseg, feats = _run(enc, sts, nt, mt)
        for j, i in enumerate(idx):
            exp = cases["segments"][off[i]:off[i + 1]]
            assert nseg[j] == len(exp), (T, nt, mt, i)
            assert np.array_equal(seg[j, :nseg[j]], exp), (T, nt, mt, i)
            if len(exp):
                fe = segment_oracle.mean_pool(sts[j], exp)
                assert np.array_equal(feats[j, :nseg[j]], fe, equal_nan=True), (T, nt, mt, i)
            checked += 1
    assert checked == len(cases["T"])


def test_random_batches_vs_oracle(enc):
    for T, mode, seed0 in [(499, "normal", 100), (499, "edge", 200), (2999, "long", 300), (1000, "degenerate", 400),
                           (37, "allspeech", 500), (1500, "allspeech", 600), (700, "allspeech", 650), (513, "allspeech", 660),
                           (512, "allspeech", 670), (1, "allspeech", 680), (2, "normal", 690)]:
        # (allspeech beyond 512 frames: ONE run of speech frames longer than the wide path keeps in LDS -> its global-slab launch;
        #  "long": runs of several hundred frames on both sides of that limit in one utterance)
        sts = [syllable_states(T, seed0 + s, mode=mode) for s in range(8)]
        seg, nseg, feats = _run(enc, sts, 2.6, 0.8)
        for j, st in enumerate(sts):
            exp = segment_oracle.get_segment(st, 2.6, 0.8).reshape(-1, 2)
            assert nseg[j] == len(exp) and np.array_equal(seg[j, :nseg[j]], exp), (T, mode, j)
            if len(exp):
                assert np.array_equal(feats[j, :nseg[j]], segment_oracle.mean_pool(st, exp), equal_nan=True)
        # the slab holds nothing the kernel reads before writing it: poisoned with NaN / 0x7F patterns, same results
        from sylber_amd import _lib
        for byte in (0xFF, 0x7F):
            _lib.check(enc.lib.sylber_debug_poison_workspace(enc.handle, byte), "poison")
            seg2, nseg2, feats2 = _run(enc, sts, 2.6, 0.8)
            assert np.array_equal(nseg2, nseg)
            for j in range(len(sts)):
                assert np.array_equal(seg2[j, :nseg[j]], seg[j, :nseg[j]]) and np.array_equal(feats2[j, :nseg[j]], feats[j, :nseg[j]], equal_nan=True)


def test_properties_full_size(enc):
    """Size-independent properties at BASELINE batch size: segments sorted, disjoint, inside [0,T],
    speech frames covered, idempotent across launches."""
    sts = [syllable_states(499, 7000 + s) for s in range(32)]
    seg, nseg, _ = _run(enc, sts, 2.6, 0.8)
    seg2, nseg2, _ = _run(enc, sts, 2.6, 0.8)
    assert np.array_equal(nseg, nseg2)
    for j in range(32):
        s = seg[j, :nseg[j]]
        assert np.array_equal(s, seg2[j, :nseg2[j]])
        assert (s[:, 0] <= s[:, 1]).all() and (s[:, 0] >= 0).all() and (s[:, 1] <= 499).all()
        assert (s[1:, 0] >= s[:-1, 1]).all()
        norms = np.sqrt((sts[j] ** 2).sum(-1) + np.float32(1e-8))
        covered = np.zeros(499, bool)
        for a, b in s:
            covered[a:b] = True
        assert np.array_equal(covered, norms >= np.float32(2.6))


def test_long_utterance_global_scratch_path(enc):
    """T > 3940 frames (78.8 s): the bookkeeping moves from LDS to a global slab; same results, bit for bit
    (the reference's get_segment has no length limit, segment_utils.py:72)."""
    for T, mode, seed0 in [(3941, "long", 900), (6000, "normal", 910), (4500, "edge", 920)]:
        sts = [syllable_states(T, seed0 + s, mode=mode) for s in range(3)]
        seg, nseg, feats = _run(enc, sts, 2.6, 0.8)
        for j, st in enumerate(sts):
            exp = segment_oracle.get_segment(st, 2.6, 0.8).reshape(-1, 2)
            assert nseg[j] == len(exp) and np.array_equal(seg[j, :nseg[j]], exp), (T, mode, j)
            if len(exp):
                assert np.array_equal(feats[j, :nseg[j]], segment_oracle.mean_pool(st, exp), equal_nan=True)
