"""Oracle pinning, CPU only: oracle/segment_ref.c against the golden get_segment vectors that
tools/gen_golden.py produced by running the reference (sylber/utils/segment_utils.py:72-131)."""
import os

import numpy as np
import pytest

from oracle import segment_oracle
from sylber_amd.synth_states import syllable_states


@pytest.fixture(scope="module")
def cases(golden_dir):
    return np.load(os.path.join(golden_dir, "segment_cases.npz"))


def test_np_sum_order_matches_numpy():
    rng = np.random.default_rng(0)
    for n in list(range(0, 200)) + [384, 768, 769, 1000, 2999]:
        a = (rng.standard_normal(n) * 10).astype(np.float32)
        assert segment_oracle.np_sum(a) == a.sum(), n


def test_scalar_pow_is_libm_powf():
    rng = np.random.default_rng(1)
    v = rng.uniform(1e-3, 1e3, 20000).astype(np.float32)
    for x in v:
        assert segment_oracle.powf_half(x) == x ** .5


def test_oracle_matches_reference_goldens(cases):
    off = cases["offsets"]
    n_checked = 0
    for i in range(len(cases["T"])):
        if int(cases["T"][i]) > 499 and i % 3:
            continue  # keep the CPU suite quick; all 2999-frame cases run in the gpu tier
        st = syllable_states(int(cases["T"][i]), int(cases["seed"][i]), mode=str(cases["mode"][i]))
        got = segment_oracle.get_segment(st, float(cases["norm_thr"][i]), float(cases["merge_thr"][i]))
        exp = cases["segments"][off[i]:off[i + 1]]
        if len(exp) == 0:
            assert got.shape == (0,) and got.dtype == np.float64
        else:
            assert got.dtype == np.int64 and np.array_equal(got, exp), i
            feats = segment_oracle.mean_pool(st, got)
            assert np.isclose(np.nan_to_num(feats).astype(np.float64).sum(), cases["feat_sum"][i], rtol=0, atol=1e-6)
        n_checked += 1
    assert n_checked > 600


def test_mean_pool_is_sequential_row_sum():
    st = syllable_states(143, 5)
    seg = np.array([[0, 1], [3, 20], [20, 143]], dtype=np.int64)
    exp = np.stack([st[s:e].mean(0) for s, e in seg])
    assert np.array_equal(segment_oracle.mean_pool(st, seg), exp)


def test_edge_cases_shapes():
    z = np.zeros((10, 768), dtype=np.float32)
    r = segment_oracle.get_segment(z, 2.6, 0.8)
    assert r.shape == (0,)
    one = np.ones((1, 768), dtype=np.float32)
    r = segment_oracle.get_segment(one, 2.6, 0.8)
    assert np.array_equal(r, np.array([[0, 1]]))


@pytest.mark.reference
def test_oracle_vs_live_reference():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools"))
    import ref_shim
    if not ref_shim.available():
        pytest.skip("reference checkout not present")
    _, seg_utils, _ = ref_shim.load()
    import warnings
    warnings.simplefilter("ignore")
    for seed in range(40):
        T = [5, 77, 300][seed % 3]
        st = syllable_states(T, 9000 + seed, mode=["normal", "edge", "long"][seed % 3])
        r = seg_utils.get_segment(st, 2.6, 0.8)
        o = segment_oracle.get_segment(st, 2.6, 0.8)
        assert r.shape == o.shape and np.array_equal(r, o)
