"""GPU tier: the drop-in ``Segmenter`` against the CPU oracle and the reference's golden dicts
(sylber/model/sylber.py:63-138): output contract (keys, dtypes, shapes), hidden states within the
bf16 budget, segmentation bit-exact GIVEN the hidden states the GPU produced."""
import os

import numpy as np
import pytest
import torch

from oracle import segment_oracle
from oracle.segmenter_ref import SegmenterRef
from sylber_amd.synth import syllable_wave
from sylber_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu


def rel_rms(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-30))


@pytest.fixture(scope="module")
def sd():
    return synthetic_state_dict(0)


@pytest.fixture(scope="module")
def S(sd):
    from sylber_amd import Segmenter
    return Segmenter(model_ckpt=sd)


def _check_contract(out, in_second):
    assert set(out) == {"segments", "segment_features", "hidden_states"}
    assert out["hidden_states"].dtype == np.float32 and out["hidden_states"].shape[1] == 768
    if len(out["segments"]):
        assert out["segments"].dtype == (np.float64 if in_second else np.int64)
        assert out["segments"].shape[1] == 2
        assert out["segment_features"].dtype == np.float32
        assert out["segment_features"].shape == (len(out["segments"]), 768)
    else:
        assert out["segments"].shape == (0,) and out["segment_features"].shape == (0,)


def _segments_consistent(out, S):
    """segmenter is bit-exact given the same hidden states (stage-wise contract)"""
    hs = out["hidden_states"]
    exp = segment_oracle.get_segment(hs, S.norm_threshold, S.merge_threshold)
    assert exp.shape == out["segments"].shape and np.array_equal(exp, out["segments"])
    if len(exp):
        assert np.array_equal(segment_oracle.mean_pool(hs, exp), out["segment_features"], equal_nan=True)


def test_sample_wav_golden(S, golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    # through the file entry point: write the fixture PCM as a wav file (config #1 input)
    import wave
    p = str(tmp_path / "sample.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(g["sample_pcm"].tobytes())
    out = S(p, in_second=False)
    _check_contract(out, False)
    assert out["hidden_states"].shape == (143, 768)
    assert rel_rms(out["hidden_states"], g["sample_hidden"]) < 2e-2
    _segments_consistent(out, S)
    sec = S(wav_file=p)["segments"]
    assert sec.dtype == np.float64 and np.array_equal(sec, out["segments"] * 1.0 / 50)
    # agreement with the reference's fp32 segmentation (not a bit-exact claim in a 16-bit mode): the floor is the
    # measured agreement on this clip minus a margin of one flipped boundary pair (profiles/r03_sample_wav_agreement.md)
    assert _boundary_recall(out["segments"], g["sample_segments"]) >= BOUNDARY_FLOOR["bf16"]


# measured (profiles/r05_sample_wav_agreement.md): bf16 78 / 81, fp16 81 / 81; floors = measured minus one flipped boundary PAIR
BOUNDARY_FLOOR = {"bf16": 0.93, "fp16": 0.975, "split": 1.0}


def _boundary_recall(got, ref):
    ref_b = set(np.asarray(ref).reshape(-1).tolist())
    got_b = set(np.asarray(got).reshape(-1).tolist())
    return len(ref_b & got_b) / max(len(ref_b), 1)


@pytest.mark.parametrize("precision", ["fp16"])
def test_sample_wav_golden_other_modes(sd, golden_dir, tmp_path, precision):
    """the same file through the other operand formats: hidden states within the mode's budget, segmentation consistent
    with the returned hidden states, boundary agreement with the REFERENCE's fp32 tables above the mode's floor"""
    from sylber_amd import Segmenter
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    import wave
    p = str(tmp_path / "sample.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(g["sample_pcm"].tobytes())
    Sm = Segmenter(model_ckpt=sd, precision=precision)
    out = Sm(p, in_second=False)
    _check_contract(out, False)
    assert rel_rms(out["hidden_states"], g["sample_hidden"]) < (2e-2 if precision == "bf16" else 3e-3)
    _segments_consistent(out, Sm)
    assert _boundary_recall(out["segments"], g["sample_segments"]) >= BOUNDARY_FLOOR[precision]


def test_list_input_and_padding(S, sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    wl = [syllable_wave(int(n), int(s)) for n, s in zip(g["batch_lengths"], g["batch_seeds"])]
    outs = S(wav=wl, in_second=False)
    assert isinstance(outs, list) and len(outs) == 3
    T = g["batch0_hidden"].shape[0]
    for i, o in enumerate(outs):
        _check_contract(o, False)
        assert o["hidden_states"].shape == (T, 768)          # full padded T for every row
        assert rel_rms(o["hidden_states"], g[f"batch{i}_hidden"]) < 2e-2
        _segments_consistent(o, S)
    single = S(wav=wl[0], in_second=True)
    assert isinstance(single, dict)
    _check_contract(single, True)


def test_matches_cpu_oracle_dict(S, sd):
    ref = SegmenterRef(sd)
    x = syllable_wave(40000, 77)
    a = S(wav=x, in_second=False)
    b = ref(x, in_second=False)
    assert a["hidden_states"].shape == b["hidden_states"].shape
    assert rel_rms(a["hidden_states"], b["hidden_states"]) < 2e-2
    _segments_consistent(a, S)


def test_silence_only_gives_empty_arrays(sd):
    from sylber_amd import Segmenter
    S2 = Segmenter(model_ckpt=sd, norm_threshold=1e9)
    out = S2(wav=syllable_wave(16000, 1))
    assert out["segments"].shape == (0,) and out["segment_features"].shape == (0,)
    assert out["hidden_states"].shape == (49, 768)


def test_edge_shapes_and_errors(S, sd):
    """shortest possible clip (one frame), odd lengths, batch of one, and the error behaviour of the ABI"""
    from sylber_amd import _lib
    ref = SegmenterRef(sd)
    for n in (400, 401, 719, 720, 1000, 16001):
        x = syllable_wave(max(n, 800), 300 + n)[:, :n].contiguous()
        a = S(wav=x, in_second=False)
        b = ref(x, in_second=False)
        assert a["hidden_states"].shape == b["hidden_states"].shape, n
        assert rel_rms(a["hidden_states"], b["hidden_states"]) < 3e-2, n
        _segments_consistent(a, S)
    # ragged list with a one-frame utterance next to a long one
    wl = [syllable_wave(800, 1)[:, :400].contiguous(), syllable_wave(24000, 2)]
    outs = S(wav=wl, in_second=True)
    assert outs[0]["hidden_states"].shape == outs[1]["hidden_states"].shape
    for o in outs:
        _check_contract(o, True)
    # too short for a single frame: the reference would fail inside the conv stack; we raise
    with pytest.raises(_lib.SylberHipError):
        S(wav=torch.zeros(1, 399))
    with pytest.raises(ValueError):
        S(wav=torch.zeros(16000))          # sylber.py:96 reads wav.shape[1]


def test_stereo_file_rows(S, tmp_path):
    """torch.cat(dim=0) at sylber.py:117 turns every channel of a multi-channel file into a batch row"""
    import wave
    pcm = (np.sin(np.arange(32000) * 0.05) * 8000).astype(np.int16).reshape(-1, 2)
    p = str(tmp_path / "st.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
    outs = S([p], in_second=False)
    assert isinstance(outs, list) and len(outs) == 2


def test_tensor_native_segment_api(S, sd):
    """`segment(input_values, attention_mask)` mirrors the reference's Sylber.segment (sylber.py:208-247)"""
    lens = [24000, 16000, 20000]
    wavs = [syllable_wave(n, 700 + i) for i, n in enumerate(lens)]
    batch = torch.zeros(3, max(lens))
    mask = torch.zeros(3, max(lens), dtype=torch.long)
    for i, w in enumerate(wavs):
        batch[i, : lens[i]] = w[0]
        mask[i, : lens[i]] = 1
    feats, segments, avg_fts = S.segment(input_values=batch, attention_mask=mask)
    ref = S(wav=wavs, in_second=False)
    assert feats.shape == (3, 74, 768) and avg_fts.shape[0] == 3 and avg_fts.shape[2] == 768
    for i, r in enumerate(ref):
        assert np.array_equal(feats[i].cpu().numpy(), r["hidden_states"])
        assert np.array_equal(segments[i], r["segments"])
        n = len(r["segments"])
        assert np.array_equal(avg_fts[i, :n].cpu().numpy(), r["segment_features"])
        assert float(avg_fts[i, n:].abs().sum()) == 0.0
    # precomputed features + explicit thresholds; a threshold nobody reaches gives the single zero row
    f2, seg2, avg2 = S.segment(features=feats, normthreshold=1e9, mergethreshold=0.5)
    assert all(s.shape == (0,) for s in seg2) and avg2.shape == (3, 1, 768) and float(avg2.abs().sum()) == 0.0


def test_two_and_a_half_minute_utterance(S):
    """One 150 s utterance (7499 frames): beyond the 78.8 s the segmenter keeps in LDS, far beyond any length the encoder
    kernels were tuned on; the API contract holds and the segments are the reference algorithm's on the returned states."""
    x = syllable_wave(150 * 16000, 31)
    out = S(wav=x, in_second=False)
    assert out["hidden_states"].shape == (7499, 768) and np.isfinite(out["hidden_states"]).all()
    assert len(out["segments"]) > 100
    _segments_consistent(out, S)


def test_pinned_output_pool_behaviour(sd):
    """the default hand-over of Segmenter.__call__ (leased page-locked blocks, sylber_amd/segmenter.py PinnedOutputPool):
    page-locking stops after the first calls, results stay intact after later calls, retained results beyond
    max_pinned_batches fall back to pageable copies, and output_memory="pageable" returns ordinary arrays"""
    import gc
    from sylber_amd import Segmenter
    wavs = [syllable_wave(16000 + 800 * i, 40 + i) for i in range(3)]
    S2 = Segmenter(model_ckpt=sd, max_pinned_batches=2)
    first = S2(wav=wavs, in_second=False)
    keep = [o["hidden_states"].copy() for o in first]
    for _ in range(5):                                   # dropped results: their block goes back to the pool
        S2(wav=wavs, in_second=False)
        gc.collect()
    assert S2.out_pool.allocations <= 2
    held = [first, S2(wav=wavs, in_second=False)]        # two batches retained = max_pinned_batches
    third = S2(wav=wavs, in_second=False)                # no block left: pageable per-utterance copies
    assert S2.out_pool.leased == 2 and S2.out_pool.allocations <= 2
    assert third[0]["hidden_states"].base is None or not _is_pool_view(third[0]["hidden_states"], S2)
    for outs in held + [third]:
        for o, k in zip(outs, keep):
            assert np.array_equal(o["hidden_states"], k)  # nothing was overwritten by the later calls
            _check_contract(o, False)
            _segments_consistent(o, S2)
    # one utterance's hidden_states keeps its whole batch block out (documented); dropping the rest does not free it
    one = held[1][1]["hidden_states"]
    del held, first, third
    gc.collect()
    assert S2.out_pool.leased == 1 and np.array_equal(one, keep[1])
    del one
    gc.collect()
    assert S2.out_pool.leased == 0
    S3 = Segmenter(model_ckpt=sd, output_memory="pageable")
    outs = S3(wav=wavs, in_second=False)
    assert S3.out_pool.allocations == 0
    for o, k in zip(outs, keep):
        assert np.array_equal(o["hidden_states"], k)


def _is_pool_view(arr, S):
    base = arr
    while getattr(base, "base", None) is not None:
        base = base.base
    return isinstance(base, np.ndarray) and base.dtype == np.uint8 and base.nbytes % S.out_pool.GRANULE == 0 and base.nbytes >= arr.nbytes and base is not arr


def test_ragged_serving_loop_is_history_independent(sd):
    """a serving loop with a new batch size and a new maximum length per call (the workspace layout changes every time) returns,
    call by call, the bits of a handle whose workspace was filled with NaN patterns just before that call: no result depends on
    what the handle did before (graph replay included)"""
    from sylber_amd import Segmenter, _lib
    rng = np.random.default_rng(11)
    A = Segmenter(model_ckpt=sd)
    Bs = Segmenter(model_ckpt=sd)
    G = Segmenter(model_ckpt=sd)
    G.speech_model.set_graph_mode(True)
    for call in range(24):
        nb = int(rng.integers(1, 9))
        wavs = [syllable_wave(int(rng.integers(4000, 70000)), 700 + 10 * call + i) for i in range(nb)]
        if call % 5 == 4:
            wavs = wavs * 2                                   # a repeated shape now and then: the graph entry is replayed
        got = A(wav=wavs, in_second=False)
        _lib.check(Bs.speech_model.lib.sylber_debug_poison_workspace(Bs.speech_model.handle, 0xFF), "poison")
        exp = Bs(wav=wavs, in_second=False)
        gg = G(wav=wavs, in_second=False)
        for g, e, h in zip(got, exp, gg):
            assert np.array_equal(g["hidden_states"], e["hidden_states"]) and np.array_equal(h["hidden_states"], e["hidden_states"]), call
            assert np.array_equal(g["segments"], e["segments"]) and np.array_equal(h["segments"], e["segments"]), call
            assert np.array_equal(g["segment_features"], e["segment_features"], equal_nan=True), call


def test_stream_of_batches_equals_calls(sd):
    """Segmenter.stream (padding + H2D of batch i + 1 and D2H + slicing of batch i - 1 under batch i's forward) yields, batch by batch,
    the bits of the synchronous __call__: ragged batches of changing size, a single-tensor item, both output modes, a batch with
    more segments than any before it (the tables' second fetch), and a consumer that keeps every result"""
    from sylber_amd import Segmenter
    rng = np.random.default_rng(21)
    batches = []
    for j in range(7):
        nb = int(rng.integers(1, 7))
        batches.append([syllable_wave(int(rng.integers(6000, 60000)), 900 + 10 * j + i) for i in range(nb)])
    batches.insert(3, syllable_wave(30000, 77))                               # a bare tensor: one dict, not a list
    batches.append([syllable_wave(200000, 78 + i) for i in range(3)])          # long clips: many segments
    ref_seg = Segmenter(model_ckpt=sd)
    ref = [ref_seg(wav=b, in_second=False) for b in batches]
    for mode in ("pinned", "pageable"):
        S = Segmenter(model_ckpt=sd, output_memory=mode, max_pinned_batches=3)
        S._kcap_seen = 16                                                     # force the overflow path early
        kept = []
        for got, exp in zip(S.stream(batches, in_second=False), ref):
            kept.append(got)
            got_l, exp_l = (got if isinstance(got, list) else [got]), (exp if isinstance(exp, list) else [exp])
            assert isinstance(got, list) == isinstance(exp, list) and len(got_l) == len(exp_l)
            for g, e in zip(got_l, exp_l):
                _check_contract(g, False)
                assert np.array_equal(g["hidden_states"], e["hidden_states"])
                assert np.array_equal(g["segments"], e["segments"])
                assert np.array_equal(g["segment_features"], e["segment_features"], equal_nan=True)
        assert len(kept) == len(batches)
        for got, exp in zip(kept, ref):                                       # nothing was overwritten by later batches
            got_l, exp_l = (got if isinstance(got, list) else [got]), (exp if isinstance(exp, list) else [exp])
            for g, e in zip(got_l, exp_l):
                assert np.array_equal(g["hidden_states"], e["hidden_states"]) and np.array_equal(g["segment_features"], e["segment_features"], equal_nan=True)
    assert list(Segmenter(model_ckpt=sd).stream([])) == []
    sec = list(Segmenter(model_ckpt=sd).stream(batches[:2], in_second=True))
    assert all(np.array_equal(a["segments"], b["segments"] / 50.0) for a, b in zip(sec[0], ref[0]))


def test_call_split_returns_the_unsplit_call(sd):
    """round 6 (an option, off by default: measured slower, tools/api_split_ab.py): `call_split = n` cuts a large host batch into n sub-batches pipelined
    INSIDE one synchronous __call__ (upload of part 2 under the forward of part 1, download of part 1 under the forward of part 2).  Every part is padded to the whole batch's longest clip (sylber.py:93-118 pads to
    the batch max and returns the padded frames) and an utterance's results do not depend on the batch it is computed in, so the call returns the
    bits of the unsplit call: ragged lengths, a stereo item (two rows), 2 / 3 / 4 parts, both output memories, the opt-in output subset; small
    batches, device tensors and a single tensor are not split"""
    from sylber_amd import Segmenter
    rng = np.random.default_rng(33)
    wavs = [syllable_wave(int(rng.integers(100000, 128000)), 500 + i) for i in range(23)]
    wavs.insert(5, torch.cat([syllable_wave(120000, 601), syllable_wave(120000, 602)], 0))      # [2, N]: two rows of the batch
    wavs[11] = syllable_wave(131072, 603)                                                         # the batch max
    ref_seg = Segmenter(model_ckpt=sd, call_split=0)
    assert ref_seg._split_plan(wavs) is None
    ref = ref_seg(wav=wavs, in_second=False)
    assert len(ref) == 25 and all(o["hidden_states"].shape == ref[0]["hidden_states"].shape for o in ref)
    for n, mode in ((2, "pinned"), (3, "pinned"), (4, "pageable")):
        S = Segmenter(model_ckpt=sd, call_split=n, output_memory=mode)
        plan = S._split_plan(wavs)
        assert plan is not None and len(plan) == min(n, 3) and sum(len(p) for p in plan) == len(wavs)      # (25 rows: at most 3 parts of 8+ rows)
        for _ in range(2):
            got = S(wav=wavs, in_second=False)
            assert len(got) == len(ref)
            for g, e in zip(got, ref):
                _check_contract(g, False)
                assert np.array_equal(g["hidden_states"], e["hidden_states"]) and np.array_equal(g["segments"], e["segments"])
                assert np.array_equal(g["segment_features"], e["segment_features"], equal_nan=True)
        sec = S(wav=wavs, in_second=True)
        assert all(np.array_equal(a["segments"], b["segments"] / 50.0) for a, b in zip(sec, ref))
    S = Segmenter(model_ckpt=sd, call_split=2, outputs=("segments", "segment_features"))
    got = S(wav=wavs, in_second=False)
    assert all(set(g) == {"segments", "segment_features"} and np.array_equal(g["segments"], e["segments"]) and
               np.array_equal(g["segment_features"], e["segment_features"], equal_nan=True) for g, e in zip(got, ref))
    # not split: few rows, little audio, device tensors, a bare tensor
    assert S._split_plan(wavs[:6]) is None and S._split_plan([w[:, :8000] for w in wavs]) is None
    assert S._split_plan([w.cuda() for w in wavs]) is None
    one = S(wav=wavs[0], in_second=False)
    assert isinstance(one, dict) and np.array_equal(one["segments"], Segmenter(model_ckpt=sd, call_split=0)(wav=wavs[0], in_second=False)["segments"])


def test_outputs_opt_in_skips_hidden_states(sd):
    """round 6: `outputs=` chooses which keys of the reference's dict (sylber.py:134-138) a call returns.  The default is the reference's
    contract (all three); without "hidden_states" the 49 MB D2H of a 32 x 10 s batch and its page-locked block are skipped, without
    "segment_features" the pooling launch too -- the keys that ARE returned hold the same bits, from __call__ and from stream, in both
    output-memory modes, also through the tables' second fetch"""
    from sylber_amd import Segmenter
    rng = np.random.default_rng(5)
    batches = [[syllable_wave(int(rng.integers(6000, 50000)), 300 + 10 * j + i) for i in range(int(rng.integers(1, 6)))] for j in range(4)]
    batches.append([syllable_wave(150000, 340 + i) for i in range(2)])
    full = Segmenter(model_ckpt=sd)
    assert full.outputs == ("segments", "segment_features", "hidden_states")
    ref = [full(wav=b, in_second=False) for b in batches]
    for outs in (("segments", "segment_features"), ("segments",), ("segments", "hidden_states")):
        for mode in ("pinned", "pageable"):
            S = Segmenter(model_ckpt=sd, outputs=outs, output_memory=mode)
            S._kcap_seen = 16
            for src in ("call", "stream"):
                got_all = [S(wav=b, in_second=False) for b in batches] if src == "call" else list(S.stream(batches, in_second=False))
                for got, exp in zip(got_all, ref):
                    for g, e in zip(got, exp):
                        assert set(g) == set(outs), (outs, mode, src)
                        for k in outs:
                            assert np.array_equal(g[k], e[k], equal_nan=True), (outs, mode, src, k)
    lean = Segmenter(model_ckpt=sd, outputs=("segments", "segment_features"))
    lean(wav=batches[0])
    with pytest.raises(ValueError):
        Segmenter(model_ckpt=sd, outputs=("hidden_states",))
    with pytest.raises(ValueError):
        Segmenter(model_ckpt=sd, outputs=("segments", "logits"))


def test_stream_abandoned_early_then_reused(sd):
    """a consumer that stops early (break / close() / an exception) leaves batches in flight whose leased page-locked
    blocks the D2H stream is still writing; the generator's exit waits for those copies before the leases go back, the budget and
    the pool are restored, and what the consumer kept plus every later call / stream on the same Segmenter are the right bits.
    Also: the generator keeps computing on the stream it started on when the caller's current stream changes between resumptions."""
    from sylber_amd import Segmenter
    batches = [[syllable_wave(48000 + 1000 * j, 500 + 10 * j + i) for i in range(6)] for j in range(6)]
    ref_seg = Segmenter(model_ckpt=sd)
    ref = [ref_seg(wav=b, in_second=False) for b in batches]
    S = Segmenter(model_ckpt=sd, max_pinned_batches=2)
    budget = S.out_pool.max_leased
    for rounds in range(3):
        gen = S.stream(batches, in_second=False)
        first = next(gen)
        gen.close()                                                            # two more batches were in flight
        assert S.out_pool.max_leased == budget and len(S.out_pool._free) + S.out_pool.leased <= budget
        again = S(wav=batches[3], in_second=False)                             # leases a block at once
        for g, e in zip(first, ref[0]):
            assert np.array_equal(g["hidden_states"], e["hidden_states"]) and np.array_equal(g["segments"], e["segments"])
        for g, e in zip(again, ref[3]):
            assert np.array_equal(g["hidden_states"], e["hidden_states"]) and np.array_equal(g["segments"], e["segments"])
            assert np.array_equal(g["segment_features"], e["segment_features"], equal_nan=True)
        del first, again
    other = torch.cuda.Stream()
    got = []
    gen = S.stream(batches, in_second=False)
    for j in range(len(batches)):
        if j % 2:
            with torch.cuda.stream(other):                                     # the caller's current stream changes between resumptions
                got.append(next(gen))
        else:
            got.append(next(gen))
    for out, exp in zip(got, ref):
        for g, e in zip(out, exp):
            assert np.array_equal(g["hidden_states"], e["hidden_states"]) and np.array_equal(g["segments"], e["segments"])


# measured on MI355X with tools/agreement_table.py (round 5, 64 / 256 synthetic 10 s clips, truth = this library's fp32 mode, which is
# bit-identical to the reference on every golden): tables identical 12 / 64 (38 / 256) bf16, 47 / 64 (183 / 256) fp16, 0 fp8, all split16;
# boundary recall 0.9943 / 0.9984 / 0.9247 / 1.0; hidden rel-RMS 5.4e-3 / 7.3e-4 / 3.7e-2 / 3.1e-6.  Floors = measured minus a margin
# of a few flipped decisions, so that a regression of a fast mode's DECISIONS is caught, not only of its hidden-state error
AGREEMENT_FLOORS = {
    "bf16": dict(recall=0.990, precision=0.989, tables=6, rel=7e-3),
    "fp16": dict(recall=0.9965, precision=0.9975, tables=38, rel=1e-3),
    "fp8": dict(recall=0.90, precision=0.95, tables=0, rel=4.5e-2),
    "split16": dict(recall=1.0, precision=1.0, tables=64, rel=1e-5),
}


def test_agreement_table_floors(sd):
    """the segment-agreement table of every fast mode (boundaries found / tables identical against the fp32 parity
    mode on 64 synthetic 10 s clips = 15 186 boundaries) with floors at the measured values; split16 must reproduce EVERY table
    (north_star's "boundaries bit-identical" is met by precision="split16" and "fp32": INTEGRATION.md)"""
    from sylber_amd import HubertEncoderHIP
    from sylber_amd.agreement import segment_agreement
    truth = HubertEncoderHIP(sd, precision="fp32")
    for prec, fl in AGREEMENT_FLOORS.items():
        e = HubertEncoderHIP(sd, precision=prec)
        r = segment_agreement(sd, e, 64, truth=truth)
        assert r["clips"] == 64 and r["boundaries_fp32"] > 10000
        assert r["boundary_recall"] >= fl["recall"], (prec, r)
        assert r["boundary_precision"] >= fl["precision"], (prec, r)
        assert r["tables_identical"] >= fl["tables"], (prec, r)
        assert r["hidden_rel_rms_vs_fp32"] <= fl["rel"], (prec, r)
        del e
