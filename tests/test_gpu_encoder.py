"""GPU tier: the HIP encoder (sylber_forward through the C-ABI) against the fp32 oracle and the
golden per-stage activations from the reference's HubertModel (sylber/model/sylber.py:122).
bf16 MFMA compute: tolerance is relative RMS, budget 1.3e-2 at the output with
per-stage bounds written below."""
import os

import numpy as np
import pytest
import torch

from oracle import hubert_ref
from sylber_amd.synth import noise_batch, syllable_wave
from sylber_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu


def rel_rms(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-30))


@pytest.fixture(scope="module")
def sd():
    return synthetic_state_dict(0)


@pytest.fixture(scope="module")
def enc(sd):
    from sylber_amd import HubertEncoderHIP
    return HubertEncoderHIP(sd)


STAGE_TOL = {"conv": 1.5e-2, "enc_in": 1.0e-2, "layer": 2.0e-2, "hidden": 2.0e-2}


def test_stages_vs_reference_goldens(enc, sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "encoder_stages.npz"))
    wav = torch.from_numpy(g["wav"]).cuda()
    lengths = [int(x) for x in g["lengths"]]
    conv = enc.forward(wav, lengths, stop_stage=1).cpu().numpy()            # [B,T,512]
    assert rel_rms(conv, g["conv6"].transpose(0, 2, 1)) < STAGE_TOL["conv"]
    enc_in = enc.forward(wav, lengths, stop_stage=2).cpu().numpy()
    assert rel_rms(enc_in, g["enc_in"]) < STAGE_TOL["enc_in"]
    for l, key in [(0, "layer0"), (4, "layer4")]:
        h = enc.forward(wav, lengths, stop_stage=3 + l).cpu().numpy()
        assert rel_rms(h, g[key]) < STAGE_TOL["layer"], key
    h = enc.forward(wav, lengths).cpu().numpy()
    assert rel_rms(h, g["layer8"]) < STAGE_TOL["hidden"]
    assert np.isfinite(h).all()


def test_ragged_batch_vs_oracle(enc, sd):
    """Padding semantics: GroupNorm statistics include the zero padding, padded
    frames are zeroed before the pos-conv, masked as attention keys, and still returned."""
    lens = [48000, 30000, 47999, 16400]
    wavs = [syllable_wave(n, 40 + i) for i, n in enumerate(lens)]
    batch = torch.zeros(len(lens), max(lens))
    for i, w in enumerate(wavs):
        batch[i, : lens[i]] = w[0]
    ref = hubert_ref.forward(sd, batch, lens)["hidden"].numpy()
    out = enc.forward(batch.cuda(), lens).cpu().numpy()
    assert out.shape == ref.shape
    for i in range(len(lens)):
        assert rel_rms(out[i], ref[i]) < STAGE_TOL["hidden"], i
    # a padded utterance differs from the same utterance run alone (the reference behaves the same way)
    alone = enc.forward(wavs[1].cuda().contiguous(), [lens[1]]).cpu().numpy()
    T1 = alone.shape[1]
    assert rel_rms(out[1, :T1], alone[0]) > 1e-3


def test_no_mask_equals_all_ones_mask(enc):
    x = noise_batch(2, 16000, seed=3).cuda()
    a = enc.forward(x, None).cpu().numpy()
    b = enc.forward(x, [16000, 16000]).cpu().numpy()
    assert np.array_equal(a, b)


def test_batch_invariance_and_determinism(enc):
    """Utterances are independent units: a row's result does not depend on its batch neighbours when
    lengths are equal, and repeated launches are bitwise reproducible."""
    x = noise_batch(4, 32000, seed=5).cuda()
    full = enc.forward(x).cpu().numpy()
    again = enc.forward(x).cpu().numpy()
    assert np.array_equal(full, again)
    one = enc.forward(x[2:3].contiguous()).cpu().numpy()
    assert np.array_equal(full[2], one[0])


def test_full_size_config_vs_oracle_sample(enc, sd):
    """BASELINE configs[1] shape (32 x 10 s): finite, deterministic rows, and two rows checked against
    the oracle (the full batch would take minutes on the CPU)."""
    x = noise_batch(32, 160000, seed=0)
    xd = x.cuda()
    out = enc.forward(xd).cpu().numpy()
    assert out.shape == (32, 499, 768) and np.isfinite(out).all()
    for _ in range(3):                                   # full-size launches must be bitwise reproducible
        assert np.array_equal(enc.forward(xd).cpu().numpy(), out)
    # utterances are independent: every row equals the same clip run in a batch of 4 (same Lmax)
    part = enc.forward(xd[8:12].contiguous()).cpu().numpy()
    assert np.array_equal(part, out[8:12])
    ref = hubert_ref.forward(sd, x[[0, 31]], None)["hidden"].numpy()
    assert rel_rms(out[0], ref[0]) < STAGE_TOL["hidden"]
    assert rel_rms(out[31], ref[1]) < STAGE_TOL["hidden"]


def test_fused_outproj_layernorm_bitwise(sd):
    """SYLBER_OPT_FUSE_OUTPROJ_LN: the attention out-projection + LayerNorm as ONE launch on full-row tiles
    (csrc/gemm_rowln.hip) returns bit for bit what the GEMM launch + the LayerNorm launch return (same expressions, same
    summation order), for a full batch, ragged lengths (tile tails) and a small batch (forced on)"""
    from sylber_amd import HubertEncoderHIP
    a, b = HubertEncoderHIP(sd), HubertEncoderHIP(sd)
    a.set_option(4, 1); b.set_option(4, -1)
    x = noise_batch(32, 160000, seed=21).cuda()
    assert torch.equal(a.forward(x), b.forward(x))
    assert torch.equal(a.forward(x), a.forward(x))                       # and reproducibly so
    y = noise_batch(3, 52000, seed=22).cuda()
    lens = [52000, 31000, 47011]
    assert torch.equal(a.forward(y, lens), b.forward(y, lens))
    h = HubertEncoderHIP(sd, precision="fp16"); g = HubertEncoderHIP(sd, precision="fp16")
    h.set_option(4, 1); g.set_option(4, -1)
    assert torch.equal(h.forward(y, lens), g.forward(y, lens))


def test_residual_prefetch_option_bitwise(sd):
    """SYLBER_OPT_RESLN_PREFETCH: the K loops of out-proj / FFN2 requesting 1, 2 or all 3 fragment columns of the fp32 residual rows
    themselves (csrc/gemm_asm.hip PRE / PC) return bit for bit what the epilogue-loaded form returns, for the full batch"""
    from sylber_amd import HubertEncoderHIP
    x = noise_batch(32, 160000, seed=23).cuda()
    ref = None
    for v in (-1, 1, 2, 3, 0):
        e = HubertEncoderHIP(sd)
        e.set_option(6, v)
        out = e.forward(x)
        if ref is None:
            ref = out
        assert torch.equal(out, ref), v
        assert torch.equal(e.forward(x), ref), v


def test_tile_selection_options_bitwise(sd):
    """round 6: which tiles a handle's GEMM launches get depends on whether it owns the chip (SYLBER_OPT_GEMM_MODEL 0, default) or shares it with
    another in-flight batch (5 = set_batches_in_flight(2)), on the 192-row tiles (SYLBER_OPT_GEMM_H192) and on a forced row split
    (SYLBER_OPT_GEMM_TAIL) -- and never changes a bit: every output element is one fp32 chain over K in the same order whatever tile computes
    it.  Batch shapes that leave partial rounds of 256 tiles (where the selections actually differ), fp16 included; and an utterance's hidden
    states are the same alone and inside such a batch under either selection."""
    from sylber_amd import HubertEncoderHIP
    for (b, n, prec) in [(24, 160000, "bf16"), (5, 481000, "bf16"), (12, 240000, "fp16")]:
        x = noise_batch(b, n, seed=600 + b).cuda()
        ref = None
        # (round 6, second half: model 6 = the tile choice without the lone-round rule and the 64-row tiles; forced ids 1 / 2 = the 64x64 / 64x128 tiles and 5 = 128x128 on
        #  eight waves on every launch -- on the 16-bit-output launches a forced id runs on the 16x16x32 family member of its shape class)
        for opts in ((), ((12, 5),), ((11, -1),), ((8, 4),), ((8, 51),), ((12, 5), (8, 3)), ((12, 6),), ((1, 1),), ((1, 2),), ((1, 5),)):
            e = HubertEncoderHIP(sd, precision=prec)
            for k, v in opts:
                e.set_option(k, v)
            out = e.forward(x)
            if ref is None:
                ref = out
                one = e.forward(x[1:2].contiguous())
                assert torch.equal(one[0], ref[1]), (b, n, prec)
            assert torch.equal(out, ref), (b, n, prec, opts)
            del e
    e = HubertEncoderHIP(sd)
    e.set_batches_in_flight(2)
    x = noise_batch(24, 160000, seed=624).cuda()
    a = e.forward(x)
    e.set_batches_in_flight(1)
    assert torch.equal(e.forward(x), a)


def test_long_form_config(enc, sd):
    """BASELINE configs[3] at its stated batch: 8 x 60 s clips (T = 2999: O(T^2) attention, 192k-step GroupNorm).
    At this size the CPU oracle would take minutes, so the full batch is checked through size-independent properties
    -- finite, bitwise reproducible, utterances independent of their batch mates, segmentation bit-exact GIVEN the
    GPU's hidden states -- and one 25 s clip (T = 1249) against the oracle."""
    from oracle import segment_oracle
    x = noise_batch(8, 960000, seed=9).cuda()
    out = enc.forward(x)
    assert out.shape == (8, 2999, 768) and bool(torch.isfinite(out).all())
    assert torch.equal(enc.forward(x), out)
    part = enc.forward(x[2:4].contiguous())                       # same Lmax, other batch mates
    assert torch.equal(part, out[2:4])
    seg, nseg, feats = enc.segment(out, 2.6, 0.8)
    torch.cuda.synchronize()
    out_h, seg_h, nseg_h, feats_h = out.cpu().numpy(), seg.cpu().numpy(), nseg.cpu().numpy(), feats.cpu().numpy()
    for i in range(8):                                            # every row of the batch
        exp = segment_oracle.get_segment(out_h[i], 2.6, 0.8).reshape(-1, 2)
        assert nseg_h[i] == len(exp) and np.array_equal(seg_h[i, :nseg_h[i]], exp)
        if len(exp):
            assert np.array_equal(feats_h[i, :nseg_h[i]], segment_oracle.mean_pool(out_h[i], exp), equal_nan=True)
    y = syllable_wave(400000, 91)                      # 25 s, T = 1249
    ref = hubert_ref.forward(sd, y, None)["hidden"].numpy()
    got = enc.forward(y.cuda().contiguous()).cpu().numpy()
    assert got.shape == ref.shape
    assert rel_rms(got, ref) < STAGE_TOL["hidden"]


def test_graph_mode_replays_bitwise(sd):
    """sylber_set_graph_mode: eager call, capturing call and replays give bit-identical hidden states, also after the
    input buffer's CONTENTS change, for two shapes alternating, and with ragged lengths"""
    from sylber_amd import HubertEncoderHIP
    e = HubertEncoderHIP(sd)
    ref = HubertEncoderHIP(sd)
    e.set_graph_mode(True)
    xa, xb = noise_batch(2, 24000, seed=41).cuda(), noise_batch(1, 16000, seed=42).cuda()
    oa, ob = torch.empty(2, 74, 768, device="cuda"), torch.empty(1, 49, 768, device="cuda")
    for it in range(5):
        if it == 3:
            xa.copy_(noise_batch(2, 24000, seed=43).cuda())          # same buffer, new audio
        ha = e.forward(xa, [24000, 18000], out=oa).clone()
        hb = e.forward(xb, None, out=ob).clone()
        assert torch.equal(ha, ref.forward(xa, [24000, 18000]))
        assert torch.equal(hb, ref.forward(xb, None))
    # lengths are data, not part of the captured graph
    assert torch.equal(e.forward(xa, [20000, 24000], out=oa), ref.forward(xa, [20000, 24000]))
    e.set_graph_mode(False)
    assert torch.equal(e.forward(xa, None, out=oa), ref.forward(xa, None))


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_ragged_batches_vs_oracle(enc, sd, seed):
    """random batch size, random lengths down to the one-frame minimum (400 samples), random batch maximum: every row
    within the stage tolerance of the CPU oracle, padded frames included, and bitwise reproducible"""
    rng = np.random.default_rng(100 + seed)
    B = int(rng.integers(1, 6))
    lmax = int(rng.integers(2000, 30000))
    lens = [int(x) for x in rng.integers(400, lmax + 1, B)]
    lens[int(rng.integers(0, B))] = lmax
    x = noise_batch(B, lmax, seed=200 + seed)
    for i, n in enumerate(lens):
        x[i, n:] = 0.0
    out = enc.forward(x.cuda(), lens)
    assert torch.equal(out, enc.forward(x.cuda(), lens))
    ref = hubert_ref.forward(sd, x, lens)["hidden"].numpy()
    got = out.cpu().numpy()
    assert got.shape == ref.shape and np.isfinite(got).all()
    for i in range(B):
        assert rel_rms(got[i], ref[i]) < STAGE_TOL["hidden"], (i, lens)


def test_more_than_512_utterances_per_batch(enc):
    """B > 512 (the valid-frame table reaches the device 512 utterances per launch; every grid scales with B): rows of a
    600-clip ragged batch equal the same clips run in a small batch with the same padded length, bitwise."""
    B, L = 600, 8000
    rng = np.random.default_rng(11)
    x = noise_batch(B, L, seed=12)
    lengths = [int(v) for v in rng.integers(400, L + 1, size=B)]
    lengths[0] = L
    for i, n in enumerate(lengths):
        x[i, n:] = 0
    x = x.cuda()
    full = enc.forward(x, lengths).cpu().numpy()
    assert np.isfinite(full).all()
    pick = [0, 1, 511, 512, 513, 599]
    small = enc.forward(x[pick].contiguous(), [lengths[i] for i in pick]).cpu().numpy()
    for j, i in enumerate(pick):
        assert np.array_equal(full[i], small[j]), i
    seg, nseg, feats = enc.segment(torch.from_numpy(full).cuda(), 2.6, 0.8)
    seg2, nseg2, _ = enc.segment(torch.from_numpy(small).cuda(), 2.6, 0.8)
    nseg, nseg2 = nseg.cpu().numpy(), nseg2.cpu().numpy()
    for j, i in enumerate(pick):
        assert nseg[i] == nseg2[j]
        assert torch.equal(seg[i, : nseg[i]], seg2[j, : nseg2[j]])


def test_nan_utterance_does_not_leak_into_neighbours(enc):
    """Utterances are independent units: a NaN waveform poisons its own row only, and the segmenter still terminates on
    it with the reference's answer for NaN states (every comparison false -> no speech frame -> no segment)."""
    from oracle import segment_oracle
    x = noise_batch(3, 16000, seed=21)
    clean = enc.forward(x.cuda()).cpu().numpy()
    x[1, 100:200] = float("nan")
    h = enc.forward(x.cuda()).cpu().numpy()
    assert np.array_equal(h[0], clean[0]) and np.array_equal(h[2], clean[2])
    assert np.isnan(h[1]).any()
    seg, nseg, _ = enc.segment(torch.from_numpy(h).cuda(), 2.6, 0.8)
    nseg = nseg.cpu().numpy()
    for i in range(3):
        ref = segment_oracle.get_segment(np.ascontiguousarray(h[i]), 2.6, 0.8).reshape(-1, 2)
        assert nseg[i] == len(ref)
        assert np.array_equal(seg[i, : nseg[i]].cpu().numpy(), ref)


@pytest.mark.parametrize("precision", ["bf16", "fp16", "fp8", "split16", "fp32"])
def test_forward_reads_nothing_it_has_not_written(precision):
    """every byte of the activation workspace is overwritten with 0xFF (NaN patterns in every operand format), then the same
    forward runs again: the result must be the same bits.  Guards the aliasing of the workspace regions (q / k / V^T / context share
    memory with the FFN intermediate) and the regions that are read without being written (halo rows, key tails, slack rows, the
    padded query rows [T, Tp) of the attention context): a stale NaN there comes back as 0 x NaN in the next layer's P.V --
    found in the fp8 mode after a batch-shape change on one handle, fixed in csrc/attention.hip attn_finalize."""
    import ctypes
    from sylber_amd import HubertEncoderHIP, _lib
    from sylber_amd.synth import syllable_wave
    from sylber_amd.weights import synthetic_state_dict
    e = HubertEncoderHIP(synthetic_state_dict(0), precision=precision)
    shapes = [(8, 48000, [48000, 30000, 48000, 9000, 48000, 48000, 20000, 41000]),        # 149 frames: 11 padded rows per utterance
              (3, 40000, None), (2, 16400, [16400, 12000])]
    for B, n, lens in shapes:
        wav = torch.cat([syllable_wave(n, 90 + i) for i in range(B)], 0).cuda()
        ref = e.forward(wav, lens).clone()
        assert torch.isfinite(ref).all()
        for byte in (0xFF, 0x7F):
            _lib.check(e.lib.sylber_debug_poison_workspace(e.handle, byte), "poison")
            got = e.forward(wav, lens)
            assert torch.equal(got, ref), (precision, B, n, hex(byte))


@pytest.mark.parametrize("precision", ["bf16", "fp8"])
def test_a_non_finite_utterance_stays_in_its_rows(precision):
    """utterances are independent units on this path: a clip of NaN / inf samples must not change one bit of its
    batch neighbours' hidden states -- no kernel may mix rows of different utterances, not even through a multiplication by zero"""
    from sylber_amd import HubertEncoderHIP
    from sylber_amd.synth import syllable_wave
    from sylber_amd.weights import synthetic_state_dict
    e = HubertEncoderHIP(synthetic_state_dict(0), precision=precision)
    lens = [48000, 30000, 48000, 9000, 48000, 48000, 20000, 41000]
    wav = torch.cat([syllable_wave(48000, 120 + i) for i in range(8)], 0).cuda()
    ref = e.forward(wav, lens).clone()
    for bad, val in ((2, float("nan")), (5, float("inf")), (7, -1e30)):
        w2 = wav.clone()
        w2[bad, 100:20000] = val
        got = e.forward(w2, lens)
        keep = [i for i in range(8) if i != bad]
        assert torch.equal(got[keep], ref[keep]), (precision, bad, val)
    assert torch.equal(e.forward(wav, lens), ref)


def test_batch_beyond_4gb_matches_small_batches():
    """addressing beyond 32-bit byte offsets: a 48 x 30 s batch (conv0 output 4.7 GB, workspace ~8 GB) reproduces row for row and bit
    for bit what the same clips give in batches of 8 (tools/big_batch_check.py runs 128 x 30 s / 16 x 150 s: 21 / 13 GB)"""
    from sylber_amd import HubertEncoderHIP
    from sylber_amd.synth import noise_batch
    from sylber_amd.weights import synthetic_state_dict
    e = HubertEncoderHIP(synthetic_state_dict(0))
    B, N = 48, 480000
    x = noise_batch(B, N, seed=5).cuda()
    lens = [N - 1000 * (i % 7) for i in range(B)]
    big = e.forward(x, lens)
    assert torch.isfinite(big).all() and e.workspace_bytes() > (1 << 32)
    for i in range(0, B, 8):
        assert torch.equal(e.forward(x[i:i + 8].contiguous(), lens[i:i + 8]), big[i:i + 8]), i
