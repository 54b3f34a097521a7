"""GPU tier, BASELINE configs[4] (fp8 MFMA for the FFN GEMMs, tolerance vs the bf16 path): the MXFP8 quantiser
bit-exact against the oracle, the block-scaled MFMA GEMM against the oracle's float64 contraction of the same
quantised operands (fp32-accumulation tolerance), and the SYLBER_FP8 forward against the bf16 forward."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import mxfp8_ref as Q

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _quant_gpu(x):
    from sylber_amd import _lib
    lib = _lib.load()
    xd = torch.from_numpy(x).cuda()
    R, K = x.shape
    d = torch.empty(R, K, dtype=torch.uint8, device="cuda")
    s = torch.empty(K // 64, R, 2, dtype=torch.uint8, device="cuda")        # K-pair-major scale layout of the library
    _lib.check(lib.sylber_op_mx_quantize(_p(xd), R, K, _p(d), _p(s), None), "mx_quantize")
    torch.cuda.synchronize()
    return d.cpu().numpy(), s.cpu().numpy().transpose(1, 0, 2).reshape(R, K // 32)


def test_quantiser_bit_exact():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((257, 384)) * np.exp(rng.uniform(-8, 8, (257, 1)))).astype(np.float32)
    x[5, 64:96] = 0.0                                       # all-zero block
    x[6, :32] = np.float32(448.0) * np.float32(2.0) ** rng.integers(-20, 20, 32)   # exactly on the scale edge
    x[7, :32] = np.linspace(-1, 1, 32, dtype=np.float32) * np.float32(2.0 ** -9) * 3   # subnormal grid ties
    x[8] = np.float32(1e-40)                                # float32 subnormals
    d, s = _quant_gpu(x)
    de, se = Q.quantize(x)
    assert np.array_equal(s, se)
    assert np.array_equal(d, de)


@pytest.mark.parametrize("M,N,K,act,cfg", [(300, 256, 128, 0, 0), (1000, 768, 3072, 0, 0), (515, 3072, 768, 1, 0),
                                            (129, 100, 256, 0, 1), (2000, 192, 384, 0, 1), (300, 256, 128, 0, 2),
                                            (1000, 768, 3072, 0, 3), (515, 3072, 768, 1, 2), (129, 100, 256, 0, 3),
                                            (2100, 500, 640, 0, 2), (777, 768, 768, 0, -1)])
def test_mxfp8_linear_matches_oracle(M, N, K, act, cfg):
    from sylber_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(M + N)
    a = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-2, 2, (M, 1)))).astype(np.float32)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    a[:, ::7] *= 8.0                                        # per-block dynamic range
    bias = rng.standard_normal(N).astype(np.float32)
    ad, wd, bd = torch.from_numpy(a).cuda(), torch.from_numpy(w).cuda(), torch.from_numpy(bias).cuda()
    out = torch.empty(M, N, device="cuda")
    _lib.check(lib.sylber_op_linear(_p(ad), _p(wd), _p(bd), _p(out), M, N, K, act, 2, cfg, None), "op_linear fp8")   # cfg -1 = automatic tile
    exp = Q.linear(a, w, bias)
    # the block-scaled MFMA does not sum its 64 products in exact fp32 (the products look aligned to the largest one
    # and truncated): measured on MI355X (tools/fp8_debug.py) 2e-5 x sum_k |a_k w_k| per output on Gaussian data
    # (mean 1.4e-6), ~1e-4 when the exponents inside a block are spread by a further 2^3, and EXACT on integer
    # data (test below) -> tolerance relative to that magnitude, two orders below the fp8 rounding noise itself
    mag = np.abs(Q.dequantize(*Q.quantize(a))) @ np.abs(Q.dequantize(*Q.quantize(w))).T + np.abs(bias)[None, :]
    if act:
        exp = 0.5 * exp * (1.0 + np.vectorize(__import__("math").erf)(exp / np.sqrt(2.0)))
    got = out.cpu().numpy().astype(np.float64)
    assert np.all(np.abs(got - exp) <= 3e-4 * mag + (1.5e-4 if act else 0.0))      # (+ GELU polynomial 6.4e-5)
    # and the quantisation itself costs what fp8 costs: a few percent against the unquantised product
    full = a.astype(np.float64) @ w.astype(np.float64).T + bias
    if not act:
        rel = np.sqrt(((got - full) ** 2).mean() / (full ** 2).mean())
        assert rel < 0.06


@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
def test_mxfp8_linear_exact_on_integer_data(cfg):
    """small integers and power-of-two block scales are exact in e4m3 x E8M0 and in the MFMA: any layout, OPSEL or
    scale-routing mistake shows up as a wrong integer"""
    from sylber_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(9)
    M, N, K = 333, 200, 384
    a = rng.integers(-3, 4, (M, K)).astype(np.float32) * (2.0 ** rng.integers(-3, 4, (M, K // 32))).repeat(32, 1).astype(np.float32)
    w = rng.integers(-2, 3, (N, K)).astype(np.float32) * (2.0 ** rng.integers(-2, 3, (N, K // 32))).repeat(32, 1).astype(np.float32)
    ad, wd = torch.from_numpy(a).cuda(), torch.from_numpy(w).cuda()
    out = torch.empty(M, N, device="cuda")
    _lib.check(lib.sylber_op_linear(_p(ad), _p(wd), None, _p(out), M, N, K, 0, 2, cfg, None), "op_linear fp8")
    assert np.array_equal(out.cpu().numpy().astype(np.float64), a.astype(np.float64) @ w.astype(np.float64).T)


@pytest.mark.parametrize("cfg,M,N,K", [(85, 512, 768, 768), (91, 512, 768, 768), (85, 256, 256, 512), (91, 768, 384, 3072),
                                        (85, 1024, 1024, 1024), (91, 16384, 768, 768), (-1, 16384, 3072, 768)])
def test_mxfp8_asm_loop_tiles(cfg, M, N, K):
    """the hand-scheduled X3 loop on MXFP8 operands (csrc/gemm_asm_f8.hip, tiles 85 = 256x256 / 91 = 256x192; -1 = the automatic
    choice, which takes it wherever it applies): EXACT on integer data with power-of-two block scales (any fragment-half, OPSEL,
    scale-routing or ring mistake is a wrong integer), and on Gaussian data bit for bit the 8-wave kernel of round 1 (same
    contraction order over K), run to run"""
    from sylber_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(K + N)
    a = rng.integers(-3, 4, (M, K)).astype(np.float32) * (2.0 ** rng.integers(-3, 4, (M, K // 32))).repeat(32, 1).astype(np.float32)
    w = rng.integers(-2, 3, (N, K)).astype(np.float32) * (2.0 ** rng.integers(-2, 3, (N, K // 32))).repeat(32, 1).astype(np.float32)
    ad, wd = torch.from_numpy(a).cuda(), torch.from_numpy(w).cuda()
    out = torch.empty(M, N, device="cuda")
    _lib.check(lib.sylber_op_linear(_p(ad), _p(wd), None, _p(out), M, N, K, 0, 2, cfg, None), "op_linear fp8")
    assert torch.equal(out.cpu().double(), (ad.double() @ wd.double().T).cpu())
    g = torch.Generator().manual_seed(M + K)
    a2 = torch.randn(M, K, generator=g).cuda(); w2 = (torch.randn(N, K, generator=g) * 0.05).cuda(); b2 = torch.randn(N, generator=g).cuda()
    ref = torch.empty(M, N, device="cuda")
    _lib.check(lib.sylber_op_linear(_p(a2), _p(w2), _p(b2), _p(ref), M, N, K, 1, 2, 2, None), "op_linear fp8")
    for _ in range(2):
        got = torch.full((M, N), float("nan"), device="cuda")
        _lib.check(lib.sylber_op_linear(_p(a2), _p(w2), _p(b2), _p(got), M, N, K, 1, 2, cfg, None), "op_linear fp8")
        assert torch.equal(got, ref), (cfg, M, N, K)


def test_fp8_forward_close_to_bf16():
    from sylber_amd import HubertEncoderHIP
    from sylber_amd.synth import syllable_wave
    from sylber_amd.weights import synthetic_state_dict
    sd = synthetic_state_dict(0)
    wav = torch.cat([syllable_wave(32000, 5), syllable_wave(32000, 6)], 0).cuda()
    lengths = [32000, 25000]
    e16 = HubertEncoderHIP(sd)
    e8 = HubertEncoderHIP(sd, precision="fp8")
    h16 = e16.forward(wav, lengths).cpu().numpy().astype(np.float64)
    h8 = e8.forward(wav, lengths).cpu().numpy().astype(np.float64)
    assert np.isfinite(h8).all()
    rel = np.sqrt(((h8 - h16) ** 2).mean() / (h16 ** 2).mean())
    # stated tolerance of the fp8 mode against the bf16 mode: 3-bit mantissas in both FFN operands of 9 layers
    assert rel < 0.08, rel
    h8b = e8.forward(wav, lengths).cpu().numpy().astype(np.float64)
    assert np.array_equal(h8, h8b)                         # deterministic
    # segmentation runs on the fp8 hidden states like on any others (bit-exact GIVEN those states)
    from oracle import segment_oracle
    seg, nseg, _ = e8.segment(torch.from_numpy(h8.astype(np.float32)).cuda(), 2.6, 0.8)
    exp = segment_oracle.get_segment(h8[0].astype(np.float32), 2.6, 0.8)
    n = int(nseg[0])
    assert n == len(exp) and (n == 0 or np.array_equal(seg[0, :n].cpu().numpy(), exp))


# measured on MI355X (profiles/r02_parity_report.md): 0 at the stages in front of the encoder (the conv stack and the
# projection stay bf16), 1.6e-2 / 3.1e-2 / 3.9e-2 relative RMS after encoder layers 0 / 4 / 8; stated tolerance of the mode
FP8_STAGE_TOL = {"conv": 1.5e-2, "enc_in": 1.0e-2, "layer0": 3.0e-2, "layer4": 5.0e-2, "layer8": 6.0e-2}


def test_fp8_forward_vs_reference_goldens(golden_dir):
    """configs[4] against the REFERENCE's per-stage goldens (not against this library's bf16 path): every stage within
    the stated fp8 tolerance, and the end-to-end segment boundaries mostly the reference's (recorded, with a floor)."""
    import os
    from oracle import segment_oracle
    from sylber_amd import HubertEncoderHIP
    from sylber_amd.weights import synthetic_state_dict
    sd = synthetic_state_dict(0)
    e8 = HubertEncoderHIP(sd, precision="fp8")
    g = np.load(os.path.join(golden_dir, "encoder_stages.npz"))
    wav = torch.from_numpy(g["wav"]).cuda()
    lengths = [int(x) for x in g["lengths"]]

    def rel(a, b):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        return float(np.sqrt(((a - b) ** 2).mean() / (b ** 2).mean()))

    assert rel(e8.forward(wav, lengths, stop_stage=1).cpu().numpy(), g["conv6"].transpose(0, 2, 1)) < FP8_STAGE_TOL["conv"]
    assert rel(e8.forward(wav, lengths, stop_stage=2).cpu().numpy(), g["enc_in"]) < FP8_STAGE_TOL["enc_in"]
    for l, key in [(0, "layer0"), (4, "layer4")]:
        assert rel(e8.forward(wav, lengths, stop_stage=3 + l).cpu().numpy(), g[key]) < FP8_STAGE_TOL[key], key
    h = e8.forward(wav, lengths).cpu().numpy()
    assert np.isfinite(h).all() and rel(h, g["layer8"]) < FP8_STAGE_TOL["layer8"]
    # the golden batch is 2 x 30 frames = 64 rows: the q / k / v launch pads it to one 256-row tile and the attention core runs on
    # MXFP8 operands (SYLBER_OPT_FP8_ATTENTION default) whatever the batch shape.  Four copies of it (utterances are independent)
    # are exactly one tile -- same goldens, same tolerances, and the same bits per utterance
    wav4, len4 = wav.repeat(4, 1), lengths * 4
    assert np.array_equal(e8.forward(wav4, len4).cpu().numpy()[:2], h)
    for l, key in [(0, "layer0"), (4, "layer4"), (8, "layer8")]:
        got = e8.forward(wav4, len4, stop_stage=(3 + l) if l < 8 else 0).cpu().numpy()
        assert np.isfinite(got).all()
        for r in range(4):
            assert rel(got[2 * r:2 * r + 2], g[key]) < FP8_STAGE_TOL[key], (key, r)
    e8.set_option(7, -1)
    assert not np.array_equal(e8.forward(wav4, len4).cpu().numpy(), got)       # (the option really switches the core)
    e8.set_option(7, 0)
    # boundaries: reference get_segment on the reference's hidden states vs the fp8 path end to end
    tot = hit = 0
    seg, nseg, _ = e8.segment(torch.from_numpy(h).cuda(), 2.6, 0.8)
    seg, nseg = seg.cpu().numpy(), nseg.cpu().numpy()
    for i in range(h.shape[0]):
        ref_s = segment_oracle.get_segment(np.ascontiguousarray(g["layer8"][i]), 2.6, 0.8).reshape(-1, 2)
        got_s = seg[i, : nseg[i]]
        rb, gb = set(ref_s.reshape(-1).tolist()), set(got_s.reshape(-1).tolist())
        tot += len(rb); hit += len(rb & gb)
    # measured 93-94 % on 16 clips (profiles/r02_parity_report.md); the goldens hold only 25 boundaries (one flip = 4 %) and
    # any change of the last bit upstream reshuffles which ones flip (observed 20-24 of 25), hence the low floor
    assert tot == 0 or hit / tot >= 0.7, (hit, tot)


@pytest.mark.parametrize("B,T,valid", [(2, 64, None), (3, 143, [143, 100, 1]), (2, 499, [499, 300]), (1, 700, None)])
def test_fp8_attention_core_op(B, T, valid):
    """the attention core on MXFP8 operands (csrc/attention.hip attention_f8_kernel; BASELINE configs[4] "fp8 MFMA for attention"):
    q, k quantised per 32 features, v per 32 keys, P to e4m3 in registers.  Against fp32 softmax attention on the SAME quantised
    operands (oracle quantiser) the error left is P's 3-bit mantissa (2.4e-2 relative RMS on Gaussian data, measured) and the
    bf16 output; against the unquantised result it is the fp8 noise of all four operands."""
    from sylber_amd import _lib
    from oracle import mxfp8_ref as Q
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 1000 + T)
    q = torch.randn(B, T, 768, generator=g); k = torch.randn(B, T, 768, generator=g); v = torch.randn(B, T, 768, generator=g)
    k[0, T // 2, :64] = 4.0 * q[0, 3, :64] / 8              # a spike: the running maximum really jumps between tiles
    vd = torch.tensor(valid, dtype=torch.int32).cuda() if valid else None
    o = torch.full((B, T, 768), float("nan"), device="cuda")
    qd, kd, vdev = q.cuda(), k.cuda(), v.cuda()
    _lib.check(lib.sylber_op_attention(_p(qd), _p(kd), _p(vdev), _p(vd), _p(o), B, T, 2, 0, None), "op_attention fp8")

    def fq(x, axis):
        x = np.moveaxis(x, axis, -1)
        d, sc = Q.quantize(np.ascontiguousarray(x).reshape(-1, x.shape[-1]))
        return np.moveaxis(Q.dequantize(d, sc).reshape(x.shape), -1, axis)

    def ref(qq, kk, vv):
        qh, kh, vh = (t.view(B, T, 12, 64).transpose(1, 2) for t in (qq, kk, vv))
        sc = qh @ kh.transpose(-1, -2)
        if valid:
            mask = torch.arange(T)[None, :] >= torch.tensor(valid)[:, None]
            sc = sc.masked_fill(mask[:, None, None, :], float("-inf"))
        return (torch.softmax(sc, -1) @ vh).transpose(1, 2).reshape(B, T, 768)
    qq = torch.from_numpy(fq((q * 0.125).numpy().reshape(B, T, 24, 32), -1).reshape(B, T, 768))
    kq = torch.from_numpy(fq(k.numpy().reshape(B, T, 24, 32), -1).reshape(B, T, 768))
    Tpad = (T + 31) // 32 * 32
    vpad = np.zeros((B, Tpad, 768), np.float32); vpad[:, :T] = v.numpy()
    vq = torch.from_numpy(fq(vpad.reshape(B, Tpad // 32, 32, 768), 2).reshape(B, Tpad, 768)[:, :T].copy())
    same, exact, got = ref(qq, kq, vq), ref(q * 0.125, k, v), o.cpu()
    assert torch.isfinite(got).all()
    rr = lambda a, b: float(((a - b).pow(2).mean() / b.pow(2).mean()).sqrt())
    assert rr(got, same) < 3.5e-2 and (got - same).abs().max().item() < 8e-2, (rr(got, same), (got - same).abs().max().item())
    assert rr(got, exact) < 8e-2, rr(got, exact)


def test_fp8_attention_core_in_the_forward():
    """SYLBER_OPT_FP8_ATTENTION: on (default) where the batch has whole 256-row tiles, the bf16 core otherwise; close to the bf16 core,
    deterministic, ragged lengths handled by the key mask"""
    from sylber_amd import HubertEncoderHIP
    from sylber_amd.synth import syllable_wave
    from sylber_amd.weights import synthetic_state_dict
    sd = synthetic_state_dict(0)
    e8 = HubertEncoderHIP(sd, precision="fp8")
    rel = lambda a, b: float(((a - b).pow(2).mean() / b.pow(2).mean()).sqrt())
    wav = torch.cat([syllable_wave(48000, 70 + i) for i in range(8)], 0).cuda()           # 8 x 149 frames: 8 x 160 rows = 5 tiles
    for lens in (None, [48000, 30000, 48000, 9000, 48000, 48000, 20000, 41000]):
        on = e8.forward(wav, lens).clone()
        assert torch.equal(on, e8.forward(wav, lens))
        e8.set_option(7, -1)
        off = e8.forward(wav, lens).clone()
        e8.set_option(7, 0)
        assert torch.isfinite(on).all() and not torch.equal(on, off)
        assert rel(on, off) < 4e-2, rel(on, off)                                           # measured 2.2e-2
    small = wav[:3].contiguous()                                                           # 480 rows: not whole 256-row tiles
    a = e8.forward(small, None).clone()
    # (regression: a smaller batch on a handle that served a larger one found stale scale bytes behind the padded query rows of
    #  the fp8 context and returned NaN from the second layer on -- csrc/attention.hip attn_finalize now writes those rows)
    assert torch.isfinite(a).all()
    # the core of the mode does not depend on the batch shape -- the q / k / v launch pads M up to whole tiles (rows
    # beyond the batch computed, not stored), so a ragged batch runs the MXFP8 core too and an utterance's hidden states are the
    # same bits alone, in a batch of 3 and in the batch of 8
    e8.set_option(7, -1)
    assert not torch.equal(a, e8.forward(small, None))
    e8.set_option(7, 0)
    full = e8.forward(wav, None).clone()
    assert torch.equal(a, full[:3])
    one = e8.forward(wav[5:6].contiguous(), None)
    assert torch.equal(one[0], full[5])


def test_fp8_same_clip_alone_and_in_a_batch():
    """one utterance, one result: ragged lengths, batch sizes that are / are not whole 256-row tiles, a poisoned workspace in between
    (the padded rows of the q / k / v launch read whatever follows the operand: their results must never be stored)"""
    from sylber_amd import HubertEncoderHIP
    from sylber_amd.synth import syllable_wave
    from sylber_amd.weights import synthetic_state_dict
    sd = synthetic_state_dict(0)
    e8 = HubertEncoderHIP(sd, precision="fp8")
    wav = torch.cat([syllable_wave(40000, 300 + i) for i in range(5)], 0).cuda()           # 5 x 124 frames -> 5 x 128 rows = 2.5 tiles
    lens = [40000, 23000, 40000, 8000, 31000]
    full = e8.forward(wav, lens).clone()
    assert torch.isfinite(full).all()
    from sylber_amd import _lib
    _lib.check(e8.lib.sylber_debug_poison_workspace(e8.handle, 0xFF), "poison")
    torch.cuda.synchronize()
    assert torch.equal(e8.forward(wav, lens), full)
    for i in (0, 3):
        # same Lmax (the batch is padded to its maximum: sylber.py:93-97), one row
        alone = e8.forward(wav[i:i + 1].contiguous(), [lens[i]])
        assert torch.equal(alone[0], full[i]), i
    pair = e8.forward(wav[1:3].contiguous(), lens[1:3])
    assert torch.equal(pair, full[1:3])


def test_fp8_full_size_config_vs_oracle_sample():
    """BASELINE configs[4] at the configs[1] batch (32 x 10 s): finite, reproducible, batch-independent rows, and two rows against the
    oracle within the mode's stated output tolerance (the fp8 mode had no counterpart of the bf16 full-size test)"""
    from oracle import hubert_ref
    from sylber_amd import HubertEncoderHIP
    from sylber_amd.synth import noise_batch
    from sylber_amd.weights import synthetic_state_dict
    sd = synthetic_state_dict(0)
    e8 = HubertEncoderHIP(sd, precision="fp8")
    x = noise_batch(32, 160000, seed=0)
    xd = x.cuda()
    out = e8.forward(xd).cpu().numpy()
    assert out.shape == (32, 499, 768) and np.isfinite(out).all()
    for _ in range(2):
        assert np.array_equal(e8.forward(xd).cpu().numpy(), out)
    part = e8.forward(xd[8:12].contiguous()).cpu().numpy()
    assert np.array_equal(part, out[8:12])
    ref = hubert_ref.forward(sd, x[[0, 31]], None)["hidden"].numpy().astype(np.float64)
    for row, r in ((0, ref[0]), (31, ref[1])):
        a = out[row].astype(np.float64)
        assert float(np.sqrt(((a - r) ** 2).mean() / (r ** 2).mean())) < FP8_STAGE_TOL["layer8"], row
