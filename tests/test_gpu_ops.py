"""GPU tier: single-kernel parity through the C-ABI op entry points against plain torch fp32 on
the same (bf16-rounded) inputs."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from sylber_amd import _lib
    assert torch.cuda.is_available(), "gpu tier needs the MI355X"
    return _lib.load()


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("M,N,K,act", [(128, 128, 64, 0), (500, 768, 768, 0), (1000, 3072, 768, 1),
                                      (333, 768, 3072, 0), (257, 512, 1536, 1), (64, 2304, 768, 0)])
def test_linear(lib, M, N, K, act):
    from sylber_amd import _lib
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ad, wd, bd = a.cuda(), w.cuda(), b.cuda()
    c = torch.full((M, N), float("nan"), device="cuda")
    _lib.check(lib.sylber_op_linear(_p(ad), _p(wd), _p(bd), _p(c), M, N, K, act, 0, -1, None), "op_linear")
    ref = _bf(a) @ _bf(w).T + b
    if act:
        ref = torch.nn.functional.gelu(ref)
    err = (c.cpu() - ref).abs().max().item()
    assert err < 2e-3, err     # same bf16 products, fp32 accumulation order differs; A-S gelu 1.5e-7


@pytest.mark.parametrize("D", [512, 768])
def test_layernorm(lib, D):
    from sylber_amd import _lib
    g = torch.Generator().manual_seed(D)
    x = torch.randn(777, D, generator=g) * 3 + 0.5
    r = torch.randn(777, D, generator=g)
    gam = torch.randn(D, generator=g)
    bet = torch.randn(D, generator=g)
    y = torch.empty(777, D, device="cuda")
    xd, rd, gd, bd = x.cuda(), r.cuda(), gam.cuda(), bet.cuda()
    _lib.check(lib.sylber_op_layernorm(_p(xd), _p(rd), _p(gd), _p(bd), _p(y), 777, D, None), "op_layernorm")
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x + r, (D,), gam, bet, 1e-5)
    assert (y.cpu() - ref).abs().max().item() < 2e-5


LOG2E = 1.4426950408889634
Q_SCALE = 0.125 * LOG2E


@pytest.mark.parametrize("qw", [0, 1, 2])       # 0: the default (hand-scheduled key loop), 1 / 2: the compiler-scheduled kernels (32 / 64 queries per wave)
@pytest.mark.parametrize("B,T,valid", [(2, 64, None), (3, 143, [143, 100, 1]), (2, 499, [499, 300]), (1, 700, None),
                                       (6, 499, None), (2, 1, None), (2, 33, [33, 32]), (3, 130, [130, 65, 64]), (1, 2999, None), (1, 6000, [4097]), (2, 193, [193, 129])])
def test_attention(lib, B, T, valid, qw):
    from sylber_amd import _lib
    g = torch.Generator().manual_seed(B * 1000 + T)
    q = torch.randn(B, T, 768, generator=g)
    k = torch.randn(B, T, 768, generator=g)
    v = torch.randn(B, T, 768, generator=g)
    # a query/key spike so that the running max really jumps between tiles (online-softmax rescale path)
    k[0, T // 2, :64] = 4.0 * q[0, 3 % T, :64] / 8
    vd = torch.tensor(valid, dtype=torch.int32).cuda() if valid else None
    o = torch.full((B, T, 768), float("nan"), device="cuda")
    qd, kd, vdev = q.cuda(), k.cuda(), v.cuda()
    _lib.check(lib.sylber_op_attention(_p(qd), _p(kd), _p(vdev), _p(vd), _p(o), B, T, 0, 32 * qw, None), "op_attention")
    # q is rounded to bf16 AFTER its pre-scaling by log2(e) / 8 (csrc/common.h SYL_Q_SCALE): the scores are in log2 units
    qh = (_bf(q * Q_SCALE) / LOG2E).view(B, T, 12, 64).transpose(1, 2)
    kh = _bf(k).view(B, T, 12, 64).transpose(1, 2)
    vh = _bf(v).view(B, T, 12, 64).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    if valid:
        mask = torch.arange(T)[None, :] >= torch.tensor(valid)[:, None]
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, T, 768)
    err = (o.cpu() - ref).abs().max().item()
    assert err < 3e-2, err     # P and the output are rounded to bf16 (rel 3.9e-3) around O(1) values
    assert (o.cpu() - ref).pow(2).mean().sqrt().item() < 4e-3


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 10, 11, 40, 51, 57, 60, 80, 85, 90, 91, 95, 97])
def test_linear_every_tile_config(lib, cfg):
    """every GEMM tile configuration (4-wave 2/3-slot rings, 8-wave staggered big tiles) gives the same
    result, including ragged M / N tails and a strided (overlapping-row) activation operand"""
    from sylber_amd import _lib
    g = torch.Generator().manual_seed(7)
    for (M, N, K) in [(700, 768, 768), (1000, 512, 1536), (333, 3072, 768), (257, 768, 3072)]:
        a = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        ad, wd, bd = a.cuda(), w.cuda(), b.cuda()
        c = torch.full((M, N), float("nan"), device="cuda")
        _lib.check(lib.sylber_op_linear(_p(ad), _p(wd), _p(bd), _p(c), M, N, K, 1, 0, cfg, None), "op_linear")
        ref = torch.nn.functional.gelu(_bf(a) @ _bf(w).T + b)
        assert (c.cpu() - ref).abs().max().item() < 2e-3, (cfg, M, N, K)


@pytest.mark.parametrize("cfg", [-1, 3, 4, 10])
def test_linear_split16(lib, cfg):
    """precision split16 (operands as hi / lo IEEE-half planes, three MFMA passes into one fp32 accumulator) is an
    fp32-grade contraction: against float64 it is within fp32 accumulation noise, three orders below the bf16 / fp16
    operand rounding, for every tile shape the mode instantiates (ragged tails, small and large K)"""
    from sylber_amd import _lib
    g = torch.Generator().manual_seed(5)
    for (M, N, K) in [(700, 768, 768), (1000, 512, 1536), (333, 3072, 768), (257, 768, 3072), (16384, 768, 768)]:
        a = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        ad, wd, bd = a.cuda(), w.cuda(), b.cuda()
        c = torch.full((M, N), float("nan"), device="cuda")
        _lib.check(lib.sylber_op_linear(_p(ad), _p(wd), _p(bd), _p(c), M, N, K, 0, 5, cfg, None), "op_linear")
        ref = a.double() @ w.double().T + b.double()
        err = (c.cpu().double() - ref).abs().max().item()
        assert err < 2e-5, (cfg, M, N, K, err)          # |ref| ~ 1; fp16 operands would give ~2e-3, bf16 ~2e-2
        f32 = (a @ w.T + b).double()
        assert err < 12 * (f32 - ref).abs().max().item() + 2e-6, (cfg, M, N, K)     # fp32 accumulation noise (a longer serial chain than the CPU's blocked sums)


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 6, 40, 51, 57, 60, 80, 85, 90, 91, 95, 97])
def test_gemm8_schedule_variants_bitwise(lib, cfg):
    """the K-loop schedule variants of the 8-wave kernel (where the LDS-DMA of step s+3 is issued) contract in the same
    order: bit-identical to the default schedule on a full-size launch, run to run (a hand-off race would show here)"""
    from sylber_amd import _lib
    g = torch.Generator().manual_seed(23)
    M, N, K = 16384, 3072, 768
    ad = torch.randn(M, K, generator=g).cuda(); wd = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); bd = torch.randn(N, generator=g).cuda()
    ref = torch.empty(M, N, device="cuda")
    _lib.check(lib.sylber_op_linear(_p(ad), _p(wd), _p(bd), _p(ref), M, N, K, 1, 0, 9010, None), "op_linear")
    for _ in range(3):
        c = torch.full((M, N), float("nan"), device="cuda")
        _lib.check(lib.sylber_op_linear(_p(ad), _p(wd), _p(bd), _p(c), M, N, K, 1, 0, 9000 + cfg, None), "op_linear")
        assert torch.equal(c, ref), cfg


def test_attention_full_batch_no_race(lib):
    """32 x 499 (1536 workgroups): the whole tensor against torch and bitwise run-to-run reproducibility.
    (Catches LDS-DMA hand-off races that small grids do not expose.)"""
    from sylber_amd import _lib
    B, T = 32, 499
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, T, 768, generator=g); k = torch.randn(B, T, 768, generator=g); v = torch.randn(B, T, 768, generator=g)
    qd, kd, vdev = q.cuda(), k.cuda(), v.cuda()
    qh = (_bf(q * Q_SCALE) / LOG2E).view(B, T, 12, 64).transpose(1, 2)
    kh = _bf(k).view(B, T, 12, 64).transpose(1, 2)
    vh = _bf(v).view(B, T, 12, 64).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2), -1) @ vh).transpose(1, 2).reshape(B, T, 768)
    for qw in (0, 1, 2):
        outs = []
        for _ in range(3):
            o = torch.full((B, T, 768), float("nan"), device="cuda")
            _lib.check(lib.sylber_op_attention(_p(qd), _p(kd), _p(vdev), None, _p(o), B, T, 0, 32 * qw, None), "op_attention")
            outs.append(o.cpu())
        assert (outs[0] - ref).abs().max().item() < 3e-2, qw
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), qw


@pytest.mark.parametrize("per_cu", [1, 2])
def test_linear_persistent_grid(lib, per_cu):
    """the 4-wave GEMM as a persistent launch (k x 256 workgroups walking the tile list) gives the same result"""
    from sylber_amd import _lib
    g = torch.Generator().manual_seed(3)
    M, N, K = 70000, 768, 768          # 547 x 4 tiles of 128x192 -> several tiles per workgroup
    a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    ad, wd, bd = a.cuda(), w.cuda(), b.cuda()
    c = torch.full((M, N), float("nan"), device="cuda")
    _lib.check(lib.sylber_op_linear(_p(ad), _p(wd), _p(bd), _p(c), M, N, K, 0, 0, 1000 * per_cu + 4, None), "op_linear")
    ref = _bf(a) @ _bf(w).T + b
    assert (c.cpu() - ref).abs().max().item() < 2e-3


@pytest.mark.parametrize("act", [0, 1])
def test_linear_persistent_prefetch_kernel(lib, act):
    """the persistent 256x256 kernel with cross-tile operand prefetch (whole tiles, more than 256 of them: several tiles
    per workgroup, seams included) returns bit for bit what the one-tile-per-workgroup kernel returns"""
    from sylber_amd import _lib
    g = torch.Generator().manual_seed(21)
    for (M, N, K) in [(256 * 300, 512, 1536), (256 * 130, 768, 128), (256 * 90, 1024, 256)]:
        a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
        ad, wd, bd = a.cuda(), w.cuda(), b.cuda()
        outs = []
        for tile in (10, 9010):                       # 9010 = tile 10, never persistent
            c = torch.full((M, N), -1, dtype=torch.int16, device="cuda")
            _lib.check(lib.sylber_op_linear16(_p(ad), _p(wd), _p(bd), _p(c), M, N, K, act, 0, tile, None), "op_linear16")
            outs.append(c)
        assert torch.equal(outs[0], outs[1]), (M, N, K)
        rows = torch.randint(0, M, (512,))
        got = outs[0][rows.cuda()].view(torch.bfloat16).float().cpu()
        ref = _bf(a[rows]) @ _bf(w).T + b
        if act:
            ref = torch.nn.functional.gelu(ref)
        assert (got - ref).abs().max().item() < 2e-2 and (got - ref).pow(2).mean().sqrt().item() < 3e-3


def test_conv3_layer_every_tile(lib):
    """a 3-tap stride-2 conv layer of the feature extractor (TP:160-175) as the forward runs it -- implicit GEMM in the chunk-major
    K order (GemmArgs::kpat: tap 0, tap 2, tap 1 per 64-channel chunk) -- against torch's conv1d, and bit for bit the same on every
    kernel that can run it: the 8-wave asm tile (97, the K order in generated asm), the hipcc-scheduled tiles (3, 4, 10, 11, 40) and
    a forced asm tile without that order (85 -> documented fallback): results must not depend on the tile a batch size picks"""
    from sylber_amd import _lib
    g = torch.Generator().manual_seed(77)
    for M in (70000, 3000, 257):
        R = 2 * M + 1
        x = torch.randn(R, 512, generator=g)
        w = torch.randn(512, 512, 3, generator=g) / (3 * 512) ** 0.5
        xd = x.cuda()
        wc = w.contiguous()
        outs = {}
        # (tile + 1000000 = the 32x32x16 kernels this test was written for: since round 6 a 16-bit-output launch maps a forced id into the 16x16x32
        #  family, whose own sweep is test_mfma16_role_16bit_outputs; 1000999 = their automatic tile)
        L = 1000000
        for tile in (L + 9010, L + 97, L + 1, L + 2, L + 3, L + 4, L + 5, L + 6, L + 11, L + 40, L + 85, L + 999):
            y = torch.full((M, 512), -1, dtype=torch.int16, device="cuda")
            _lib.check(lib.sylber_op_conv3(_p(xd), ctypes.c_void_p(wc.data_ptr()), _p(y), R, M, tile, None), "op_conv3")
            outs[tile - L] = y
            assert torch.equal(y, outs[9010]), (M, tile)
        rows = torch.randint(0, M, (256,))
        got = outs[97][rows.cuda()].view(torch.bfloat16).float().cpu()
        xr = torch.stack([_bf(x[2 * rows + t]) for t in range(3)], -1)            # [256, 512 in, 3]
        ref = torch.nn.functional.gelu(torch.einsum("rct,oct->ro", xr, _bf(w)))
        assert (got - ref).abs().max().item() < 2e-2 and (got - ref).pow(2).mean().sqrt().item() < 3e-3, M


def test_mfma16_family(lib):
    """the v_mfma_f32_16x16x32 family (csrc/gemm_asm16.hip: tiles 13 / 14 hipcc-scheduled two per CU on four waves, 15 / 16 the same on eight, 17 = 64x64 on a
    three-slot ring -- incl. K of one to four K steps, where its prologue paths differ --, 46 / 47 generated loops) -- the kernels the
    16-bit-output GEMMs (conv1-5 TP:154-213, FFN1 TP:347-368) run on by default.  An output element's fp32 chain adds 32-k blocks where the
    32x32x16 kernels add 16-k blocks, so: every member gives the SAME bits (whole and ragged tiles, several rounds of the persistent walk, with and
    without GELU), run to run; against torch at the tolerance of the other tiles; against tile 97 within fp32 accumulation noise"""
    from sylber_amd import _lib
    g = torch.Generator().manual_seed(47)
    for (M, N, K, act) in [(700, 768, 768, 1), (1000, 512, 1536, 1), (333, 3072, 768, 0), (257, 768, 3072, 1), (16384, 3072, 768, 1), (4096, 4096, 4096, 0),
                           (24064, 3072, 768, 1), (130, 512, 64, 1), (200, 512, 128, 1), (200, 512, 192, 0), (300, 768, 256, 1)]:
        a = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        ad, wd, bd = a.cuda(), w.cuda(), b.cuda()
        outs = {}
        for cfg in (47, 97, 46, 13, 14, 15, 16, 17, 47, 9047):
            if K < 256 and cfg in (46, 47, 97, 9047):
                continue                                     # the generated loops need four K steps; the small tiles take any K % 64 == 0
            c = torch.full((M, N), float("nan"), device="cuda")
            _lib.check(lib.sylber_op_linear(_p(ad), _p(wd), _p(bd), _p(c), M, N, K, act, 0, cfg, None), "op_linear")
            outs.setdefault(cfg, c)
            if cfg != 97:
                first = next(v for k_, v in outs.items() if k_ != 97)
                assert torch.equal(c, first), ("family member / run to run", cfg, M, N, K)
        rows = torch.randint(0, M, (min(M, 512),))
        ref = _bf(a[rows]) @ _bf(w).T + b
        if act:
            ref = torch.nn.functional.gelu(ref)
        got = outs[14][rows.cuda()].cpu()
        assert (got - ref).abs().max().item() < (2e-3 if K < 4096 else 2e-2), (M, N, K)
        if 97 in outs:
            d = (outs[14] - outs[97]).abs().max().item()
            assert d <= 2e-5 * outs[97].abs().max().item() + 1e-6, (M, N, K, d)


def test_mfma16_role_16bit_outputs(lib):
    """the role itself (EPI_BF16 through sylber_op_linear16 / sylber_op_conv3): whatever tile id is forced, a 16-bit-output launch runs on a member of
    the 16x16x32 family and returns the same bits; tile + 1000000 (GemmArgs::tune_mfma16 = -1) puts it back on the 32x32x16 kernels, whose 16-bit
    outputs differ from the family's by at most one rounding step of the stored format on a small fraction of the elements"""
    from sylber_amd import _lib
    g = torch.Generator().manual_seed(48)
    for (M, N, K) in [(16384, 3072, 768), (1000, 512, 1024), (24064, 3072, 768), (6437, 3072, 768), (256 * 90, 1024, 512)]:     # (persistent seams, whole and ragged)
        a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
        ad, wd, bd = a.cuda(), w.cuda(), b.cuda()
        outs = {}
        for tile in (-1, 47, 46, 13, 14, 15, 16, 17, 1, 97, 57, 3, 4, 10, 85, 91, 1000097, 1000004, 1000999):
            c = torch.full((M, N), -1, dtype=torch.int16, device="cuda")
            _lib.check(lib.sylber_op_linear16(_p(ad), _p(wd), _p(bd), _p(c), M, N, K, 1, 0, tile, None), "op_linear16")
            outs[tile] = c
        for tile in (47, 46, 13, 14, 15, 16, 17, 1, 97, 57, 3, 4, 10, 85, 91):
            assert torch.equal(outs[tile], outs[-1]), (tile, M, N, K)
        assert torch.equal(outs[1000097], outs[1000004]) and torch.equal(outs[1000097], outs[1000999]), (M, N, K)
        f, l = outs[-1].view(torch.bfloat16).float(), outs[1000097].view(torch.bfloat16).float()
        d = (f - l).abs()
        assert (d > 0).float().mean().item() < 0.02 and d.max().item() <= 2.0 ** -7 * l.abs().max().item(), (M, N, K)
        ref = torch.nn.functional.gelu(_bf(a[:256]) @ _bf(w).T + b)
        assert (f[:256].cpu() - ref).abs().max().item() < 3e-2
    # the 3-tap stride-2 conv layers' chunk-major K order (TP:160-175): family against torch's conv and against the legacy kernels
    for M in (70000, 257):
        R = 2 * M + 1
        x = torch.randn(R, 512, generator=g)
        w = torch.randn(512, 512, 3, generator=g) / (3 * 512) ** 0.5
        xd = x.cuda()
        wc = w.contiguous()
        ys = {}
        for tile in (-1, 47, 13, 14, 15, 16, 17, 1000097):
            y = torch.full((M, 512), -1, dtype=torch.int16, device="cuda")
            _lib.check(lib.sylber_op_conv3(_p(xd), ctypes.c_void_p(wc.data_ptr()), _p(y), R, M, tile, None), "op_conv3")
            ys[tile] = y
            if tile > 0 and tile < 1000000:
                assert torch.equal(y, ys[-1]), (tile, M)
        f, l = ys[-1].view(torch.bfloat16).float(), ys[1000097].view(torch.bfloat16).float()
        assert ((f - l).abs() > 0).float().mean().item() < 0.02, M
        rows = torch.randint(0, M, (256,))
        got = f[rows.cuda()].cpu()
        xr = torch.stack([_bf(x[2 * rows + t]) for t in range(3)], -1)
        ref = torch.nn.functional.gelu(torch.einsum("rct,oct->ro", xr, _bf(w)))
        assert (got - ref).abs().max().item() < 2e-2 and (got - ref).pow(2).mean().sqrt().item() < 3e-3, M


@pytest.mark.parametrize("tile", [1, 2, 4, 5, 6, 51, 90, 91, 96])
def test_residual_gemm_tiles(lib, tile):
    """the residual GEMM of an encoder block (out-projection K = 768, FFN2 K = 3072: EPI_F32_RESLN, in place) against torch, and
    bit for bit against the HIP-scheduled 128x192 kernel -- tile 91 with whole tiles runs the K loop that also prefetches the
    residual rows into registers (gemm_asm_x3_n3_p4 / _p7), 96 the same tile without it; a ragged M falls back to the plain loop"""
    from sylber_amd import _lib
    g = torch.Generator().manual_seed(41)
    for (M, N, K) in [(16384, 768, 768), (16384, 768, 3072), (4096, 768, 1536), (1000, 768, 768), (512, 384, 1152)]:
        a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
        pre = torch.randn(M, N, generator=g) * 2.0 + 0.3
        gamma = 1.0 + 0.2 * torch.randn(N, generator=g); beta = 0.1 * torch.randn(N, generator=g)
        mean = pre.mean(-1); rstd = (pre.var(-1, unbiased=False) + 1e-5).rsqrt()
        stats = torch.stack([mean, rstd], -1).contiguous()
        ad, wd, bd, gd, bed, sd_ = a.cuda(), w.cuda(), b.cuda(), gamma.cuda(), beta.cuda(), stats.cuda()
        outs = []
        for t in (9004, tile, tile):
            c = pre.clone().cuda()
            _lib.check(lib.sylber_op_linear_resln(_p(ad), _p(wd), _p(bd), _p(c), _p(sd_), _p(gd), _p(bed), M, N, K, t, None), "op_linear_resln")
            outs.append(c)
        assert torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0]), (tile, M, N, K)
        ref = _bf(a) @ _bf(w).T + b + ((pre - mean[:, None]) * rstd[:, None] * gamma + beta)
        assert (outs[1].cpu() - ref).abs().max().item() < 3e-3, (tile, M, N, K)


@pytest.mark.parametrize("tile", [51, 57, 80, 85, 86, 90, 91, 95, 97])
def test_asm_tiles_persistent_seams(lib, tile):
    """the hand-scheduled kernels as persistent workgroups (more tiles than CUs: the next tile's operands requested before
    the epilogue, counted waits across its stores) return bit for bit what the HIP 8-wave kernel returns, with whole tiles
    and with a ragged last row tile (where the seam wait drains the stores instead of counting them), run to run"""
    from sylber_amd import _lib
    g = torch.Generator().manual_seed(33)
    for (M, N, K, act) in [(256 * 90, 1024, 512, 1), (6437, 3072, 768, 1), (256 * 70, 1536, 256, 0)]:
        a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
        ad, wd, bd = a.cuda(), w.cuda(), b.cuda()
        ref = torch.full((M, N), -1, dtype=torch.int16, device="cuda")
        # (+ 1000000: on the 32x32x16 kernels -- a 16-bit-output launch otherwise maps the forced id into the 16x16x32 family, test_mfma16_role_16bit_outputs)
        _lib.check(lib.sylber_op_linear16(_p(ad), _p(wd), _p(bd), _p(ref), M, N, K, act, 0, 1009010, None), "op_linear16")
        for _ in range(3):
            c = torch.full((M, N), -1, dtype=torch.int16, device="cuda")
            _lib.check(lib.sylber_op_linear16(_p(ad), _p(wd), _p(bd), _p(c), M, N, K, act, 0, 1000000 + tile, None), "op_linear16")
            assert torch.equal(c, ref), (tile, M, N, K)


NOSPLIT, TAIL = 100000, lambda t: 100000 * (t + 2)      # `tile` argument: + 100000 = never split by rows; + 100000 (t + 2) = split, tail on tile t


@pytest.mark.parametrize("tail", [3, 4, 0, 91, 97, 86, 51, 57])
def test_tail_split_is_bitwise(lib, tail):
    """round 6 tail policy (gemm_bf16.hip launch_f): a launch whose tile count is not a whole number of rounds of 256 persistent workgroups
    runs the rows of its full rounds on the chosen tile and the remaining rows, as a second launch over rows [M1, M), on another tile.  Every
    output element is one fp32 chain over K in the same order whatever tile computes it, so the split changes no bit: 16-bit GELU epilogue
    (FFN1 / conv shape), the in-place fp32 residual epilogue (out-proj / FFN2, incl. the loop that prefetches the residual rows), the plain
    fp32 epilogue and the 3-tap conv K order, each with whole and with ragged last row tiles, forced main tile and automatic."""
    from sylber_amd import _lib
    g = torch.Generator().manual_seed(600 + tail)
    # ---- EPI_BF16 + GELU: 94 / 75.x row tiles x 12 column tiles of 256 x 256 = 4.4 / 3.5 rounds
    for (M, N, K, main) in [(256 * 94, 3072, 768, 97), (19237, 3072, 768, 97), (256 * 94, 3072, 768, -1), (256 * 37, 2304, 256, 91)]:
        a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
        ad, wd, bd = a.cuda(), w.cuda(), b.cuda()
        outs = []
        for t in (9010 + NOSPLIT, (main if main >= 0 else 0) + (TAIL(tail) if main >= 0 else 0) + (-1 if main < 0 else 0)):
            c = torch.full((M, N), -1, dtype=torch.int16, device="cuda")
            _lib.check(lib.sylber_op_linear16(_p(ad), _p(wd), _p(bd), _p(c), M, N, K, 1, 0, t, None), "op_linear16")
            outs.append(c)
        assert torch.equal(outs[0], outs[1]), ("bf16", M, N, K, main, tail)
    # ---- EPI_F32_RESLN in place: 1.47 rounds of 256 x 192 (the 8 x 60 s out-projection / FFN2) and a ragged M
    for (M, N, K) in [(256 * 94, 768, 768), (256 * 94, 768, 3072), (256 * 70 + 100, 768, 768)]:
        a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
        pre = torch.randn(M, N, generator=g) * 2.0 + 0.3
        gamma = 1.0 + 0.2 * torch.randn(N, generator=g); beta = 0.1 * torch.randn(N, generator=g)
        mean = pre.mean(-1); rstd = (pre.var(-1, unbiased=False) + 1e-5).rsqrt()
        stats = torch.stack([mean, rstd], -1).contiguous()
        ad, wd, bd, gd, bed, sd_ = a.cuda(), w.cuda(), b.cuda(), gamma.cuda(), beta.cuda(), stats.cuda()
        outs = []
        for t in (9004 + NOSPLIT, 91 + TAIL(tail), -1):
            c = pre.clone().cuda()
            _lib.check(lib.sylber_op_linear_resln(_p(ad), _p(wd), _p(bd), _p(c), _p(sd_), _p(gd), _p(bed), M, N, K, t, None), "op_linear_resln")
            outs.append(c)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), ("resln", M, N, K, tail)
    # ---- plain fp32 epilogue
    M, N, K = 256 * 41 + 17, 2304, 768
    a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    ad, wd, bd = a.cuda(), w.cuda(), b.cuda()
    outs = []
    for t in (9004 + NOSPLIT, 85 + TAIL(tail)):
        c = torch.full((M, N), float("nan"), device="cuda")
        _lib.check(lib.sylber_op_linear(_p(ad), _p(wd), _p(bd), _p(c), M, N, K, 0, 0, t, None), "op_linear")
        outs.append(c)
    assert torch.equal(outs[0], outs[1]), ("f32", tail)
    # ---- 3-tap conv K order (conv5 of 8 x 60 s: 1.47 rounds of 256 x 256); a tail tile without that order falls back as documented
    M = 256 * 188
    R = 2 * M + 1
    x = torch.randn(R, 512, generator=g); wc = (torch.randn(512, 512, 3, generator=g) / (3 * 512) ** 0.5).contiguous()
    xd = x.cuda()
    outs = []
    for t in (9010 + NOSPLIT, 97 + TAIL(tail), -1):
        y = torch.full((M, 512), -1, dtype=torch.int16, device="cuda")
        _lib.check(lib.sylber_op_conv3(_p(xd), ctypes.c_void_p(wc.data_ptr()), _p(y), R, M, t, None), "op_conv3")
        outs.append(y)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), ("conv3", tail)
