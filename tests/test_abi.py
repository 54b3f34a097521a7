"""CPU tier: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/sylber_hip.h declares (no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from sylber_amd import build, _lib
    build.build()
    return _lib.load()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "sylber_hip.h")).read()
    declared = set(re.findall(r"\b(sylber_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("sylber_ctx")
    assert {"sylber_create", "sylber_forward", "sylber_segment", "sylber_destroy", "sylber_last_error"} <= declared
    for name in sorted(declared):
        assert hasattr(lib, name), name
    from sylber_amd import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    # development aids live in their own header and are not part of the drop-in ABI
    dev = open(os.path.join(ROOT, "include", "sylber_hip_dev.h")).read()
    dev_declared = set(re.findall(r"\b(sylber_[a-z0-9_]+)\s*\(", dev))
    assert dev_declared == set(_lib.DEV_EXPORTS) and not (dev_declared & declared)
    for name in sorted(dev_declared):
        assert hasattr(lib, name), name


def test_no_process_global_tuning_state():
    """tuning overrides are per handle (sylber_set_option) or per call: the old process-global force switches are gone"""
    for f in ("gemm_bf16.hip", "gemm_mxfp8.hip", "attention.hip", "api.hip"):
        src = open(os.path.join(ROOT, "sylber_amd", "csrc", f)).read()
        assert "force_cfg" not in src and "g_force" not in src and "xpad_rows &" not in src, f


def test_num_frames_matches_conv_formula(lib):
    assert lib.sylber_num_frames(160000) == 499
    assert lib.sylber_num_frames(46080) == 143
    assert lib.sylber_num_frames(960000) == 2999
    assert lib.sylber_num_frames(400) == 1


def test_product_path_refuses_cpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sylber_amd import Segmenter, _lib
    with pytest.raises(_lib.SylberHipError):
        Segmenter(model_ckpt=None)
    with pytest.raises(_lib.SylberHipError):
        Segmenter(model_ckpt=None, device="cpu")


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under sylber_amd/ may import it."""
    pkg = os.path.join(ROOT, "sylber_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "segment_ref" not in src and "hubert_ref" not in src, f


def test_bench_cli_parses():
    """bench.py's argument parser must build (a duplicated flag once broke the default run)"""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--gpus" in r.stdout and "--steps" in r.stdout and "--warmup" in r.stdout


def test_no_packed_fp32_valu_in_the_library(lib):
    """v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 are banned from every kernel: beside MFMA waves of another kernel on
    the same SIMD they returned wrong values in lanes 48-63 (profiles/r02_packed_f32_hazard.md)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    from sylber_amd import _lib
    text = check_isa.device_disassembly(_lib.LIB_PATH)                  # raises unless it really is the library's ISA
    n_dma = len(re.findall(r"\bbuffer_load_dwordx4\b[^\n]*\blds\b", text)) + len(re.findall(r"\bglobal_load_lds_dwordx4\b", text))
    assert len(re.findall(r"\bv_mfma_", text)) > 1000 and n_dma > 1000          # LDS-DMA staging (buffer descriptors since round 3)
    assert check_isa.banned_instructions(_lib.LIB_PATH) == {}


def test_ingest_rejects_rates_without_a_compact_resampling_table(lib):
    """host arithmetic only (no GPU): a corrupt WAVE header must not make the resampler plan a multi-GB polyphase table (a C++
    bad_alloc would cross the C ABI); common rates stay supported"""
    for sr, n16 in ((16000, 100000), (8000, 200000), (44100, 36282), (48000, 33334), (22050, 72563), (96000, 16667), (11025, 145125)):
        assert lib.sylber_ingest_num_frames(100000, sr) == n16 and lib.sylber_ingest_workspace_bytes(sr) > 0
    for sr in (0, -5, 44101, 15999, 2147483647):
        assert lib.sylber_ingest_num_frames(100000, sr) == -1 and lib.sylber_ingest_workspace_bytes(sr) == -1


def test_gemm_tile_cost_model_picks():
    """the tile cost model of the 16-bit GEMM launcher (csrc/gemm_bf16.hip TileModel) is host arithmetic: pinned here on the CPU tier so that an
    edit of its table cannot silently move the headline's kernels (round 6: a trailing comment once swallowed tile 91's row and FFN2 went to the
    256x256 four-wave tile, 0.73 -> 0.87 ms per forward).  Model 0 = the handle owns the chip, 5 = it shares it (two batches in flight)."""
    from sylber_amd import _lib
    lib = _lib.load()
    pick = lib.sylber_debug_gemm_pick
    QK, RESLN, BF16, GELU = 3, 6, 0, 1
    M = 32 * 512                                           # 32 x 10 s: 64 row tiles of 256
    # the headline batch: exactly 3 rounds (FFN1, q,k,v) / 1 round (out-proj, FFN2) of 256 tiles
    # (16-bit-output launches: the v_mfma_f32_16x16x32 family, the model offers ids 14 / 15 / 17 / 46 / 47; model + 100 = SYLBER_OPT_GEMM_MFMA16 -1, the 32x32x16 kernels)
    assert pick(M, 3072, 768, BF16, GELU, 0, 0, 0) == 47 and pick(M, 3072, 768, BF16, GELU, 0, 5, 0) == 47
    assert pick(M, 3072, 768, BF16, GELU, 0, 100, 0) == 97 and pick(M, 3072, 768, BF16, GELU, 0, 105, 0) == 97
    assert pick(M, 768, 3072, RESLN, 0, 0, 0, 0) == 91 and pick(M, 768, 3072, RESLN, 0, 0, 5, 0) == 91
    assert pick(M, 768, 768, RESLN, 0, 0, 0, 0) == 91 and pick(M, 768, 768, RESLN, 0, 0, 5, 0) == 91
    assert pick(M, 2304, 768, QK, 0, 0, 0, 0) == 4 and pick(M, 2304, 768, QK, 0, 0, 5, 0) == 91
    assert pick(M * 32, 512, 1536, BF16, GELU, 0, 0, 1) == 47          # conv1 in its chunk-major K order
    assert pick(M * 32, 512, 1536, BF16, GELU, 0, 100, 1) == 97        # (32x32x16 kernels: the only generated tile that has it)
    # 8 x 60 s (94 row tiles: 1.47 rounds of 256x192): alone on the chip the 192-row tile fills its second round, sharing it the round-5 choice stays
    ML = 8 * 3008
    assert pick(ML, 768, 3072, RESLN, 0, 0, 0, 0) == 51 and pick(ML, 768, 3072, RESLN, 0, 0, 5, 0) == 91
    assert pick(ML, 768, 768, RESLN, 0, 0, 0, 0) == 51
    assert pick(ML, 3072, 768, BF16, GELU, 0, 0, 0) == 47 and pick(ML, 3072, 768, BF16, GELU, 0, 100, 0) == 97
    assert pick(ML, 2304, 768, QK, 0, 0, 0, 0) == 4
    # 24 x 10 s (48 row tiles): 64 row tiles of 192 = exactly one round
    assert pick(24 * 512, 768, 3072, RESLN, 0, 0, 0, 0) == 51 and pick(24 * 512, 3072, 768, BF16, GELU, 0, 0, 0) == 46 and pick(24 * 512, 3072, 768, BF16, GELU, 0, 100, 0) == 57
    # small batches: 64-row tiles for the launches of one or two clips (fewer tiles than workgroup slots), not beyond (model + 6 = the round-6a choice)
    assert pick(512, 768, 3072, RESLN, 0, 0, 0, 0) == 1 and pick(512, 768, 768, RESLN, 0, 0, 0, 0) == 1 and pick(512, 2304, 768, QK, 0, 0, 0, 0) == 1
    assert pick(1024, 2304, 768, QK, 0, 0, 0, 0) == 2 and pick(2048, 768, 3072, RESLN, 0, 0, 0, 0) == 1 and pick(2048, 2304, 768, QK, 0, 0, 0, 0) == 3
    assert pick(512, 768, 3072, RESLN, 0, 0, 6, 0) == 51 and pick(8192, 768, 3072, RESLN, 0, 0, 0, 0) == 3
    # small batches of the 16-bit-output role: the 128x128 tile on eight waves
    assert pick(1024, 3072, 768, BF16, GELU, 0, 0, 0) == 15 and pick(8192, 512, 1536, BF16, GELU, 0, 0, 1) == 15 and pick(512, 3072, 768, BF16, GELU, 0, 100, 0) in (1, 2, 3, 4)
    # ... and its 64x64 three-slot form where a launch has at most ~one tile per workgroup slot (the conv layers of one or two clips)
    assert pick(2048, 512, 1536, BF16, GELU, 0, 0, 1) == 17 and pick(1024, 512, 1024, BF16, GELU, 0, 0, 0) == 17 and pick(512, 3072, 768, BF16, GELU, 0, 0, 0) == 17
    assert pick(2048, 512, 1536, BF16, GELU, 0, 6, 1) == 47
    # fp16 has the same tiles for the epilogues its forward launches
    assert pick(M, 768, 3072, RESLN, 0, 1, 0, 0) == 91 and pick(ML, 768, 3072, RESLN, 0, 1, 0, 0) == 51
    # every answer is a tile id that exists
    for m in (256, 1000, 6144, 12288, 16384, 24064, 49152):
        for (n, k, e, a) in ((2304, 768, QK, 0), (768, 768, RESLN, 0), (3072, 768, BF16, GELU), (768, 3072, RESLN, 0), (512, 1024, BF16, GELU)):
            for model in (0, 5, 100, 105):
                ids = (14, 15, 17, 46, 47) if (e == BF16 and model < 100) else (1, 2, 3, 4, 10, 85, 91, 97, 86, 51, 57, 80, 90, 95)
                assert pick(m, n, k, e, a, 0, model, 0) in ids, (m, n, k, e, model)
