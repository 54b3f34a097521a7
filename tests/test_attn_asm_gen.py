"""The hand-scheduled attention key loop (tools/gen_attn_asm.py -> attn_asm_*.inc, csrc/attention.hip attention_asm_kernel) is
generated from a small IR.  CPU tier: (1) the IR passes the static issue-hazard checks the compiler cannot do inside an asm statement;
(2) the IR is EXECUTED by tools/attn_asm_emu.py -- four waves of one workgroup, LDS-DMA landing as early / as late as the waits allow,
fragment registers written only at the covering lgkmcnt wait -- and compared with softmax(q k^T + key mask) v in float64
(transformers eager_attention_forward TP:234-259 reached from sylber/model/sylber.py:122) on single / first+last / many-tile shapes,
ragged valid lengths and a score spike that sends the lazy maximum through its slow path; (3) the generator's output is pinned by hash."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def prog():
    import gen_attn_asm as G
    p = G.Gen("bf16").build()
    assert G.check_hazards(p.ins)
    return p


def test_f16_format_of_the_loop():
    """the fp16 operand format (precision="fp16"): same schedule, f16 MFMA / pack / reference grid"""
    import attn_asm_emu as E
    import gen_attn_asm as G
    p = G.Gen("f16").build()
    assert G.check_hazards(p.ins)
    rng = np.random.default_rng(3)
    q = rng.standard_normal((150, 64)).astype(np.float32) * 0.54
    k = rng.standard_normal((150, 64)).astype(np.float32)
    v = rng.standard_normal((150, 64)).astype(np.float32)
    k[70] = 16.0 * q[5]
    ref = E.reference(q, k, v, 131, fmt="f16")
    for qb in range(2):
        ctx, _ = E.run_workgroup(q, k, v, 131, qblock=qb, fmt="f16", prog=p)
        nq = min(128, 150 - qb * 128)
        assert np.abs(ctx[:nq] - ref[qb * 128:qb * 128 + nq]).max() < 4e-3


@pytest.mark.parametrize("T,nvalid", [(1, 1), (31, 31), (33, 32), (64, 33), (65, 65), (100, 70), (128, 97), (143, 1), (200, 200), (300, 193)])
def test_emulated_key_loop_matches_softmax_attention(prog, T, nvalid):
    import attn_asm_emu as E
    rng = np.random.default_rng(T * 1000 + nvalid)
    q = rng.standard_normal((T, 64)).astype(np.float32) * 0.54           # (pre-scaled by log2(e) / 8: scores in log2 units)
    k = rng.standard_normal((T, 64)).astype(np.float32)
    v = rng.standard_normal((T, 64)).astype(np.float32)
    if T > 40:
        k[T // 2] = 16.0 * q[3]                               # the running maximum of query 3 jumps mid-way: slow path + deferred rescale
    ref = E.reference(q, k, v, nvalid)
    for mode, order in (("late", None), ("early", [3, 2, 1, 0])):
        for qb in range((T + 127) // 128):
            ctx, _ = E.run_workgroup(q, k, v, nvalid, qblock=qb, mode=mode, order=order, prog=prog)
            nq = min(128, T - qb * 128)
            assert np.isfinite(ctx[:nq]).all(), (mode, qb)
            assert np.abs(ctx[:nq] - ref[qb * 128:qb * 128 + nq]).max() < 2.5e-2, (mode, qb)


def test_emulator_catches_a_missing_wait(prog):
    """the checker checks: with the s_waitcnt in front of the tile barriers removed, late-landing LDS-DMA data is read as poison"""
    import copy
    import attn_asm_emu as E
    import gen_attn_asm as G
    broken = copy.copy(prog)
    broken.ins = [d for d in prog.ins if not (d["kind"] == "waitcnt" and d["vm"] == 0 and d["lgkm"] == 0)]
    rng = np.random.default_rng(5)
    q, k, v = (rng.standard_normal((200, 64)).astype(np.float32) for _ in range(3))
    ctx, _ = E.run_workgroup(q * 0.18, k, v, 200, mode="late", prog=broken)
    assert not np.isfinite(ctx).all()


def test_hazard_checker_rejects_an_early_read_of_an_mfma_result(prog):
    import gen_attn_asm as G
    ins = list(prog.ins)
    i = next(j for j, d in enumerate(ins) if d["kind"] == "mfma" and d["dst"][1] == G.R["SB"])
    bad = ins[:i + 1] + [dict(kind="valu", op="v_max_f32", dst=G.v(G.R["mx"]), src=[G.v(G.R["SB"]), G.v(G.R["SB"] + 1)], trans=False)] + ins[i + 1:]
    with pytest.raises(AssertionError):
        G.check_hazards(bad)


def test_generated_text_matches_the_committed_hashes(tmp_path):
    env = dict(os.environ, GEN_GEMM_ASM_OUT=str(tmp_path))
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_attn_asm.py"), "product"], check=True, env=env, capture_output=True)
    want = json.load(open(os.path.join(ROOT, "tools", "attn_asm_hashes.json")))
    made = sorted(os.listdir(tmp_path))
    assert made == sorted(want)
    for f in made:
        got = hashlib.sha256(open(os.path.join(tmp_path, f), "rb").read()).hexdigest()
        assert got == want[f], "%s: the generator's output changed; if intended, run `python tools/gen_attn_asm.py hashes`" % f
