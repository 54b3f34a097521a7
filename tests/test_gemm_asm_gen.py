"""The inline-asm K loops (gemm_asm*.inc) are GENERATED at build time by tools/gen_gemm_asm.py into sylber_amd/build/gen/ (round 5:
the 33 k generated lines are no longer committed).  What IS committed is the generator and a hash per loop
(tools/gemm_asm_hashes.json): a change of the generator that changes a shipped schedule has to update the list in the same commit,
so the schedule documented in the generator stays the one that ships.  `python tools/gen_gemm_asm.py hashes` rewrites the list."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "tools", "gen_gemm_asm.py")


def _generate(tmp_path, what):
    env = dict(os.environ, GEN_GEMM_ASM_OUT=str(tmp_path))
    subprocess.run([sys.executable, GEN, what], check=True, env=env, capture_output=True)
    return sorted(os.listdir(tmp_path))


def test_generated_loops_match_the_committed_hashes(tmp_path):
    made = _generate(tmp_path, "product")
    assert len(made) >= 7 and "gemm_asm_x3_w8.inc" in made and "gemm_asm_k64.inc" in made and "gemm_asm_loop.inc" in made
    want = json.load(open(os.path.join(ROOT, "tools", "gemm_asm_hashes.json")))
    assert sorted(want) == made, (set(want) ^ set(made))
    for f in made:
        got = hashlib.sha256(open(os.path.join(tmp_path, f), "rb").read()).hexdigest()
        assert got == want[f], "%s: the generator's output changed; if intended, run `python tools/gen_gemm_asm.py hashes`" % f
    # nothing generated is committed beside the sources any more
    csrc = os.path.join(ROOT, "sylber_amd", "csrc")
    assert not [f for f in os.listdir(csrc) if f.endswith(".inc")]


def test_experiment_loops_are_generated_not_committed(tmp_path):
    """the knock-out / schedule variants (timing only, results wrong by construction) are generated into the build directory by a
    SYLBER_EXPERIMENTS=1 build; none of them lives in csrc/"""
    made = _generate(tmp_path, "experiments")
    assert len(made) >= 21 and all("_v" in f for f in made), made


def test_every_loop_keeps_its_hazard_rules(tmp_path):
    """static checks of the generated text: an LDS-DMA never follows its M0 write without an instruction in between, every
    barrier is preceded by the waits that make it meaningful, and the X3 loops never use an instruction offset on a DMA"""
    for f in _generate(tmp_path, "product"):
        if not f.startswith("gemm_asm"):
            continue
        lines = [ln.strip() for ln in open(os.path.join(tmp_path, f)) if ln.strip().startswith(('"', "MF"))]
        ins = [ln for ln in lines if not ln.startswith('"; ')]
        for i, ln in enumerate(ins):
            if "buffer_load_dword" in ln and " lds" in ln:
                assert "m0" not in ins[i - 1], (f, ins[i - 1], ln)          # one wait state between the M0 write and its use
                assert "offset:" not in ln or "x3" not in f, (f, ln)        # an instruction offset moves the LDS address too
            if ln.startswith('"s_barrier') and i > 0 and "k64" in f or (ln.startswith('"s_barrier') and "x3" in f):
                prev = ins[i - 1]
                assert "s_waitcnt" in prev or "v_mfma" in prev or prev.startswith("MF"), (f, prev)
