"""The inline-asm K loops in sylber_amd/csrc/gemm_asm*.inc are GENERATED (tools/gen_gemm_asm.py): the committed files must be
exactly what the committed generator writes, so that the schedule documented in the generator is the one that ships."""
import filecmp
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_loops_are_current(tmp_path):
    env = dict(os.environ, GEN_GEMM_ASM_OUT=str(tmp_path))
    gen = os.path.join(ROOT, "tools", "gen_gemm_asm.py")
    subprocess.run([sys.executable, gen, "product"], check=True, env=env, capture_output=True)
    made = sorted(os.listdir(tmp_path))
    assert len(made) >= 7 and "gemm_asm_x3_w8.inc" in made and "gemm_asm_k64.inc" in made and "gemm_asm_loop.inc" in made
    csrc = os.path.join(ROOT, "sylber_amd", "csrc")
    shipped = sorted(f for f in os.listdir(csrc) if f.startswith("gemm_asm") and f.endswith(".inc"))
    assert shipped == made, (set(shipped) ^ set(made))
    for f in made:
        assert filecmp.cmp(os.path.join(tmp_path, f), os.path.join(csrc, f), shallow=False), f


def test_experiment_loops_are_generated_not_committed(tmp_path):
    """the knock-out / schedule variants (timing only, results wrong by construction) are generated into the build directory by a
    SYLBER_EXPERIMENTS=1 build; none of them lives in csrc/"""
    env = dict(os.environ, GEN_GEMM_ASM_OUT=str(tmp_path))
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_gemm_asm.py"), "experiments"], check=True, env=env, capture_output=True)
    made = sorted(os.listdir(tmp_path))
    assert len(made) >= 21 and all("_v" in f for f in made), made
    csrc = os.path.join(ROOT, "sylber_amd", "csrc")
    assert not [f for f in os.listdir(csrc) if f.startswith("gemm_asm") and "_v" in f]


def test_every_loop_keeps_its_hazard_rules():
    """static checks of the generated text: an LDS-DMA never follows its M0 write without an instruction in between, every
    barrier is preceded by the waits that make it meaningful, and the X3 loops never use an instruction offset on a DMA"""
    csrc = os.path.join(ROOT, "sylber_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if not (f.startswith("gemm_asm") and f.endswith(".inc")):
            continue
        lines = [ln.strip() for ln in open(os.path.join(csrc, f)) if ln.strip().startswith(('"', "MF"))]
        ins = [ln for ln in lines if not ln.startswith('"; ')]
        for i, ln in enumerate(ins):
            if "buffer_load_dword" in ln and " lds" in ln:
                assert "m0" not in ins[i - 1], (f, ins[i - 1], ln)          # one wait state between the M0 write and its use
                assert "offset:" not in ln or "x3" not in f, (f, ln)        # an instruction offset moves the LDS address too
            if ln.startswith('"s_barrier') and i > 0 and "k64" in f or (ln.startswith('"s_barrier') and "x3" in f):
                prev = ins[i - 1]
                assert "s_waitcnt" in prev or "v_mfma" in prev or prev.startswith("MF"), (f, prev)
