"""CPU tier: sylber_amd/csrc/powf_half.h (the device restatement of glibc powf(x, .5f) that the
segmentation kernel needs for numpy's scalar `** .5`) against the host libm."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_replica_matches_libm(tmp_path):
    exe = str(tmp_path / "powf_check")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp",
                           os.path.join(ROOT, "tests", "powf_replica_check.c"), "-o", exe, "-lm"])
    stride = "1" if os.environ.get("SYLBER_EXHAUSTIVE") == "1" else "61"
    out = subprocess.run([exe, stride], capture_output=True, text=True)
    n, bad = (int(x) for x in out.stdout.split())
    assert out.returncode == 0 and bad == 0 and n > 3e7, out.stdout
