"""bench.py's launcher contract on a box without GPUs (CPU tier): it never oversubscribes, never falls back to the CPU,
and its FLOP accounting matches SURVEY.md §8(d)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*args):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=600, env=env)


def test_gpus_n_refuses_when_fewer_gpus_are_visible():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs visible: the refusal path does not apply")
    r = run("--gpus", "2")
    assert r.returncode == 2
    assert "refusing to oversubscribe" in r.stderr
    assert r.stdout.strip() == ""                       # no JSON line that could be mistaken for a measurement


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    r = run("--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "no CPU fallback" in r.stderr
    assert "{" not in r.stdout


def test_flop_accounting():
    sys.path.insert(0, ROOT)
    import bench
    f = bench.gemm_flops_per_forward(32)
    assert abs(sum(f.values()) - 3.6069e12) < 1e9        # 112.7 GFLOP per 10 s clip x 32
    # one 10 s clip: 499 frames; conv1 is 2 * L1 * 512 * 512 * 3
    f1 = bench.gemm_flops_per_forward(1)
    assert f1["gemm_conv1"] == 2.0 * 15999 * 512 * 512 * 3
    assert f1["gemm_ffn1"] == 9 * 2.0 * 499 * 768 * 3072
    assert len(bench.csrc_sha16()) == 16
