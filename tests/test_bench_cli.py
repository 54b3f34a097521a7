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


def test_roofline_by_peak_scores_fp8_launches_against_the_fp8_peak():
    """the MXFP8 launches answer to 5 PF, the 16-bit conv stack to 2.5 PF."""
    sys.path.insert(0, ROOT)
    import bench
    fl = bench.gemm_flops_per_forward(32)
    # every launch at exactly 1000 TFLOP/s
    kernels = {k: fl[k] / 1000e12 * 1e3 for k in fl}
    r8 = bench.roofline_by_peak(kernels, 32, bench.CLIP_SAMPLES, "fp8")
    assert r8["encoder_gemms"]["peak"] == 5000.0 and abs(r8["encoder_gemms"]["frac"] - 0.2) < 1e-3
    assert r8["conv_stack_and_projection"]["peak"] == 2500.0 and abs(r8["conv_stack_and_projection"]["frac"] - 0.4) < 1e-3
    r16 = bench.roofline_by_peak(kernels, 32, bench.CLIP_SAMPLES, "bf16")
    assert r16["encoder_gemms"]["peak"] == 2500.0 and abs(r16["encoder_gemms"]["frac"] - 0.4) < 1e-3
    rs = bench.roofline_by_peak(kernels, 32, bench.CLIP_SAMPLES, "split16")
    assert abs(rs["encoder_gemms"]["issued_frac"] - 1.2) < 1e-3        # three MFMA passes per algorithmic contraction


def test_attention_roofline_accounting():
    sys.path.insert(0, ROOT)
    import bench
    r = bench.attention_roofline({"attention": 0.333}, 32, 499, "bf16")
    assert abs(r["achieved"] - 9 * 4.0 * 32 * 12 * 499 * 499 * 64 / 0.333e-3 / 1e12) < 0.1 and r["peak"] == 2500.0
    assert bench.attention_roofline({"attention": 0.38}, 32, 499, "fp8")["peak"] == 5000.0
    assert bench.attention_roofline({}, 32, 499, "bf16") is None


def test_exchange_rehearsal_child_failure_is_reported_not_raised():
    """the one-rank RCCL self-test of `other_configs` runs as a CHILD process so that a communicator that fails or hangs cannot take the
    headline line with it: without a GPU the child dies at once, and the parent gets an `error` entry (not an exception, not a fake figure)"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: the child would run the real self-test")
    sys.path.insert(0, ROOT)
    import bench
    r = bench.exchange_rehearsal(timeout_s=300)
    assert set(r) == {"error"} and "self-test child" in r["error"]
