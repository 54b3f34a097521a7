"""K sweep at fixed M x N: time = overhead + steps x per-step.  Separates the K loop's rate from a tile's fixed costs
(launch, prologue, epilogue) for our tile configs and for the vendor library (torch.matmul -> hipBLASLt) on the same box."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sylber_amd import _lib

lib = _lib.load()
cfgs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [10, 60]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
KS = [512, 1024, 2048, 4096, 8192]


def ours(cfg, k):
    ms = ctypes.c_float()
    _lib.check(lib.sylber_debug_gemm_bench(M, N, k, k, 0, 0, cfg, 30, ctypes.byref(ms)), "gemm_bench")
    return ms.value * 1e3


def vendor(k):
    a = torch.randn(M, k, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, k, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        torch.matmul(a, w.T)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        torch.matmul(a, w.T)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30 * 1e3


def fit(ts):
    # least squares of t = a + b * steps over the sweep
    xs = [k / 32 for k in KS]
    n = len(xs)
    mx, mt = sum(xs) / n, sum(ts) / n
    b = sum((x - mx) * (t - mt) for x, t in zip(xs, ts)) / sum((x - mx) ** 2 for x in xs)
    return mt - b * mx, b


print(f"M={M} N={N}  tiles of 256x256: {((M + 255) // 256) * ((N + 255) // 256)}")
rows = [("cfg%d" % c, [ours(c, k) for k in KS]) for c in cfgs] + [("hipBLASLt", [vendor(k) for k in KS])]
print("%-10s" % "K" + "".join("%10d" % k for k in KS) + "   overhead us   us / K=32 step   loop TF")
for name, ts in rows:
    a, b = fit(ts)
    print("%-10s" % name + "".join("%10.1f" % t for t in ts) + "   %8.1f   %12.3f   %8.0f" % (a, b, 2.0 * M * N * 32 / (b * 1e-6) / 1e12))
