"""Yardstick only, never product: what the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reaches on the GEMM shapes of
the hot path, on the same box, same random operand distribution as tools/gemm_bench.py (uniform [-0.5, 0.5) -- zero or
small-integer fills clock 15-20 % higher, MI355X_MICROARCH.md DVFS note).  VERDICT r2 item 1(b): turns "structural
ceiling" into a number.  The conv shapes run as plain [M, K] x [K, N] (same FLOPs; the library has no overlapping-row A).

    python tools/blaslt_yardstick.py            (on the GPU box)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

B, Tp = 32, 512
M = B * Tp
SHAPES = [("conv1", B * Tp * 32, 512, 1536), ("conv3", B * Tp * 8, 512, 1536), ("conv6", B * Tp, 512, 1024),
          ("qkv", M, 2304, 768), ("out", M, 768, 768), ("ffn1", M, 3072, 768), ("ffn2", M, 768, 3072),
          ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192)]


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda")
    print("| shape | M | N | K | bf16 us | bf16 TF | bf16+bias us | TF | fp32 us | fp32 TF |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for name, m, n, k in SHAPES:
        g = torch.Generator(device=dev); g.manual_seed(1)
        x = (torch.rand(m, k, device=dev, generator=g) - 0.5)
        w = (torch.rand(n, k, device=dev, generator=g) - 0.5)
        bias = torch.rand(n, device=dev, generator=g)
        xb, wb, bb = x.bfloat16(), w.bfloat16(), bias.bfloat16()
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        fl = 2.0 * m * n * k
        t1 = bench(lambda: torch.matmul(xb, wb.t(), out=out))
        t2 = bench(lambda: torch.nn.functional.linear(xb, wb, bb))
        if m * n <= 16384 * 3072 * 2 and name != "sq8192":
            o32 = torch.empty(m, n, device=dev)
            t3 = bench(lambda: torch.matmul(x, w.t(), out=o32), iters=5)
            f3 = "%.1f | %.0f" % (t3 * 1e3, fl / (t3 * 1e-3) / 1e12)
        else:
            f3 = "- | -"
        print("| %s | %d | %d | %d | %.1f | %.0f | %.1f | %.0f | %s |" % (
            name, m, n, k, t1 * 1e3, fl / (t1 * 1e-3) / 1e12, t2 * 1e3, fl / (t2 * 1e-3) / 1e12, f3), flush=True)


if __name__ == "__main__":
    main()
