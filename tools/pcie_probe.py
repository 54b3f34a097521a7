"""PCIe probe on the GPU box: D2H / H2D of the 49 MB of hidden states of a 32 x 10 s batch, as one copy and as row chunks on
several streams (one SDMA engine per stream), pinned host memory."""
import time, torch
n = 32 * 499 * 768
d = torch.randn(n, device="cuda"); h = torch.empty(n, dtype=torch.float32, pin_memory=True)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
for chunks in (1, 2, 4, 8):
    ss = [torch.cuda.Stream() for _ in range(chunks)]
    b = [(i * n // chunks, (i + 1) * n // chunks) for i in range(chunks)]
    def d2h():
        for s, (lo, hi) in zip(ss, b):
            with torch.cuda.stream(s): h[lo:hi].copy_(d[lo:hi], non_blocking=True)
    def h2d():
        for s, (lo, hi) in zip(ss, b):
            with torch.cuda.stream(s): d[lo:hi].copy_(h[lo:hi], non_blocking=True)
    a, c = t(d2h), t(h2d)
    print("%d stream(s): D2H %.2f ms = %.1f GB/s   H2D %.2f ms = %.1f GB/s" % (chunks, a * 1e3, n * 4 / a / 1e9, c * 1e3, n * 4 / c / 1e9))
def both():
    with torch.cuda.stream(ss[0]): h[: n // 2].copy_(d[: n // 2], non_blocking=True)
    with torch.cuda.stream(ss[1]): d[n // 2:].copy_(h[n // 2:], non_blocking=True)
a = t(both)
print("D2H of one half + H2D of the other, concurrently: %.2f ms" % (a * 1e3))
