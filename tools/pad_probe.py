"""Host padding of a 32 x 10 s batch into the pinned staging buffer (Segmenter.encode_batch): numpy slice assignment vs ctypes.memmove, by thread count.
Run on the GPU box (the staging buffer is page-locked there): which one scales with threads, i.e. which one really releases the GIL."""
import ctypes
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B, N = 32, 160000
rows = [torch.randn(N) for _ in range(B)]
stage = torch.empty(B, N, pin_memory=torch.cuda.is_available())
sn = stage.numpy()
base = stage.data_ptr()


def fill_np(lo, hi):
    for i in range(lo, hi):
        sn[i, :N] = rows[i].detach().numpy()


def fill_mm(lo, hi):
    for i in range(lo, hi):
        ctypes.memmove(base + i * N * 4, rows[i].data_ptr(), N * 4)


print("cpus", os.cpu_count(), "pinned", stage.is_pinned())
for name, fn in (("numpy", fill_np), ("memmove", fill_mm)):
    for nt in (1, 2, 4, 8, 16):
        pool = ThreadPoolExecutor(max_workers=nt)
        ng = 2 * nt if nt > 1 else 1
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            b = [(g * B // ng, (g + 1) * B // ng) for g in range(ng)]
            fs = [pool.submit(fn, lo, hi) for lo, hi in b]
            for f in fs:
                f.result()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print("%-8s threads %2d groups %2d: median %.3f ms  min %.3f ms" % (name, nt, ng, 1e3 * ts[len(ts) // 2], 1e3 * ts[0]))
        pool.shutdown()
