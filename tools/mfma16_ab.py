"""Tile 97 (v_mfma_f32_32x32x16) against tile 47 (the same tile geometry on v_mfma_f32_16x16x32), same box, alternating (development aid).

The shapes are the launches tile 97 serves in the forward (conv1-5 in the 3-tap K order, FFN1) plus the 4096^3 yardstick; operands random, hot.
Output: one markdown table row per shape (profiles/r06_mfma16_loop.md)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sylber_amd import _lib

lib = _lib.load()
B, Tp = 32, 512
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
SHAPES = [  # name, M, N, K, ldx, epi, act, kpat
    ("conv1 (3-tap order)", B * Tp * 32, 512, 1536, 1024, 0, 1, True),
    ("conv2 (3-tap order)", B * Tp * 16, 512, 1536, 1024, 0, 1, True),
    ("conv4 (3-tap order)", B * Tp * 4, 512, 1536, 1024, 0, 1, True),
    ("conv5 (2-tap)", B * Tp * 2, 512, 1024, 1024, 0, 1, False),
    ("FFN1 32x10s", B * Tp, 3072, 768, 768, 0, 1, False),
    ("FFN1 8x60s", 8 * 3008, 3072, 768, 768, 0, 1, False),
    ("4096^3 plain", 4096, 4096, 4096, 4096, 0, 0, False),
    ("8192x8192x4096 plain", 8192, 8192, 4096, 4096, 0, 0, False),
]


def run(m, n, k, ldx, epi, act, cfg):
    ms = ctypes.c_float()
    _lib.check(lib.sylber_debug_gemm_bench(m, n, k, ldx, epi, act, cfg, 20, ctypes.byref(ms)), "gemm_bench")
    return ms.value * 1e3


print("| launch | M x N x K | tile 97 us (32x32x16) | tile 47 us (16x16x32) | TF 97 | TF 47 | 47 vs 97 |")
print("|---|---|---:|---:|---:|---:|---:|")
for name, m, n, k, ldx, epi, act, kpat in SHAPES:
    t = {97: [], 47: []}
    for _ in range(REPS):
        for tile in (97, 47):
            t[tile].append(run(m, n, k, ldx, epi, act, tile + (400000 if kpat else 0)))
    a, b = sorted(t[97])[len(t[97]) // 2], sorted(t[47])[len(t[47]) // 2]
    fl = 2.0 * m * n * k
    print("| %s | %d x %d x %d | %.1f (%s) | %.1f (%s) | %.0f | %.0f | %+.1f %% |" % (
        name, m, n, k, a, " ".join("%.1f" % v for v in t[97]), b, " ".join("%.1f" % v for v in t[47]), fl / a / 1e6, fl / b / 1e6, (a / b - 1) * 100), flush=True)
