"""The 32x32x16 kernels against their siblings of the v_mfma_f32_16x16x32 family (csrc/gemm_asm16.hip), same box, alternating (development aid):
tile 97 vs 47 (256x256, eight waves, generated loops), 57 vs 46 (192x256), 3 vs 13 and 4 vs 14 (hipcc-scheduled 128x128 / 128x192, two per CU).
cfg + 1000000 = the launch on the 32x32x16 kernels (GemmArgs::tune_mfma16 = -1): without it a 16-bit-output launch maps any forced id into the family.

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
PAIRS = [(97, 47)]
SMALL = [  # the small tiles on the launches of small batches, and the 192-row siblings where a 256-row tile count leaves a partial round
    ("FFN1 1x10s", 512, 3072, 768, 768, 0, 1, False, (3, 13)),
    ("FFN1 1x10s", 512, 3072, 768, 768, 0, 1, False, (4, 14)),
    ("FFN1 4x10s", 2048, 3072, 768, 768, 0, 1, False, (3, 13)),
    ("FFN1 4x10s", 2048, 3072, 768, 768, 0, 1, False, (4, 14)),
    ("conv6 32x10s", 16384, 512, 1024, 1024, 0, 1, False, (3, 13)),
    ("conv4 1x10s (3-tap)", 2048, 512, 1536, 1024, 0, 1, True, (4, 14)),
    ("FFN1 24x10s", 24 * 512, 3072, 768, 768, 0, 1, False, (57, 46)),
    ("conv5 24x10s", 24 * 1024, 512, 1024, 1024, 0, 1, False, (57, 46)),
    # inside the family: the four-wave small tiles against their eight-wave forms (same tile, same bits)
    ("FFN1 1x10s", 512, 3072, 768, 768, 0, 1, False, (13, 15)),
    ("FFN1 1x10s", 512, 3072, 768, 768, 0, 1, False, (14, 16)),
    ("FFN1 2x10s", 1024, 3072, 768, 768, 0, 1, False, (13, 15)),
    ("FFN1 4x10s", 2048, 3072, 768, 768, 0, 1, False, (13, 15)),
    ("FFN1 4x10s", 2048, 3072, 768, 768, 0, 1, False, (14, 16)),
    ("FFN1 8x10s", 4096, 3072, 768, 768, 0, 1, False, (13, 15)),
    ("conv6 32x10s", 16384, 512, 1024, 1024, 0, 1, False, (13, 15)),
    ("conv4 1x10s (3-tap)", 2048, 512, 1536, 1024, 0, 1, True, (13, 15)),
    ("conv3 1x10s (3-tap)", 4096, 512, 1536, 1024, 0, 1, True, (13, 15)),
    ("conv5 1x10s", 1024, 512, 1024, 1024, 0, 1, False, (13, 15)),
]


def run(m, n, k, ldx, epi, act, cfg):
    ms = ctypes.c_float()
    _lib.check(lib.sylber_debug_gemm_bench(m, n, k, ldx, epi, act, cfg, 20, ctypes.byref(ms)), "gemm_bench")
    return ms.value * 1e3


print("| launch | M x N x K | tiles | first us | second us | TF | TF | second vs first |")
print("|---|---|---|---:|---:|---:|---:|---:|")
for row in [r + ((97, 47),) for r in SHAPES] + SMALL:
    name, m, n, k, ldx, epi, act, kpat, (old, new) = row
    t = {old: [], new: []}
    for _ in range(REPS):
        for tile in (old, new):
            t[tile].append(run(m, n, k, ldx, epi, act, tile + (400000 if kpat else 0) + (1000000 if (tile == old and old not in (13, 14)) else 0)))
    a, b = sorted(t[old])[len(t[old]) // 2], sorted(t[new])[len(t[new]) // 2]
    fl = 2.0 * m * n * k
    print("| %s | %d x %d x %d | %d vs %d | %.1f (%s) | %.1f (%s) | %.0f | %.0f | %+.1f %% |" % (
        name, m, n, k, old, new, a, " ".join("%.1f" % v for v in t[old]), b, " ".join("%.1f" % v for v in t[new]), fl / a / 1e6, fl / b / 1e6, (a / b - 1) * 100), flush=True)
