"""The two-per-CU small tiles on four waves against the same tiles on EIGHT waves, same box, alternating (development aid; profiles/r06_small_tiles.md):
32x32x16 kernels 3 vs 5 (128x128) and 4 vs 6 (128x192) on the encoder's launches (q,k,v / out-projection / FFN2; bit-identical outputs), and the
16x16x32 family's 13 vs 15, 14 vs 16 on FFN1, for batches of 1 ... 32 ten-second clips (512 frames per clip)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sylber_amd import _lib

lib = _lib.load()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ALT = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else None      # other id quadruple for the 32x32x16 rows, e.g. 3,7,4,8 (three-slot rings)
ALTF = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else None     # the same for the family's FFN1 rows
LAUNCHES = [("q,k,v", 2304, 768, 3, 0), ("out-proj", 768, 768, 6, 0), ("FFN2", 768, 3072, 6, 0), ("FFN1", 3072, 768, 0, 1)]


def run(m, n, k, epi, act, cfg):
    ms = ctypes.c_float()
    _lib.check(lib.sylber_debug_gemm_bench(m, n, k, k, epi, act, cfg, 20, ctypes.byref(ms)), "gemm_bench")
    return ms.value * 1e3


print("| launch | clips | M x N x K | 128x128: first us | second us | gain | 128x192: first us | second us | gain | best of the four |")
print("|---|---:|---|---:|---:|---:|---:|---:|---:|---|")
for name, n, k, epi, act in LAUNCHES:
    fam = epi == 0
    ids = (tuple(ALTF) if ALTF else (13, 15, 14, 16)) if fam else (tuple(ALT) if ALT else (3, 5, 4, 6))
    for clips in (1, 2, 4, 8, 16, 32):
        m = clips * 512
        t = {c: [] for c in ids}
        for _ in range(REPS):
            for c in ids:
                t[c].append(run(m, n, k, epi, act, c))
        med = {c: sorted(v)[len(v) // 2] for c, v in t.items()}
        best = min(med, key=med.get)
        print("| %s | %d | %d x %d x %d | %.1f | %.1f | %+.0f %% | %.1f | %.1f | %+.0f %% | %d (%.1f us) |" % (
            name, clips, m, n, k, med[ids[0]], med[ids[1]], (med[ids[0]] / med[ids[1]] - 1) * 100, med[ids[2]], med[ids[3]], (med[ids[2]] / med[ids[3]] - 1) * 100, best, med[best]), flush=True)
