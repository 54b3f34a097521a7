"""Tile 47 on the shapes it serves, one line (development aid): run once per build variant through tools/with_lib.py to compare schedules of the Y3 loop
(tools/gen_gemm_asm.py emit_y3: Y3_GB, Y3_DSTRIDE, Y3_DPOS, Y3_HPOS) on one box."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sylber_amd import _lib

lib = _lib.load()
SHAPES = [("conv1t", 524288, 512, 1536, 1024, 400047), ("conv4t", 65536, 512, 1536, 1024, 400047), ("conv5", 32768, 512, 1024, 1024, 47),
          ("ffn1", 16384, 3072, 768, 768, 47), ("sq4096", 4096, 4096, 4096, 4096, 47)]
out = []
for name, m, n, k, ldx, cfg in SHAPES:
    ts = []
    for _ in range(3):
        ms = ctypes.c_float()
        _lib.check(lib.sylber_debug_gemm_bench(m, n, k, ldx, 0, 1 if name != "sq4096" else 0, cfg, 20, ctypes.byref(ms)), "gemm_bench")
        ts.append(ms.value * 1e3)
    out.append("%s %.1f us %.0f TF" % (name, sorted(ts)[1], 2.0 * m * n * k / sorted(ts)[1] / 1e6))
print("%-8s | %s" % (os.environ.get("SYLBER_DEV_LIB", "product").split("_")[-1].replace(".so", ""), " | ".join(out)), flush=True)
