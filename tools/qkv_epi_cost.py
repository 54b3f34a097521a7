"""what the fused q/k/v epilogue (head-major q, k + LDS-transposed V^T) costs against a plain 16-bit epilogue on the same tile (GPU box)"""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
from sylber_amd import _lib
lib = _lib.load()
for rep in range(2):
    row = []
    for label, epi in (("EPI_QK", 3), ("EPI_BF16", 0)):
        for cfg in (91, 4):
            ms = ctypes.c_float()
            _lib.check(lib.sylber_debug_gemm_bench(16384, 2304, 768, 768, epi, 0, cfg, 20, ctypes.byref(ms)), "x")
            row.append("%s tile %d: %.1f us" % (label, cfg, ms.value * 1e3))
    print(" | ".join(row), flush=True)
