"""MXFP8 GEMM micro-benchmark on the GPU box: round 1's kernels (cfg 102 / 103 = 8-wave 256x256 / 256x192) against the hand-
scheduled X3 loop (cfg 185 / 191 = tiles 85 / 91 of csrc/gemm_asm_f8.hip) on the encoder shapes (development aid)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sylber_amd import _lib
lib = _lib.load()
M = 16384
SHAPES = [("qkv", M, 2304, 768, 3, 0), ("out", M, 768, 768, 6, 0), ("ffn1", M, 3072, 768, 0, 1), ("ffn2", M, 768, 3072, 6, 0), ("sq4096", 4096, 4096, 4096, 1, 0),
          ("sq8192", 8192, 8192, 8192, 1, 0)]
for name, m, n, k, epi, act in SHAPES:
    row = []
    for cfg in (102, 103, 185, 192, 193):
        ms = ctypes.c_float()
        rc = lib.sylber_debug_gemm_bench(m, n, k, k, epi, act, cfg, 20, ctypes.byref(ms))
        row.append("cfg %d: %s" % (cfg - 100, "%7.1f us %6.0f TF" % (ms.value * 1e3, 2.0 * m * n * k / (ms.value * 1e-3) / 1e12) if rc == 0 else "n/a (%s)" % lib.sylber_last_error().decode()[:40]))
    print("%-7s M=%-6d N=%-5d K=%-5d | " % (name, m, n, k) + " | ".join(row), flush=True)
