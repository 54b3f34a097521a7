"""MXFP8 GEMM micro-benchmark on the GPU box: FFN shapes x tile configs (development aid)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sylber_amd import _lib
lib = _lib.load()
M = 16000
SHAPES = [("ffn1", M, 3072, 768, 0, 1), ("ffn2", M, 768, 3072, 6, 0), ("qkv", M, 2304, 768, 1, 0), ("sq4096", 4096, 4096, 4096, 1, 0)]
cfgs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 3]
for name, m, n, k, epi, act in SHAPES:
    row = []
    for cfg in cfgs:
        ms = ctypes.c_float()
        _lib.check(lib.sylber_debug_gemm_bench(m, n, k, k, epi, act, 100 + cfg, 20, ctypes.byref(ms)), "gemm_bench")
        row.append("cfg%2d %7.1f us %6.0f TF" % (cfg, ms.value * 1e3, 2.0 * m * n * k / (ms.value * 1e-3) / 1e12))
    print("%-7s M=%-7d N=%-5d K=%-5d | " % (name, m, n, k) + " | ".join(row), flush=True)
