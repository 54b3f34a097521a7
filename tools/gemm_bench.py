"""GEMM micro-benchmark on the GPU box: the hot-path shapes x tile configs (development aid)."""
import ctypes
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sylber_amd import _lib

lib = _lib.load()
B, Tp = 32, 512
M = B * Tp
SHAPES = [  # name, M, N, K, ldx, epi, act
    ("conv1", B * Tp * 32, 512, 1536, 1024, 0, 1),
    ("conv3", B * Tp * 8, 512, 1536, 1024, 0, 1),
    ("conv6", B * Tp, 512, 1024, 1024, 0, 1),
    ("qkv", M, 2304, 768, 768, 3, 0),
    ("out", M, 768, 768, 768, 6, 0),
    ("ffn1", M, 3072, 768, 768, 0, 1),
    ("ffn2", M, 768, 3072, 3072, 6, 0),
    ("sq4096", 4096, 4096, 4096, 4096, 0, 0),
]
cfgs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [-1, 3, 4, 10, 11]
for name, m, n, k, ldx, epi, act in SHAPES:
    row = []
    for cfg in cfgs:
        ms = ctypes.c_float()
        _lib.check(lib.sylber_debug_gemm_bench(m, n, k, ldx, epi, act, cfg, 20, ctypes.byref(ms)), "gemm_bench")
        row.append("cfg%2d %7.1f us %6.0f TF" % (cfg, ms.value * 1e3, 2.0 * m * n * k / (ms.value * 1e-3) / 1e12))
    print("%-7s M=%-7d N=%-5d K=%-5d | " % (name, m, n, k) + " | ".join(row), flush=True)
