"""Same GEMM launch with and without the GELU epilogue (what the activation costs; development aid)."""
import ctypes, sys, os
sys.path.insert(0, os.getcwd())
from sylber_amd import _lib
lib = _lib.load()
B, Tp = 32, 512
M = B * Tp
S = [("conv1", M * 32, 512, 1536, 1024), ("conv3", M * 8, 512, 1536, 1024), ("ffn1", M, 3072, 768, 768), ("ffn1_K3072", M, 3072, 3072, 3072)]
for name, m, n, k, ldx in S:
    row = []
    for act in (1, 0):
        ms = ctypes.c_float()
        _lib.check(lib.sylber_debug_gemm_bench(m, n, k, ldx, 0, act, 10, 20, ctypes.byref(ms)), "gemm_bench")
        row.append("act%d %7.1f us %5.0f TF" % (act, ms.value * 1e3, 2.0 * m * n * k / (ms.value * 1e-3) / 1e12))
    print("%-11s" % name, " | ".join(row), flush=True)
