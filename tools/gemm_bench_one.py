"""one shape x one cfg, few iterations (for PMC collection)"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sylber_amd import _lib
lib = _lib.load()
m, n, k, ldx, epi, act, cfg = [int(x) for x in sys.argv[1:8]]
ms = ctypes.c_float()
_lib.check(lib.sylber_debug_gemm_bench(m, n, k, ldx, epi, act, cfg, 5, ctypes.byref(ms)), "gemm_bench")
print("%.1f us %.0f TF" % (ms.value * 1e3, 2.0 * m * n * k / (ms.value * 1e-3) / 1e12))
