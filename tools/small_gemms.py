import ctypes, os, sys
sys.path.insert(0, os.getcwd())
from sylber_amd import _lib
lib = _lib.load()
for name, m, n, k, ldx, epi, act in [("conv5", 32768, 512, 1024, 1024, 0, 1), ("conv6", 16384, 512, 1024, 1024, 0, 1), ("conv4", 65536, 512, 1536, 1024, 0, 1)]:
    row = []
    for cfg in (-1, 3, 4, 10, 85, 91, 97):
        ms = ctypes.c_float()
        rc = lib.sylber_debug_gemm_bench(m, n, k, ldx, epi, act, cfg, 20, ctypes.byref(ms))
        row.append("cfg%d %.1f us %.0f TF" % (cfg, ms.value * 1e3, 2.0 * m * n * k / (ms.value * 1e-3) / 1e12) if rc == 0 else "cfg%d -" % cfg)
    print(name, " | ".join(row), flush=True)
