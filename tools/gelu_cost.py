import ctypes, os, sys
sys.path.insert(0, os.getcwd())
from sylber_amd import _lib
lib = _lib.load()
for name, m, n, k, ldx in [("ffn1", 16384, 3072, 768, 768), ("conv3", 131072, 512, 1536, 1024), ("conv1", 524288, 512, 1536, 1024)]:
    for cfg in (95, 97):
        row = []
        for act in (0, 1):
            ms = ctypes.c_float()
            _lib.check(lib.sylber_debug_gemm_bench(m, n, k, ldx, 0, act, cfg, 20, ctypes.byref(ms)), "gemm_bench")
            row.append(ms.value * 1e3)
        tiles_per_cu = (m // 256) * (n // 256) / 256
        print("%-6s cfg%d  act0 %7.1f us  gelu %7.1f us  -> GELU costs %5.2f us per tile (%.1f tiles per CU)" % (name, cfg, row[0], row[1], (row[1] - row[0]) / tiles_per_cu, tiles_per_cu))
