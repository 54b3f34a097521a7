import ctypes, os, sys
sys.path.insert(0, os.getcwd())
from sylber_amd import _lib
lib = _lib.load()
for name, m, n, k, ldx in [("ffn1", 16384, 3072, 768, 768), ("conv3", 131072, 512, 1536, 1024)]:
    res = {}
    for label, cfg, act in [("full", 97, 1), ("no gelu", 97, 0), ("no stores", 89, 1), ("no epilogue", 88, 1)]:
        ms = ctypes.c_float()
        _lib.check(lib.sylber_debug_gemm_bench(m, n, k, ldx, 0, act, cfg, 20, ctypes.byref(ms)), "gemm_bench")
        res[label] = ms.value * 1e3
    tiles = (m // 256) * (n // 256) / 256
    print(name, " ".join("%s %.1f us" % kv for kv in res.items()), "| per tile: epilogue %.2f us, of it stores %.2f, gelu %.2f" % (
        (res["full"] - res["no epilogue"]) / tiles, (res["full"] - res["no stores"]) / tiles, (res["full"] - res["no gelu"]) / tiles))
