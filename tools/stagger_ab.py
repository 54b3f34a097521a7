import ctypes, os, sys
sys.path.insert(0, os.getcwd())
from sylber_amd import _lib
lib = _lib.load()
for name, m, n, k, ldx in [("conv1", 524288, 512, 1536, 1024), ("conv2", 262144, 512, 1536, 1024), ("conv3", 131072, 512, 1536, 1024), ("ffn1", 16384, 3072, 768, 768)]:
    out = []
    for cfg in (97, 87, 97, 87):
        ms = ctypes.c_float()
        _lib.check(lib.sylber_debug_gemm_bench(m, n, k, ldx, 0, 1, cfg, 20, ctypes.byref(ms)), "gemm_bench")
        out.append("%d: %.1f us" % (cfg, ms.value * 1e3))
    print(name, " | ".join(out))
