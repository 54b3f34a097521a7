"""What the fp32-residual epilogue of out-proj / FFN2 costs on tile 91 (SYLBER_EXPERIMENTS=1 build; timing only): the full
kernel against knock-outs -- residual rows read from one cache-hot line (no HBM read of `pre`), no epilogue -- with
operands hot (back-to-back launches) and cold (1 GiB written between the launches)."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
from sylber_amd import _lib
lib = _lib.load()
for cold in (0, 200000):
    for name, m, n, k in [("out-proj", 16384, 768, 768), ("ffn2", 16384, 768, 3072)]:
        res = {}
        for label, cfg in [("full", 91), ("residual hot line", 92), ("no epilogue", 93)]:
            ms = ctypes.c_float()
            _lib.check(lib.sylber_debug_gemm_bench(m, n, k, k, 6, 0, cfg + cold, 20, ctypes.byref(ms)), "gemm_bench")
            res[label] = ms.value * 1e3
        print("%-8s %-4s " % (name, "cold" if cold else "hot") + "  ".join("%s %.1f us" % kv for kv in res.items()), flush=True)
