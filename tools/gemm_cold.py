"""hot (operands cache-resident from the previous iteration) vs cold (1 GiB flushed through the caches between launches)
GEMM timings per tile config: the in-forward launches see cold activations (tools/gemm_bench.py measures hot)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sylber_amd import _lib
lib = _lib.load()
SH = {"ffn2": (16384, 768, 3072, 3072, 6, 0), "ffn1": (16384, 3072, 768, 768, 0, 1), "out": (16384, 768, 768, 768, 6, 0),
      "qkv": (16384, 2304, 768, 768, 3, 0), "conv3": (131072, 512, 1536, 1024, 0, 1)}
want = sys.argv[1].split(",") if len(sys.argv) > 1 else list(SH)
cfgs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 10, 80, 90, 95]
for name in want:
    m, n, k, ldx, epi, act = SH[name]
    for cfg in cfgs:
        out = []
        for cold in (0, 200000):
            ms = ctypes.c_float()
            rc = lib.sylber_debug_gemm_bench(m, n, k, ldx, epi, act, cfg + cold, 10, ctypes.byref(ms))
            out.append(ms.value * 1e3 if rc == 0 else float("nan"))
        print("%-6s cfg%-3d hot %7.1f us %6.0f TF | cold %7.1f us %6.0f TF" % (name, cfg, out[0], 2.0 * m * n * k / out[0] / 1e6, out[1], 2.0 * m * n * k / out[1] / 1e6), flush=True)
