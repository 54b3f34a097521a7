"""Phase proportions of the persistent asm GEMM kernel (tile 97) from s_memtime stamps of one wave (SYLBER_EXPERIMENTS=1 build)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sylber_amd import _lib
lib = _lib.load()
for name, m, n, k, ldx in [("ffn1", 16384, 3072, 768, 768), ("conv3", 131072, 512, 1536, 1024), ("conv1", 524288, 512, 1536, 1024)]:
    out = (ctypes.c_uint64 * 20)(); ms = ctypes.c_float()
    _lib.check(lib.sylber_debug_gemm_trace(m, n, k, ldx, 0, 200, out, ctypes.byref(ms)), "gemm_trace")
    t = [int(x) for x in out[:6]]; tot = sum(t[:4]) or 1
    print("%-6s tiles %d  K loop %.1f %%  bias+prefetch issue %.1f %%  epilogue %.1f %%  seam %.1f %%   (ticks per tile %.0f; launch %.1f us; first epilogue block %.1f %% of the epilogue)" % (
        name, t[4], 100 * t[0] / tot, 100 * t[1] / tot, 100 * t[2] / tot, 100 * t[3] / tot, tot / max(t[4], 1), ms.value * 1e3, 100 * t[5] / max(t[2], 1)))
