"""kernel-only timing of the attention core (development aid): the hand-scheduled key loop (default) against the compiler-scheduled
kernels (32 / 64 queries per wave), HIP events around `iters` launches on packed random operands"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sylber_amd import _lib
lib = _lib.load()
lib.sylber_debug_attention_bench.argtypes = [ctypes.c_int32] * 4 + [ctypes.POINTER(ctypes.c_float)]
torch.zeros(1, device="cuda")
for (B, T) in [(32, 499), (8, 2999)]:
    fl = 4.0 * B * 12 * T * T * 64
    for name, prec in (("asm", 0), ("hipcc qw=1", 132), ("hipcc qw=2", 164), ("fp8", 2)):
        best = 1e9
        for _ in range(3):
            ms = ctypes.c_float()
            _lib.check(lib.sylber_debug_attention_bench(B, T, prec, 20, ctypes.byref(ms)), "bench")
            best = min(best, ms.value)
        print("B=%d T=%d %-11s %.1f us  %.0f TF" % (B, T, name, best * 1e3, fl / (best * 1e-3) / 1e12), flush=True)
    ms = ctypes.c_float()
    _lib.check(lib.sylber_debug_attention_bench(B, T, 0, -20, ctypes.byref(ms)), "bench")
    print("B=%d T=%d asm on ALL-ZERO operands (DVFS probe): %.1f us" % (B, T, ms.value * 1e3), flush=True)
if os.environ.get("SYLBER_DEV_LIB", "").endswith("_exp.so"):
    names = {1: "no exp", 2: "no softmax", 3: "no MFMA", 4: "no frag reads", 5: "no DMA/barrier", 6: "no max", 7: "MFMA + softmax only", 8: "MFMA only", 9: "data movement only"}
    for (B, T) in [(32, 499), (8, 2999)]:
        for var in sorted(names):
            best = 1e9
            for _ in range(3):
                ms = ctypes.c_float()
                _lib.check(lib.sylber_debug_attention_bench(B, T, 200 + var, 20, ctypes.byref(ms)), "bench")
                best = min(best, ms.value)
            print("B=%d T=%d knock-out %d (%s): %.1f us" % (B, T, var, names[var], best * 1e3), flush=True)
