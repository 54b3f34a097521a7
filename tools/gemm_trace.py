"""Per-phase cycle account of the 8-wave 256x256 GEMM's K loop from s_memtime stamps of one wave per stagger group
(the thread-trace decoder is not in this image, so the kernel stamps itself; tile id 30 is a
separate instantiation, the shipping kernels carry no instrumentation).  Run on the GPU box."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sylber_amd import _lib

lib = _lib.load()
SHAPES = [("sq4096", 4096, 4096, 4096, 4096, 0, 0), ("ffn1", 16384, 3072, 768, 768, 0, 1), ("ffn2", 16384, 768, 3072, 3072, 6, 0),
          ("out", 16384, 768, 768, 768, 6, 0), ("conv3", 131072, 512, 1536, 1024, 0, 1)]
print("| shape | wave | steps | A: reads + DMA retire | barrier after A | B: 16 MFMA + 4 DMA | barrier after B | step total | stamp cost | "
      "prologue | K loop | epilogue | tile |")
print("|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for name, m, n, k, ldx, epi, act in SHAPES:
    out = (ctypes.c_uint64 * 20)()
    ms = ctypes.c_float()
    _lib.check(lib.sylber_debug_gemm_trace(m, n, k, ldx, epi, act, out, ctypes.byref(ms)), "gemm_trace")
    for g in range(2):
        o = out[g * 10:(g + 1) * 10]
        st = max(int(o[5]), 1)
        print("| %s | group %d | %d | %.0f | %.0f | %.0f | %.0f | %.0f | %d | %d | %d | %d | %d |" % (
            name, g, st, o[0] / st, o[1] / st, o[2] / st, o[3] / st, o[4] / st, o[6], o[8], o[4], o[7], o[9]), flush=True)
print("\ncycles per step and wave (shader clock, s_memtime); every segment contains one stamp (its cost in the 'stamp cost' column);")
print("16 v_mfma_f32_32x32x16_bf16 occupy the SIMD's matrix pipe for 512 cycles, and two waves share a SIMD: 1024 per step is the floor.")

print("\n### unstaggered 8-wave kernel (tile id 40), same stamps: [A] = kk0 group (8 MFMA + 6 reads + 2 DMA, until its fragments landed), "
      "[barrier after A] = vmcnt wait, [B] = s_barrier\n")
print("| shape | wave | steps | kk0 group | vmcnt wait | barrier | - | step total | stamp cost | prologue | K loop |")
print("|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for name, m, n, k, ldx, epi, act in SHAPES[:3]:
    out = (ctypes.c_uint64 * 20)()
    ms = ctypes.c_float()
    _lib.check(lib.sylber_debug_gemm_trace(m, n, k, ldx, 0, 100, out, ctypes.byref(ms)), "gemm_trace")
    for g in range(2):
        o = out[g * 10:(g + 1) * 10]
        st = max(int(o[5]), 1)
        print("| %s | wave %d | %d | %.0f | %.0f | %.0f | %.0f | %.0f | %d | %d | %d |" % (
            name, 4 * g, st, o[0] / st, o[1] / st, o[2] / st, o[3] / st, o[4] / st, o[6], o[8], o[4]), flush=True)
