"""One GEMM shape, plain 16-bit epilogue, hot operands: the vendor library (torch.matmul -> hipBLASLt) and this library's kernel, 40 launches each, back to back in
one process -- the workload of the PMC pass that answers "same MFMA count, different clock?" (VERDICT r5 item 3):

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d out -- python tools/yardstick_pair.py sq4096
    python tools/pmc_mfma.py out/.../counter_collection.csv profiles/r06_yardstick_pmc_sq4096.md
Without rocprofv3 it prints the two event-timed figures (yardstick only, never product)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sylber_amd import _lib

SHAPES = {"sq4096": (4096, 4096, 4096), "ffn2": (16384, 768, 3072), "ffn1": (16384, 3072, 768), "conv3": (131072, 512, 1536)}
name = sys.argv[1] if len(sys.argv) > 1 else "sq4096"
m, n, k = SHAPES[name]
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(1)
x = (torch.rand(m, k, device=dev, generator=g) - 0.5).bfloat16()
w = (torch.rand(n, k, device=dev, generator=g) - 0.5).bfloat16()
out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


t_v = timed(lambda: torch.matmul(x, w.t(), out=out), 40)
lib = _lib.load()
ms, ms32 = ctypes.c_float(), ctypes.c_float()
# the automatic tile of a 16-bit-output launch is a member of the v_mfma_f32_16x16x32 family (tile 47 at these shapes); cfg 1000097 = tile 97, the
# same geometry on v_mfma_f32_32x32x16 (rounds 3-6a): the three kernels of profiles/r06_yardstick_pmc.md
_lib.check(lib.sylber_debug_gemm_bench(m, n, k, k, 0, 0, -1, 40, ctypes.byref(ms)), "gemm_bench")
_lib.check(lib.sylber_debug_gemm_bench(m, n, k, k, 0, 0, 1000097, 40, ctypes.byref(ms32)), "gemm_bench")
fl = 2.0 * m * n * k
print("%s M=%d N=%d K=%d | hipBLASLt %.1f us %.0f TF | this library (16x16x32 family) %.1f us %.0f TF | tile 97 (32x32x16) %.1f us %.0f TF | vendor / ours %.3f" % (
    name, m, n, k, t_v * 1e3, fl / t_v / 1e9, ms.value * 1e3, fl / ms.value / 1e9, ms32.value * 1e3, fl / ms32.value / 1e9, t_v / ms.value))
