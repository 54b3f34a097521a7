"""Tail policy of the persistent GEMM tiles (gemm_bf16.hip launch_f): us per launch of the encoder GEMM shapes over a range of row counts,
the round-5 launcher against the 192-row tiles and the row split -- the data the cost model's constants (GEMM_H192_EFF, GEMM_SOLO_EFF, GEMM_SPLIT_US) are fitted to.

    python tools/tail_sweep.py [row_tiles,...] > profiles/r06_tail_policy.md       (on the GPU box)
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sylber_amd import _lib

lib = _lib.load()
SHAPES = [  # name, N, K, ldx, epi, act
    ("qkv", 2304, 768, 768, 3, 0),
    ("out", 768, 768, 768, 6, 0),
    ("ffn1", 3072, 768, 768, 0, 1),
    ("ffn2", 768, 3072, 3072, 6, 0),
    ("conv5", 512, 1024, 1024, 0, 1),
]
row_tiles = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [24, 40, 48, 64, 72, 80, 94, 96, 112, 128, 160, 188]
BIG = {"qkv": (91, 51), "out": (91, 51), "ffn1": (97, 57), "ffn2": (91, 51), "conv5": (97, 57)}     # the shape's 256-row tile and its 192-row sibling


def run(m, n, k, ldx, epi, act, cfg=-1, tail_code=1, h192=True):
    ms = ctypes.c_float()
    rc = lib.sylber_debug_gemm_bench(m, n, k, ldx, epi, act + 100 * tail_code + (0 if h192 else 10000), cfg, 20, ctypes.byref(ms))
    return ms.value * 1e3 if rc == 0 else float("nan")


print("us per launch (hot operands, tools/tail_sweep.py).  `r5` = the round-5 launcher (256-row tiles only, one launch); `h192` = the cost model "
      "with the 192-row tiles (51 / 57), one launch; `auto` = as shipped (192-row tiles + row split where the model predicts > 5 %); "
      "`t256` / `t192` = the shape's hand-scheduled tile forced at 256 / 192 rows, one launch")
print()
print("| shape | rows / 256 | r5 | h192 | auto | t256 | t192 | TF r5 | TF auto | auto / r5 |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for name, n, k, ldx, epi, act in SHAPES:
    for rt in row_tiles:
        m = rt * 256 * (2 if name == "conv5" else 1)
        r5 = run(m, n, k, ldx, epi, act, h192=False)
        h = run(m, n, k, ldx, epi, act)
        auto = run(m, n, k, ldx, epi, act, tail_code=0)
        t256 = run(m, n, k, ldx, epi, act, cfg=BIG[name][0])
        t192 = run(m, n, k, ldx, epi, act, cfg=BIG[name][1])
        fl = 2.0 * m * n * k
        print("| %s | %d | %.1f | %.1f | %.1f | %.1f | %.1f | %.0f | %.0f | %.3f |" % (name, m // 256, r5, h, auto, t256, t192, fl / r5 / 1e6, fl / auto / 1e6, auto / r5), flush=True)
