#!/bin/bash
# same-box A/B of the residual GEMMs (runs ON THE GPU BOX): tile 96 (no residual prefetch) against tile 91 (K loop prefetches the
# residual rows), isolated hot / cold (tools/gemm_bench.py shapes "out" and "ffn2") and inside the forward
python - <<'PY'
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
from sylber_amd import _lib
lib = _lib.load()
for rep in range(2):
    for cold in (0, 200000):
        for name, m, n, k in [("out-proj", 16384, 768, 768), ("ffn2", 16384, 768, 3072)]:
            row = []
            for cfg in (96, 91, 4):
                ms = ctypes.c_float()
                _lib.check(lib.sylber_debug_gemm_bench(m, n, k, k, 6, 0, cfg + cold, 20, ctypes.byref(ms)), "gemm_bench")
                row.append("tile %d %.1f us" % (cfg, ms.value * 1e3))
            print("%-8s %-4s " % (name, "cold" if cold else "hot") + "  ".join(row), flush=True)
PY
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_forward']; r=d['roofline']; print('%8.1f audio-s/s  %6.3f ms/step  gemm_out %.4f  gemm_ffn2 %.4f ms  encoder gemms %.4f' % (d['value'], d['ms_per_step'], k['gemm_out'], k['gemm_ffn2'], r['encoder_gemms']['frac']))"; }
for rep in 1 2 3; do
  echo -n "ref lib            "; python tools/with_lib.py ref bench.py --no-cpu-baseline --no-api "$@" 2>/dev/null | line
  echo -n "new                "; python bench.py --no-cpu-baseline --no-api "$@" 2>/dev/null | line
done
