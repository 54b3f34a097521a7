#!/bin/bash
# same-box A/B of the first-round stagger of the 8-wave GEMM (development aid): gemm_bench shapes + the whole step
for d in 0 2000 4000 6000 9000; do
  echo "== SYLBER_GEMM_STAGGER=$d"
  SYLBER_GEMM_STAGGER=$d python tools/gemm_bench.py 10 2>&1 | grep -E "conv1|conv3|ffn1|sq4096"
  SYLBER_GEMM_STAGGER=$d python tools/gemm_bench.py 11 2>&1 | grep -E "qkv"
  SYLBER_GEMM_STAGGER=$d python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-api | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('   step', d['value'], d['ms_per_step'], {k:v for k,v in d['kernel_ms_per_forward'].items() if k.startswith('gemm_')})"
done
