#!/bin/bash
# same-box A/B of sylber_set_option values inside the forward (runs ON THE GPU BOX).
#   usage: VARIANTS="6=-1 6=1 6=2,7=-1" bash tools/ab_opt.sh [bench.py arguments]   (a variant = comma-separated key=value pairs)
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_forward']; r=d['roofline']
print('%8.1f audio-s/s  %6.3f ms/step (median %.3f)  out %.4f  ffn2 %.4f  layernorm %.4f  sum %.3f  enc %.4f' % (d['value'], d['ms_per_step'], d['ms_per_step_median'], k.get('gemm_out', 0), k.get('gemm_ffn2', 0), k.get('layernorm', 0), sum(k.values()), r['encoder_gemms']['frac']))"; }
for rep in 1 2 3; do
  for v in $VARIANTS; do
    printf "%-12s " "$v"; python bench.py --no-cpu-baseline --no-api $(echo $v | tr ',' '\n' | sed 's/^/--opt /' | tr '\n' ' ') "$@" 2>/dev/null | line
  done
done
