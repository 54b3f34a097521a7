#!/bin/bash
# The 16-bit-output GEMM role on the v_mfma_f32_16x16x32 family (default) against the 32x32x16 kernels (--opt 13=-1), same box, alternating:
# the 32 x 10 s headline, 8 x 60 s, and the small batches the hipcc-scheduled small tiles serve (1 x 10 s, 4 x 10 s).
# usage: tools/mfma16_forward_ab.sh [reps]   -> one line per run: batch, seconds, option, ms/step, encoder / dominant fractions
REPS=${1:-2}
Q="--no-api --no-other-configs --no-exchange-rehearsal --no-cpu-baseline --agreement-clips 0"
for r in $(seq $REPS); do
  for opt in "" "--opt 13=-1"; do
    for shape in "32 10" "8 60" "1 10" "4 10"; do
      set -- $shape
      python bench.py --batch $1 --clip-seconds $2 --steps 20 --warmup 5 $Q $opt 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('B=%s x %ss  %-12s  %.3f ms/step  value %.0f  gemm_family %.4f  enc %.4f  per-launch %s' % ('$1', '$2', '${opt:-family}', d['ms_per_step'], d['value'], r['frac'], r.get('encoder_gemms', {}).get('frac', 0), {k: v for k, v in r['per_launch_tflops'].items() if k in ('gemm_conv1', 'gemm_conv4', 'gemm_conv5', 'gemm_conv6', 'gemm_ffn1')}))
"
    done
  done
done
