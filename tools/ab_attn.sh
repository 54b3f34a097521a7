#!/bin/bash
# same-box A/B of the attention kernel variants inside the forward (runs ON THE GPU BOX): reference build (sylber_amd/libsylber_hip_ref.so)
# against the in-tree build with SYLBER_OPT_ATTN_QUERIES_PER_WAVE = the given codes (attention.hip launch_attention_any)
CODES=${CODES:-"32 64"}
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_forward']; print('%8.1f audio-s/s  %6.3f ms/step  attention %.4f ms  (sum of kernels %.3f)' % (d['value'], d['ms_per_step'], k['attention'], sum(k.values())))"; }
for rep in 1 2; do
  for args in "--no-overlap" "--no-overlap --batch 8 --clip-seconds 60"; do
    echo "== rep $rep  [$args]"
    echo -n "ref          "; python tools/with_lib.py ref bench.py --no-cpu-baseline --no-api $args 2>/dev/null | line
    for v in $CODES; do
      echo -n "new opt 2=$v  "; python bench.py --no-cpu-baseline --no-api --opt 2=$v $args 2>/dev/null | line
    done
  done
done
