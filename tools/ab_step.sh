#!/bin/bash
# same-box A/B of the whole step: sylber_amd/libsylber_hip_ref.so (another commit's build) against the in-tree build (runs ON THE GPU BOX)
for rep in 1 2 3; do
  echo "== ref (rep $rep)"; python tools/with_lib.py ref bench.py --no-cpu-baseline --no-api "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  echo "== new (rep $rep)"; python bench.py --no-cpu-baseline --no-api "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
