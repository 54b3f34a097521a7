#!/bin/bash
# same-box A/B of the whole step with one GEMM tile id forced (runs ON THE GPU BOX):  bash tools/ab_tile.sh 80 [more bench flags]
T=$1; shift
for rep in 1 2 3; do
  echo "== auto (rep $rep)"; python bench.py --no-cpu-baseline --no-api "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  echo "== tile $T (rep $rep)"; python bench.py --no-cpu-baseline --no-api --gemm-tile $T "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
