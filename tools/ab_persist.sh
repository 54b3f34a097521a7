#!/bin/bash
# what does boundary detection cost in the pipelined step, and why?  (bench.py --skip-segment / --out-sets / persistent vs hardware-dispatched tiles)
run() { python bench.py --no-cpu-baseline --no-api --steps 40 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2; do
echo "full             $(run)"
echo "skip             $(run --skip-segment)"
echo "full, 4 out sets $(run --out-sets 4)"
echo "skip, 4 out sets $(run --out-sets 4 --skip-segment)"
echo "full, 6 out sets $(run --out-sets 6)"
done
