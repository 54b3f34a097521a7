#!/bin/bash
# Same-box A/B of two builds of libsylber_hip.so (runs ON THE GPU BOX): the reference build sylber_amd/libsylber_hip_ref.so
# (built from another commit, see DESIGN.md method note) against the in-tree build, interleaved, same process order.
#   usage: bash tools/ab_lib.sh [gemm_bench cfg list]
CFGS=${1:--1,4,10}
for rep in 1 2; do
  echo "== ref (rep $rep)"; python tools/with_lib.py ref tools/gemm_bench.py $CFGS 2>&1 | grep -v amdgpu.ids
  echo "== new (rep $rep)"; python tools/gemm_bench.py $CFGS 2>&1 | grep -v amdgpu.ids
done
for rep in 1 2; do
  echo "== ref bench (rep $rep)"; python tools/with_lib.py ref bench.py --no-cpu-baseline --no-api 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['per_launch_tflops'])"
  echo "== new bench (rep $rep)"; python bench.py --no-cpu-baseline --no-api 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['per_launch_tflops'])"
done
