python -m pytest tests/test_gpu_fp8.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2 3; do
  echo "== ref fp8 (rep $rep)"; python tools/with_lib.py ref bench.py --precision fp8 --no-cpu-baseline --no-api 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  echo "== new fp8 (rep $rep)"; python bench.py --precision fp8 --no-cpu-baseline --no-api 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
