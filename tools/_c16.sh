python -m pytest tests/test_gpu_encoder.py tests/test_gpu_split16.py tests/test_gpu_fp16.py -x -q 2>&1 | tail -2
for rep in 1 2; do
for L in ref new; do
  if [ $L = ref ]; then export SYLBER_HIP_LIB=$(pwd)/sylber_amd/libsylber_hip_ref.so; else unset SYLBER_HIP_LIB; fi
  echo "== $L"; python bench.py --no-cpu-baseline --no-api 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_forward']; print(d['value'], d['ms_per_step'], {x:k[x] for x in ('attention','posconv')})"
done; done
