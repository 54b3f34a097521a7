python -m pytest tests/test_gpu_encoder.py tests/test_gpu_fp16.py tests/test_gpu_concurrency.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -3
for rep in 1 2; do
for L in valu mfma; do
  if [ $L = valu ]; then export SYLBER_CONV0_VALU=1; else unset SYLBER_CONV0_VALU; fi
  echo "== conv0 $L"; python bench.py --no-cpu-baseline --no-api 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_forward']; print(d['value'], d['ms_per_step'], k['conv0_gn_gelu'], k['gemm_conv1'], d['roofline_frontend']['frac'])"
done; done
