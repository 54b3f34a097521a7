python -m pytest tests/test_gpu_encoder.py -x -q 2>&1 | tail -2
for W in 256 512 1024 4096; do
  export SYLBER_CONV0_WGS=$W
  echo "== conv0 mfma wgs $W"; python bench.py --no-cpu-baseline --no-api 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_forward']; print(d['value'], d['ms_per_step'], k['conv0_gn_gelu'], k['gemm_conv1'], d['roofline_frontend']['frac'])"
done
unset SYLBER_CONV0_WGS; export SYLBER_CONV0_VALU=1; echo "== valu"; python bench.py --no-cpu-baseline --no-api 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_forward']; print(d['value'], d['ms_per_step'], k['conv0_gn_gelu'], k['gemm_conv1'], d['roofline_frontend']['frac'])"
