for rep in 1 2 3; do
for f in "" "--conv0-valu"; do echo "== conv0 ${f:-mfma}"; python bench.py --no-cpu-baseline --no-api $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_forward']; print(d['value'], d['ms_per_step'], k['conv0_gn_gelu'], k['gemm_conv1'])"; done
done
