mkdir -p gpurun_out/r3c9
python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py -x -q > gpurun_out/r3c9/pytest.log 2>&1; tail -3 gpurun_out/r3c9/pytest.log
python tools/gemm_bench.py -1,4,10,9010,9040 > gpurun_out/r3c9/gemm.txt 2>&1
python tools/gemm_bench.py -1,4,10,9010,9040 >> gpurun_out/r3c9/gemm.txt 2>&1
cat gpurun_out/r3c9/gemm.txt
python bench.py --no-cpu-baseline --no-api > gpurun_out/r3c9/bench.json 2>gpurun_out/r3c9/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c9/bench.json'))
print(d['value'], d['ms_per_step']); print(d['roofline']['per_launch_tflops']); print(d['roofline']['encoder_gemms']); print(d['kernel_ms_per_forward'])
PY
