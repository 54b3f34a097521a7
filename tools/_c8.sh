mkdir -p gpurun_out/r3c8
python -m pytest tests/test_gpu_ops.py -x -q -k "tile_config or schedule_variants" > gpurun_out/r3c8/pytest.log 2>&1; tail -3 gpurun_out/r3c8/pytest.log
python tools/gemm_bench.py 9010,9040 > gpurun_out/r3c8/gemm_u.txt 2>&1
python tools/gemm_bench.py 9010,9040 >> gpurun_out/r3c8/gemm_u.txt 2>&1
cat gpurun_out/r3c8/gemm_u.txt
