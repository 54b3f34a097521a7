python -m pytest tests/test_gpu_ops.py -x -q -k "tile_config or schedule_variants" 2>&1 | tail -2
python tools/gemm_bench.py 10,60,61,62 2>&1 | grep -v amdgpu
python tools/gemm_bench.py 10,60,61,62 2>&1 | grep -v amdgpu
