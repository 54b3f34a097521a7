set -x
mkdir -p gpurun_out/r06c
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_segment.py -x -q > gpurun_out/r06c/pytest_ops_seg.log 2>&1
timeout 600 python tools/tail_sweep.py > gpurun_out/r06c/tail_sweep.md 2> gpurun_out/r06c/tail_sweep.err
timeout 300 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/r06c/bench_head.json 2> gpurun_out/r06c/bench_head.err
timeout 300 python bench.py --no-cpu-baseline --no-api --no-other-configs --batch 8 --clip-seconds 60 > gpurun_out/r06c/bench_long.json 2> gpurun_out/r06c/bench_long.err
timeout 300 python bench.py --no-cpu-baseline --no-api --no-other-configs --batch 8 --clip-seconds 60 --opt 11=-1 --opt 8=-1 --opt 9=-1 > gpurun_out/r06c/bench_long_r5.json 2> gpurun_out/r06c/bench_long_r5.err
timeout 300 python tools/api_timeline.py > gpurun_out/r06c/api_timeline.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q > gpurun_out/r06c/pytest_e2e.log 2>&1
