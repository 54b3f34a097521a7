set -x
mkdir -p gpurun_out/r06b
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "tail_split or every_tile or residual_gemm or persistent_seams or attention" > gpurun_out/r06b/pytest_ops.log 2>&1
timeout 600 python tools/tail_sweep.py > gpurun_out/r06b/tail_sweep.md 2> gpurun_out/r06b/tail_sweep.err
timeout 600 python tools/shape_sweep.py --batches 8,24,32,48 --seconds 10,15,60 --json gpurun_out/r06b/sweep_auto.json > gpurun_out/r06b/sweep_auto.md 2> gpurun_out/r06b/sweep_auto.err
timeout 600 python tools/shape_sweep.py --batches 8,24,32,48 --seconds 10,15,60 --opt 8=-1 > gpurun_out/r06b/sweep_nosplit.md 2> gpurun_out/r06b/sweep_nosplit.err
timeout 300 python bench.py --no-cpu-baseline --no-api --no-other-configs > gpurun_out/r06b/bench_head.json 2> gpurun_out/r06b/bench_head.err
