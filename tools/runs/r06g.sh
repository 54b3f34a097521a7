set -x
mkdir -p gpurun_out/r06g
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06g/pytest_gpu.log 2>&1
timeout 900 python bench.py > gpurun_out/r06g/bench_full.json 2> gpurun_out/r06g/bench_full.err
timeout 300 python tools/api_timeline.py > gpurun_out/r06g/api_timeline.txt 2>&1
