set -x
mkdir -p gpurun_out/r06k
cd $GRAFT_REPO_ROOT
./tools/ubench/mfma_shape > gpurun_out/r06k/mfma_shape.txt 2>&1
timeout 300 python tools/api_timeline.py > gpurun_out/r06k/api_timeline.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/r06k/bench_api.json 2> gpurun_out/r06k/bench_api.err
