set -x
mkdir -p gpurun_out/r06z
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
timeout 300 python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>>gpurun_out/r06z/err.txt | tail -1 > gpurun_out/r06z/A_$rep.json
timeout 300 python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --gc-in-timing 2>>gpurun_out/r06z/err.txt | tail -1 > gpurun_out/r06z/G_$rep.json
done
timeout 600 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/r06z/bench_api.json 2>>gpurun_out/r06z/err.txt
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_e2e.py -x -q > gpurun_out/r06z/pytest.log 2>&1
