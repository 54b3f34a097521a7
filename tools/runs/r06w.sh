set -x
mkdir -p gpurun_out/r06w
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5 6; do
timeout 300 python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>>gpurun_out/r06w/err.txt | tail -1 > gpurun_out/r06w/A_$rep.json
done
timeout 600 python -m pytest tests/test_gpu_dist.py -x -q > gpurun_out/r06w/pytest.log 2>&1
