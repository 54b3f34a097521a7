set -x
mkdir -p gpurun_out/r06q
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_dist.py -x -q > gpurun_out/r06q/pytest.log 2>&1
for rep in 1 2 3; do
timeout 300 python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>>gpurun_out/r06q/err.txt | tail -1 > gpurun_out/r06q/ring_$rep.json
timeout 300 python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --exchange-fresh-results 2>>gpurun_out/r06q/err.txt | tail -1 > gpurun_out/r06q/fresh_$rep.json
done
