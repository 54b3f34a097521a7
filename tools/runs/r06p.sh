set -x
mkdir -p gpurun_out/r06p
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
timeout 300 python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>>gpurun_out/r06p/err.txt | tail -1 > gpurun_out/r06p/ring_$rep.json
TORCH_NCCL_AVOID_RECORD_STREAMS=1 timeout 300 python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>>gpurun_out/r06p/err.txt | tail -1 > gpurun_out/r06p/ringenv_$rep.json
done
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_encoder.py -x -q > gpurun_out/r06p/pytest.log 2>&1
