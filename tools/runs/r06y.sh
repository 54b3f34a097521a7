set -x
mkdir -p gpurun_out/r06y
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
timeout 300 python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>>gpurun_out/r06y/err.txt | tail -1 > gpurun_out/r06y/A_$rep.json
timeout 300 python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --exchange-default-stream 2>>gpurun_out/r06y/err.txt | tail -1 > gpurun_out/r06y/D_$rep.json
done
