set -x
mkdir -p gpurun_out/r06t
cd $GRAFT_REPO_ROOT
for rep in 1; do
timeout 300 python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>>gpurun_out/r06t/err.txt | tail -1 > gpurun_out/r06t/A_$rep.json
done
