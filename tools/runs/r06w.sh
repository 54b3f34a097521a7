set -x
mkdir -p gpurun_out/r06w
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
timeout 300 python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>>gpurun_out/r06w/err.txt | tail -1 > gpurun_out/r06w/A_$rep.json
done
