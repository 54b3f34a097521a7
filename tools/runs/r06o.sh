set -x
mkdir -p gpurun_out/r06o
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
timeout 300 python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>>gpurun_out/r06o/err.txt | tail -1 > gpurun_out/r06o/st_$rep.json
done
timeout 300 python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 60 --warmup 20 2>>gpurun_out/r06o/err.txt | tail -1 > gpurun_out/r06o/st_long.json
