set -x
mkdir -p gpurun_out/r06j
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
 for cfg in "1 engine" "2 own" "2 engine" "1 own"; do
  set -- $cfg
  timeout 300 python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --exchange-lookahead $1 --exchange-ingest-stream $2 2>>gpurun_out/r06j/err.txt | tail -1 > gpurun_out/r06j/st_$1_$2_$rep.json
 done
done
