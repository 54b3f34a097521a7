set -x
mkdir -p gpurun_out/r06s
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; timeout 300 "$@" 2>>gpurun_out/r06s/err.txt | tail -1 > gpurun_out/r06s/$tag.json; }
for rep in 1 2; do
run A_$rep python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5
run B_$rep python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --exchange-lookahead 1 --exchange-ingest-stream engine --exchange-fresh-results
run C_$rep env SYLBER_NO_STREAM_PROBE=1 python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5
run D_$rep env GPU_MAX_HW_QUEUES=4 python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5
run E_$rep python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --plain-streams
done
