set -x
mkdir -p gpurun_out/r06x
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r06x/trace$rep -- python $ROOT/bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 12 --warmup 4 --gather root > $ROOT/gpurun_out/r06x/selftest$rep.log 2>&1
F=$(find $ROOT/gpurun_out/r06x/trace$rep -name '*kernel_trace.csv' | head -1)
python $ROOT/tools/trace_cadence.py $F > $ROOT/gpurun_out/r06x/cadence$rep.txt 2>&1
grep '^{' $ROOT/gpurun_out/r06x/selftest$rep.log | tail -1 | cut -c1-200 >> $ROOT/gpurun_out/r06x/cadence$rep.txt
rm -rf $ROOT/gpurun_out/r06x/trace$rep
done
