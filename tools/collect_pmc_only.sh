#!/bin/bash
# the rocprofv3 passes of tools/collect_profiles.sh alone (kernel traces + PMC), into gpurun_out/final (bench lines left as they are)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
F="--no-cpu-baseline --no-api --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pipelined -- python $ROOT/bench.py --steps 10 --warmup 3 $F > $OUT/trace_pipelined.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sequential -- python $ROOT/bench.py --steps 10 --warmup 3 $F --no-overlap > $OUT/trace_sequential.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $ROOT/bench.py --steps 3 --warmup 1 $F --no-overlap > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $ROOT/bench.py --steps 3 --warmup 1 $F --no-overlap > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/pmc_mfma -- python $ROOT/bench.py --steps 3 --warmup 1 $F --no-overlap > $OUT/pmc_mfma.log 2>&1
cd $ROOT
cp $(find $OUT/pmc_mfma -name '*counter_collection.csv' | head -1) $OUT/mfma_counters.csv
rm -rf $OUT/pmc_mfma
mkdir -p $OUT/traffic
cp $(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1) $OUT/traffic/FETCH_SIZE_counter_collection.csv
cp $(find $OUT/pmc_write -name '*counter_collection.csv' | head -1) $OUT/traffic/WRITE_SIZE_counter_collection.csv
for t in pipelined sequential; do cp $(find $OUT/trace_$t -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$t.csv; done
rm -rf $OUT/trace_pipelined $OUT/trace_sequential $OUT/pmc_fetch $OUT/pmc_write
ls $OUT/traffic
