#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the bench lines, rocprofv3 kernel traces and PMC passes that profiles/ is built
# from.  Everything lands in gpurun_out/; tools/rocprof_summary.py / tools/pmc_traffic.py turn it into profiles/.
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh'      then, here:  bash tools/publish_profiles.sh r03
# The PMC passes run AFTER the bench lines, so the bench lines of this call cannot carry `roofline.traffic` yet: commit the
# published profiles/rNN_hbm_traffic.json and run `python bench.py --agreement-clips 256` once more for the headline line.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
python bench.py --agreement-clips 256 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
python bench.py --precision fp16 --no-cpu-baseline --agreement-clips 256 > $OUT/bench_fp16.json 2> $OUT/bench_fp16.err
python bench.py --precision fp8 --no-cpu-baseline --no-api --no-other-configs > $OUT/bench_fp8.json 2> $OUT/bench_fp8.err
python bench.py --precision fp32 --no-cpu-baseline --no-api --steps 5 --warmup 1 --inflight 1 > $OUT/bench_fp32.json 2> $OUT/bench_fp32.err
python bench.py --precision split16 --no-cpu-baseline --no-api --steps 10 --warmup 2 --agreement-clips 256 > $OUT/bench_split16.json 2> $OUT/bench_split16.err
python bench.py --batch 8 --clip-seconds 60 --no-cpu-baseline --no-api --no-other-configs > $OUT/bench_longform.json 2> $OUT/bench_longform.err
python bench.py --no-overlap --no-cpu-baseline --no-api --no-other-configs > $OUT/bench_sequential.json 2> $OUT/bench_sequential.err
python bench.py --ragged --no-cpu-baseline --no-api --no-other-configs > $OUT/bench_ragged.json 2> $OUT/bench_ragged.err
python bench.py --exchange-selftest --no-cpu-baseline --no-api 2> $OUT/bench_selftest.err | tail -1 > $OUT/bench_exchange_selftest.json
python tools/attn_bench.py > $OUT/attn_bench.txt 2>&1
python tools/gemm_bench.py -1,4,91,97 > $OUT/gemm_bench.txt 2>&1
# round 6, second collection: the 16x16x32 family against the 32x32x16 kernels and against the vendor library on THIS box
python tools/blaslt_yardstick.py > $OUT/yardstick_hipblaslt.md 2>&1
python tools/mfma16_ab.py 3 > $OUT/mfma16_ab.md 2>&1
bash tools/mfma16_forward_ab.sh 2 > $OUT/mfma16_forward_ab.txt 2>&1
python tools/small_batch_ab.py > $OUT/small_batch_ab.md 2>&1
python tools/small_tiles_ab.py 3 > $OUT/small_tiles_ab.md 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pipelined -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-api --no-other-configs > $OUT/trace_pipelined.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sequential -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-api --no-other-configs --no-overlap > $OUT/trace_sequential.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-other-configs --no-overlap > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-other-configs --no-overlap > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/pmc_mfma -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-other-configs --no-overlap > $OUT/pmc_mfma.log 2>&1
# round 6: the same passes for BASELINE configs[4] (fp8) and configs[3] (8 x 60 s), so that the `other_configs` rooflines of the
# bench line are reproducible from profiles/ (kernel durations + HBM traffic per launch)
for cfg in "fp8 --precision fp8" "longform --batch 8 --clip-seconds 60"; do
  set -- $cfg; tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-api --no-other-configs --no-overlap "$@" > $OUT/trace_$tag.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$tag -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-other-configs --no-overlap "$@" > $OUT/pmc_fetch_$tag.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$tag -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-other-configs --no-overlap "$@" > $OUT/pmc_write_$tag.log 2>&1
  mkdir -p $OUT/traffic_$tag
  cp $(find $OUT/trace_$tag -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_${tag}_sequential.csv
  cp $(find $OUT/pmc_fetch_$tag -name '*counter_collection.csv' | head -1) $OUT/traffic_$tag/FETCH_SIZE_counter_collection.csv
  cp $(find $OUT/pmc_write_$tag -name '*counter_collection.csv' | head -1) $OUT/traffic_$tag/WRITE_SIZE_counter_collection.csv
  rm -rf $OUT/trace_$tag $OUT/pmc_fetch_$tag $OUT/pmc_write_$tag
done
cd $ROOT
cp $(find $OUT/pmc_mfma -name '*counter_collection.csv' | head -1) $OUT/mfma_counters.csv
rm -rf $OUT/pmc_mfma
mkdir -p $OUT/traffic
cp $(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1) $OUT/traffic/FETCH_SIZE_counter_collection.csv
cp $(find $OUT/pmc_write -name '*counter_collection.csv' | head -1) $OUT/traffic/WRITE_SIZE_counter_collection.csv
for t in pipelined sequential; do cp $(find $OUT/trace_$t -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$t.csv; done
# keep the merge-back small: the raw traces are not needed
rm -rf $OUT/trace_pipelined $OUT/trace_sequential $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT $OUT/traffic
tail -c 300 $OUT/bench_bf16.json
