set -x
mkdir -p gpurun_out/r06i
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
timeout 300 python bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>gpurun_out/r06i/selftest.err | tail -1 > gpurun_out/r06i/selftest_$rep.json
done
timeout 300 python tools/api_timeline.py > gpurun_out/r06i/api_timeline.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_e2e.py -x -q > gpurun_out/r06i/pytest.log 2>&1
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for shp in sq4096 ffn2 ffn1; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $ROOT/gpurun_out/r06i/pmc_$shp -- python $ROOT/tools/yardstick_pair.py $shp > $ROOT/gpurun_out/r06i/pair_pmc_$shp.log 2>&1
  cp $(find $ROOT/gpurun_out/r06i/pmc_$shp -name '*counter_collection.csv' | head -1) $ROOT/gpurun_out/r06i/pmc_$shp.csv
  rm -rf $ROOT/gpurun_out/r06i/pmc_$shp
  python $ROOT/tools/yardstick_pair.py $shp > $ROOT/gpurun_out/r06i/pair_$shp.txt 2>&1
done
