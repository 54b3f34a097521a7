mkdir -p gpurun_out/r3c3
python -m pytest tests/test_gpu_ops.py -x -q -k "tile_config or schedule_variants" > gpurun_out/r3c3/pytest.log 2>&1; tail -3 gpurun_out/r3c3/pytest.log
python tools/gemm_bench.py 9010,9020,9021,9022 > gpurun_out/r3c3/gemm_var.txt 2>&1
python tools/gemm_bench.py 9010,9020,9021,9022 >> gpurun_out/r3c3/gemm_var.txt 2>&1
cat gpurun_out/r3c3/gemm_var.txt
python tools/api_profile.py 2>&1 | head -4
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3c3/blaslt_trace -- python $GRAFT_REPO_ROOT/tools/blaslt_yardstick.py > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; cut -d, -f1-4 $(find gpurun_out/r3c3/blaslt_trace -name '*kernel_stats.csv' | head -1) | head -40; find gpurun_out/r3c3/blaslt_trace -name '*kernel_trace.csv' -delete
