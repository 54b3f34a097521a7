set -x
mkdir -p gpurun_out/r06e
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-api --no-other-configs > gpurun_out/r06e/head_new_$rep.json 2> gpurun_out/r06e/err.txt
  timeout 300 python bench.py --no-cpu-baseline --no-api --no-other-configs --opt 12=5 > gpurun_out/r06e/head_r5_$rep.json 2>> gpurun_out/r06e/err.txt
done
timeout 600 python tools/shape_sweep.py --batches 8,16,24,32,48 --seconds 10,15,30,60 > gpurun_out/r06e/sweep_new.md 2> gpurun_out/r06e/sweep_new.err
timeout 600 python tools/shape_sweep.py --batches 8,16,24,32,48 --seconds 10,15,30,60 --opt 12=5 > gpurun_out/r06e/sweep_r5.md 2> gpurun_out/r06e/sweep_r5.err
timeout 600 python tools/tile_pick_sweep.py --tiles -1 > gpurun_out/r06e/tile_pick_auto.md 2> gpurun_out/r06e/tile_pick_auto.err
timeout 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_e2e.py -x -q > gpurun_out/r06e/pytest.log 2>&1
