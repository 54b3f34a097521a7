set -x
mkdir -p gpurun_out/r06n
cd $GRAFT_REPO_ROOT
timeout 900 python tools/tile_pick_sweep.py --shapes 8x60,24x10,24x15,48x10,32x10,16x30 --tiles=-1,51,91,4 > gpurun_out/r06n/tile_pick.md 2> gpurun_out/r06n/tile_pick.err
timeout 600 python tools/shape_sweep.py --batches 8,16,24,32,48 --seconds 10,15,30,60 --json gpurun_out/r06n/sweep.json > gpurun_out/r06n/sweep_default.md 2> gpurun_out/r06n/sweep.err
timeout 600 python bench.py --no-cpu-baseline --no-api > gpurun_out/r06n/bench.json 2> gpurun_out/r06n/bench.err
