set -x
mkdir -p gpurun_out/r06l
cd $GRAFT_REPO_ROOT
./tools/ubench/mfma_shape > gpurun_out/r06l/mfma_shape.txt 2>&1
timeout 600 python tools/tile_pick_sweep.py --shapes 8x60,24x10,24x15 --tiles -1,91,51,97,57,4 > gpurun_out/r06l/tile_pick.md 2> gpurun_out/r06l/tile_pick.err
timeout 600 python tools/tile_pick_sweep.py --shapes 8x60,24x10,24x15 --tiles -1 --opt 12=5 > gpurun_out/r06l/tile_pick_m5.md 2>> gpurun_out/r06l/tile_pick.err
