set -x
mkdir -p gpurun_out/r06d
cd $GRAFT_REPO_ROOT
timeout 1500 python tools/tile_pick_sweep.py > gpurun_out/r06d/tile_pick.md 2> gpurun_out/r06d/tile_pick.err
