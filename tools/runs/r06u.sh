set -x
mkdir -p gpurun_out/r06u
cd $GRAFT_REPO_ROOT
timeout 300 python tools/rccl_selfcopy_probe.py > gpurun_out/r06u/probe.txt 2>&1
