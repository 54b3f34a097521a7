set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06final
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06final/pytest_gpu.log 2>&1
timeout 300 python __graft_entry__.py --smoke > gpurun_out/r06final/smoke.log 2>&1
timeout 3000 bash tools/collect_profiles.sh > gpurun_out/r06final/collect.log 2>&1
timeout 300 python tools/api_timeline.py > gpurun_out/r06final/api_timeline.txt 2>&1
