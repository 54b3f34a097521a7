set -x
mkdir -p gpurun_out/r06h
cd $GRAFT_REPO_ROOT
timeout 300 python tools/pad_probe.py > gpurun_out/r06h/pad_probe.txt 2>&1
timeout 900 bash tools/exchange_channels_sweep.sh > gpurun_out/r06h/exchange_channels.md 2> gpurun_out/r06h/exchange_channels.err
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r06h/pytest_gpu.log 2>&1
