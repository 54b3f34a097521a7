set -x
mkdir -p gpurun_out/r06a
cd $GRAFT_REPO_ROOT
python tools/shape_sweep.py --json gpurun_out/r06a/sweep_base.json > gpurun_out/r06a/sweep_base.md 2> gpurun_out/r06a/sweep_base.err
python tools/blaslt_yardstick.py > gpurun_out/r06a/yardstick.md 2> gpurun_out/r06a/yardstick.err
python tools/gemm_bench.py -1,4,10,85,91,97 > gpurun_out/r06a/gemm_bench.txt 2>&1
python tools/gemm_cold.py ffn2,ffn1,out,qkv 91,97,85 > gpurun_out/r06a/gemm_cold.txt 2>&1
