#!/bin/bash
# Runs ON THE GPU BOX: PMC account of the GEMM kernels in isolation (tools/gemm_bench_one.py), one counter group per
# pass (8 SQ slots), no tracing flags beside --pmc.  Output: gpurun_out/gemm_pmc/<shape>/pass_*/...
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/gemm_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
declare -A SH
SH[sq4096]="4096 4096 4096 4096 0 0 10"
SH[ffn1]="16384 3072 768 768 0 1 10"
SH[out]="16384 768 768 768 6 0 4"
SH[qkv]="16384 2304 768 768 3 0 4"
SH[ffn2]="16384 768 3072 3072 6 0 4"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"
P3="SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"
for name in "${!SH[@]}"; do
  i=0; mkdir -p $OUT/$name
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    rocprofv3 --pmc $P --output-format csv -d $OUT/$name/pass_$i -- python $ROOT/tools/gemm_bench_one.py ${SH[$name]} > $OUT/$name/pass_$i.log 2>&1 || echo "pass $i of $name failed" >> $OUT/errors.txt
  done
  python $ROOT/tools/pmc_gemm_account.py $OUT/$name "$name ${SH[$name]}" > $OUT/$name.md 2>> $OUT/errors.txt
  find $OUT/$name -name '*.csv' -size +2M -delete
done
cat $OUT/*.md | head -150
