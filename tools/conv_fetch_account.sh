#!/bin/bash
# Runs ON THE GPU BOX: where the 2x over-fetch of the conv GEMMs (dominant kernel, tile 97) comes from.  FETCH_SIZE of the conv1 launch
# as it is (N = 512: two column tiles per row tile; ldx = 1024: output row m reads input rows 2m .. 2m+2, so tap 2 of row m is tap 0 of
# row m+1), with ONE column tile (N = 256), and without overlapping rows (ldx = K = 1536), cold operands in every case
# (tools/gemm_bench_one.py cfg + 200000 writes 1 GiB between the launches).  One --pmc pass per shape, no tracing flags.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/conv_fetch
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A SH
SH[conv1_n512_overlap]="524288 512 1536 1024 0 1 200097"
SH[conv1_n256_overlap]="524288 256 1536 1024 0 1 200097"
SH[conv1_n512_norows]="524288 512 1536 1536 0 1 200097"
SH[conv1_n256_norows]="524288 256 1536 1536 0 1 200097"
SH[conv1_n512_chunkmajor]="524288 512 1536 1024 0 1 600097"
SH[conv1_n256_chunkmajor]="524288 256 1536 1024 0 1 600097"
for name in conv1_n512_overlap conv1_n256_overlap conv1_n512_norows conv1_n256_norows conv1_n512_chunkmajor conv1_n256_chunkmajor; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$name -- python $ROOT/tools/gemm_bench_one.py ${SH[$name]} > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name '*counter_collection.csv' | head -1)
  python - "$name" "$f" "${SH[$name]}" <<'PY'
import csv, sys
name, path, shape = sys.argv[1], sys.argv[2], sys.argv[3].split()
m, n, k, ldx = (int(x) for x in shape[:4])
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "gemmb_bf16_kernel" in r["Kernel_Name"]]
fetch = sum(v) / len(v) * 1024 * 2                      # gfx950: FETCH_SIZE counts half of wide coalesced reads
rows_in = (m - 1) * ldx + k                             # distinct A elements the launch touches
print("%-22s launches %2d  fetch %8.1f MB per launch  | distinct A bytes %8.1f MB, W %.1f MB  -> fetch / distinct = %.2f" % (
    name, len(v), fetch / 1e6, rows_in * 2 / 1e6, n * k * 2 / 1e6, fetch / (rows_in * 2 + n * k * 2)))
PY
  grep -h "us " $OUT/$name.log | tail -1
done
rm -rf $OUT/*/
