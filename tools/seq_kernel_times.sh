#!/bin/bash
# per-kernel durations of the sequential (one batch in flight) forward for the in-tree build and, when present, the
# reference build sylber_amd/libsylber_hip_ref.so (runs ON THE GPU BOX)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/seqk; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/new -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-api --no-overlap > $OUT/new.log 2>&1
if [ -f $ROOT/sylber_amd/libsylber_hip_ref.so ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ref -- python $ROOT/tools/with_lib.py ref $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-api --no-overlap > $OUT/ref.log 2>&1
fi
for w in ref new; do
  f=$(find $OUT/$w -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && { echo "== $w"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel time per step: %.3f ms" % (tot / 13 / 1e6))
for r in rows[:14]:
    print("%6.2f%% %5d calls %9.1f us avg  %s" % (100 * float(r['TotalDurationNs']) / tot, int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:90]))
PY
  }
done
