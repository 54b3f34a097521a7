#!/bin/bash
# Turns what tools/collect_profiles.sh left in gpurun_out/final (merged back by gpurun) into the tracked profiles/rNN_* files.
#   bash tools/publish_profiles.sh r03
set -eu
R=${1:-r03}
F=gpurun_out/final
python tools/pmc_traffic.py $F/traffic profiles/${R}_hbm_traffic > /dev/null
python tools/pmc_mfma.py $F/mfma_counters.csv profiles/${R}_mfma_util.md > /dev/null
cp $F/kernel_stats_pipelined.csv profiles/${R}_kernel_stats_pipelined.csv
cp $F/kernel_stats_sequential.csv profiles/${R}_kernel_stats_sequential.csv
for n in bf16 fp16 fp8 fp32 split16 longform sequential ragged exchange_selftest; do tail -1 $F/bench_$n.json > profiles/${R}_bench_$n.json; done
for tag in fp8 longform; do
  if [ -d $F/traffic_$tag ]; then
    python tools/pmc_traffic.py $F/traffic_$tag profiles/${R}_hbm_traffic_$tag > /dev/null || true
    cp $F/kernel_stats_${tag}_sequential.csv profiles/${R}_kernel_stats_${tag}_sequential.csv
  fi
done
python - "$R" <<'PY'
import json, sys
R = sys.argv[1]
for n in ("bf16", "fp16", "fp8", "fp32", "split16", "longform", "sequential", "ragged", "exchange_selftest"):
    d = json.load(open("profiles/%s_bench_%s.json" % (R, n)))
    r = d.get("roofline") or {}
    print("%-18s %8.1f audio-s/s %7.3f ms/step  frac %s  enc %s" % (n, d["value"], d["ms_per_step"], r.get("frac"), (r.get("encoder_gemms") or {}).get("frac")))
PY
