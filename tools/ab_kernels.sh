#!/bin/bash
# same-box A/B of two builds (runs ON THE GPU BOX): per-kernel ms per forward of the reference build (sylber_amd/libsylber_hip_ref.so)
# and of the in-tree build, interleaved; extra arguments go to bench.py (e.g. --no-overlap)
for rep in 1 2; do
  python tools/with_lib.py ref bench.py --no-cpu-baseline --no-api "$@" 2>/dev/null | tail -1 > /tmp/ab_ref.json
  python bench.py --no-cpu-baseline --no-api "$@" 2>/dev/null | tail -1 > /tmp/ab_new.json
  python - <<'PY'
import json
a = json.load(open("/tmp/ab_ref.json")); b = json.load(open("/tmp/ab_new.json"))
print("ref %.1f audio-s/s %.3f ms/step (median %.3f) | new %.1f %.3f (median %.3f)" % (a["value"], a["ms_per_step"], a["ms_per_step_median"], b["value"], b["ms_per_step"], b["ms_per_step_median"]))
ka, kb = a["kernel_ms_per_forward"], b["kernel_ms_per_forward"]
print("  " + "  ".join("%s %.3f>%.3f" % (k, ka[k], kb.get(k, 0)) for k in ka))
print("  sum %.3f > %.3f ; encoder gemms %.4f > %.4f" % (sum(ka.values()), sum(kb.values()), a["roofline"]["encoder_gemms"]["frac"], b["roofline"]["encoder_gemms"]["frac"]))
PY
done
