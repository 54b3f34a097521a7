#!/bin/bash
# samples socket power and shader clock (rocm-smi) while a workload runs; prints min / median / max.   bash tools/power_probe.sh "<command>"
CMD=${1:-"python bench.py --no-api --no-cpu-baseline --no-other-configs --steps 2000 --warmup 20"}
LOG=/tmp/power_probe.log; : > $LOG
( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n' >> $LOG; echo >> $LOG; sleep 0.2; done ) &
SP=$!
sleep 1
bash -c "$CMD" > /tmp/power_probe.out 2>/dev/null
kill $SP
rocm-smi --showmaxpower 2>/dev/null | grep -i -E "max|power" | head -3
python - <<'PY'
import json, statistics
pw, ck = [], []
for ln in open("/tmp/power_probe.log"):
    ln = ln.strip()
    if not ln: continue
    try: d = json.loads(ln)
    except Exception: continue
    for card, v in d.items():
        for k, x in v.items():
            kl = k.lower()
            if "power" in kl and "(w)" in kl:
                try: pw.append(float(x))
                except Exception: pass
            if "sclk" in kl and "clock" in kl:
                try: ck.append(float(str(x).strip("()Mhz ")))
                except Exception: pass
def s(a): return "n=%d min %.0f median %.0f max %.0f" % (len(a), min(a), statistics.median(a), max(a)) if a else "none"
print("power W:", s(pw)); print("sclk MHz:", s(ck))
PY
tail -c 300 /tmp/power_probe.out | head -c 300; echo
head -c 600 /tmp/power_probe.log
