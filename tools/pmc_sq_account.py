"""Per-kernel SQ accounting from rocprofv3 PMC passes (development aid): where the waves of a kernel spend their cycles.
    python tools/pmc_sq_account.py a.csv [b.csv ...]      (counter_collection.csv files of separate --pmc passes over the same command)
WAVE_CYCLES ~ WAIT_ANY (parked: s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY (MI355X_MICROARCH.md, SQ counters)."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
nl = collections.defaultdict(set)
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        nl[k].add(r["Dispatch_Id"])
names = sorted({c for v in acc.values() for c in v})
print("| kernel | launches | " + " | ".join(names) + " |")
print("|---|---:|" + "---:|" * len(names))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if "gemm" not in k and "Cijk" not in k:
        continue
    n = max(1, len(nl[k]))
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1.0
    cells = []
    for c in names:
        x = v.get(c, 0.0)
        cells.append("%.3g%s" % (x / n, " (%.0f %%)" % (100 * x / wc) if c != "SQ_WAVE_CYCLES" and c.startswith("SQ_W") or c.startswith("SQ_ACTIVE") else ""))
    print("| `%s` | %d | %s |" % (k[:70], n, " | ".join(cells)))
