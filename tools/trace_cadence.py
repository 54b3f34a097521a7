"""start-to-start cadence of the forwards in a rocprofv3 kernel trace (ms), with hardware queue and duration, plus when the gather copies of each step ran
relative to the forward that produced their data.   python tools/trace_cadence.py kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = float(r["Start_Timestamp"]) / 1e6
    r["e"] = float(r["End_Timestamp"]) / 1e6
rows.sort(key=lambda r: r["s"])
fw, open_ = [], {}
for r in rows:
    q, nm = r["Queue_Id"], r["Kernel_Name"]
    if nm.startswith("conv0_stats_kernel"):
        open_[q] = [q, r["s"], r["e"]]
        fw.append(open_[q])
    elif q in open_ and not nm.startswith(("segment_", "__amd", "void at::", "at::")):
        open_[q][2] = max(open_[q][2], r["e"])
fw.sort(key=lambda f: f[1])
copies = [r for r in rows if r["Kernel_Name"].startswith("__amd_rocclr_copyBuffer") and r["e"] - r["s"] > 0.02]
print("forward  queue  start-to-start  duration  gap-to-previous-end   big copies (>20 us) that started during it: queue@offset")
for j, (q, s, e) in enumerate(fw):
    d = s - fw[j - 1][1] if j else 0.0
    prev_end = max(f[2] for f in fw[:j]) if j else s
    cs = ["%s@%.1f" % (c["Queue_Id"], c["s"] - s) for c in copies if s <= c["s"] < (fw[j + 1][1] if j + 1 < len(fw) else e)]
    print("%5d    %-4s   %8.2f      %7.2f    %8.2f      %s" % (j, q, d, e - s, s - prev_end, " ".join(cs[:8])))
