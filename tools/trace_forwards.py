"""Reconstructs, from a rocprofv3 --kernel-trace CSV, when each forward ran (conv0_stats_kernel ... the last layernorm before the next conv0_stats on the same
queue) and prints the last `n` of them with their hardware queue, overlap with the previous forward, and the copy / boundary-detection kernels in between:
shows at a glance whether two batches in flight really overlap (development aid for the exchange step).   python tools/trace_forwards.py trace.csv [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in rows:
    r["s"] = float(r["Start_Timestamp"]) / 1e6
    r["e"] = float(r["End_Timestamp"]) / 1e6
rows.sort(key=lambda r: r["s"])
t0 = rows[0]["s"]
fw = []          # [queue, start, end, kernels]
open_ = {}
for r in rows:
    q, nm = r["Queue_Id"], r["Kernel_Name"]
    if nm.startswith("conv0_stats_kernel"):
        open_[q] = [q, r["s"], r["e"], 1]
        fw.append(open_[q])
    elif q in open_ and not nm.startswith(("segment_", "__amd", "void at::", "at::")):
        open_[q][2] = max(open_[q][2], r["e"]); open_[q][3] += 1
fw.sort(key=lambda f: f[1])
last = fw[-n:]
print("forwards in the trace: %d; the last %d (ms relative to the first of them)" % (len(fw), len(last)))
b = last[0][1]
prev_end = None
for q, s, e, k in last:
    ov = "" if prev_end is None else ("overlaps the previous by %.2f ms" % (prev_end - s) if prev_end > s else "starts %.2f ms AFTER the previous ended" % (s - prev_end))
    print("  queue %-3s %8.2f -> %8.2f  (%.2f ms, %d kernels)  %s" % (q, s - b, e - b, e - s, k, ov))
    prev_end = e if prev_end is None else max(prev_end, e)
print("other kernels in that window (copies, boundary detection, reductions):")
for r in rows:
    if r["s"] >= b and (r["Kernel_Name"].startswith(("segment_", "__amd_rocclr_copy")) or "elementwise" in r["Kernel_Name"] or "nccl" in r["Kernel_Name"].lower()):
        print("  queue %-3s %8.2f -> %8.2f  %s" % (r["Queue_Id"], r["s"] - b, r["e"] - b, r["Kernel_Name"][:50]))
