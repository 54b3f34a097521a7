"""Summarise rocprofv3 PMC passes over tools/gemm_bench_one.py into a per-kernel cycle account of the GEMM K loop
(VERDICT r2 item 1(a)).  Every pass directory holds one *counter_collection.csv; counters are summed over their instances
by rocprofv3 (SQ_*: all SEs/SIMDs; GRBM_GUI_ACTIVE: 8 XCDs).

    python tools/pmc_gemm_account.py <dir with pass_* subdirs> <label>

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles (x4 = shader
cycles) summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (32 per v_mfma_f32_32x32x16_bf16)."""
import collections
import csv
import glob
import os
import sys


def main(root, label):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    nd = collections.defaultdict(lambda: collections.defaultdict(set))
    wall = collections.defaultdict(dict)
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm" not in k:
                continue
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            nd[k][r["Counter_Name"]].add(r["Dispatch_Id"])
            if "Start_Timestamp" in r:
                wall[k][(f, r["Dispatch_Id"])] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    print("## %s" % label)
    for k, c in per.items():
        print("\n`%s`\n" % k[:110])
        avg = {n: v / max(len(nd[k][n]), 1) for n, v in c.items()}
        w = sorted(wall[k].values())
        wus = w[len(w) // 2] / 1e3 if w else float("nan")
        print("| counter | per dispatch | reading |")
        print("|---|---:|---|")
        gui = avg.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        wc = avg.get("SQ_WAVE_CYCLES", 0.0)
        for n in sorted(avg):
            v = avg[n]
            note = ""
            if n == "GRBM_GUI_ACTIVE" and wus == wus:
                note = "/8 XCDs = %.0f cycles resident; %.2f GHz over the %.1f us dispatch" % (gui, gui / (wus * 1e3), wus)
            elif n == "SQ_VALU_MFMA_BUSY_CYCLES" and gui:
                note = "MfmaUtil = %.1f %% of 1024 SIMDs x resident cycles" % (100 * v / (gui * 1024))
            elif n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
                       "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_VMEM",
                       "SQ_ACTIVE_INST_FLAT", "SQ_INST_CYCLES_SALU") and wc:
                note = "%.1f %% of SQ_WAVE_CYCLES" % (100 * v / wc)
            elif n == "SQ_WAVE_CYCLES" and gui:
                note = "x4 / (resident cycles x 2048 wave slots of 8-wave WGs) = %.1f %% wave occupancy of 2 waves/SIMD" % (
                    100 * v * 4 / (gui * 1024 * 2))
            elif n == "SQ_BUSY_CYCLES" and gui:
                note = "(per-SE busy; informational)"
            print("| %s | %.4g | %s |" % (n, v, note))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
