"""Summarise rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, collected in SEPARATE runs as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) into per-kernel HBM traffic per launch.

    python tools/pmc_traffic.py gpurun_out/traffic profiles/r01_hbm_traffic

Units/corrections (guide, section HBM): counters are in KiB; on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled here; WRITE_SIZE is
calibrated on conv0_gn_gelu's known 1.0486 GB output (reads 1024000 KiB: exact)."""
import collections
import csv
import hashlib
import json
import os
import sys


def csrc_sha16():
    """identity of the kernel sources the counters were collected with (bench.py repeats `traffic` only on a match): bench.py's own
    function -- kernel sources AND the loop generators"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    return bench.csrc_sha16()


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main(src, dst):
    f = load(src + "/FETCH_SIZE_counter_collection.csv")
    w = load(src + "/WRITE_SIZE_counter_collection.csv")
    rows = []
    for k in f:
        n = len(f[k])
        fetch = sum(f[k]) / n * 1024.0 * 2.0          # bytes per launch, gfx950 half-count correction
        write = sum(w.get(k, [0.0])) / max(len(w.get(k, [])), 1) * 1024.0
        rows.append((k, n, fetch, write))
    rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
    gemm = [r for r in rows if "gemm" in r[0]]
    n_gemm = sum(r[1] for r in gemm)
    tot = sum((r[2] + r[3]) * r[1] for r in gemm)
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-overlap (tools/collect_profiles.sh)",
           "csrc_sha16": csrc_sha16(), "fetch_correction": 2.0, "gemm_family_bytes_per_launch": tot / n_gemm, "gemm_launches_profiled": n_gemm,
           "kernels": {r[0]: {"launches": r[1], "fetch_bytes_per_launch": r[2], "write_bytes_per_launch": r[3]} for r in rows}}
    json.dump(out, open(dst + ".json", "w"), indent=1)
    with open(dst + ".md", "w") as fh:
        fh.write("# HBM traffic per launch (rocprofv3 PMC, gfx950)\n\n" + out["source"] + "\n\n")
        fh.write("FETCH_SIZE x2 (gfx950 half-count of wide coalesced reads), WRITE_SIZE as reported; KiB -> bytes.\n\n")
        fh.write("| kernel | launches | fetch MB/launch (corrected) | write MB/launch |\n|---|---:|---:|---:|\n")
        for k, n, fe, wr in rows:
            fh.write("| `%s` | %d | %.1f | %.1f |\n" % (k[:100], n, fe / 1e6, wr / 1e6))
        fh.write("\nGEMM family: %.1f MB per launch averaged over %d launches\n" % (tot / n_gemm / 1e6, n_gemm))
    print(open(dst + ".md").read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
