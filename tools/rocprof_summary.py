"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel stats table.

    python tools/rocprof_summary.py gpurun_out/prof/r01_results.db > profiles/r01_kernel_stats.md
"""
import sqlite3
import sys
from collections import defaultdict


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    scols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    rows = cur.execute(
        "select s.%s, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id" % name_col
    ).fetchall()
    agg = defaultdict(list)
    for name, st, en in rows:
        agg[name].append(en - st)
    total = sum(sum(v) for v in agg.values())
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        short = name if len(name) < 110 else name[:107] + "..."
        print("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (short, len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3,
                                                               min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / total))
    print("\ntotal kernel time %.3f ms over %d dispatches" % (total / 1e6, len(rows)))


if __name__ == "__main__":
    main(sys.argv[1])
