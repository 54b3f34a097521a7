set -x
mkdir -p gpurun_out/r06r
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r06r/trace -- python $ROOT/bench.py --exchange-selftest --no-api --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --gather root > $ROOT/gpurun_out/r06r/selftest.log 2>&1
cp $(find $ROOT/gpurun_out/r06r/trace -name '*kernel_stats.csv' | head -1) $ROOT/gpurun_out/r06r/kernel_stats.csv
python - <<'PY'
import csv, glob, os, collections
root = os.environ.get("GRAFT_REPO_ROOT", ".")
f = glob.glob(root + "/gpurun_out/r06r/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# per stream/queue: busy time and kernel names
by = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for r in rows:
    q = r.get("Queue_Id") or r.get("Stream_Id") or "?"
    d = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    by[q][0] += 1; by[q][1] += d; by[q][2][r["Kernel_Name"][:60]] += d
with open(root + "/gpurun_out/r06r/queues.txt", "w") as fh:
    t0 = min(float(r["Start_Timestamp"]) for r in rows); t1 = max(float(r["End_Timestamp"]) for r in rows)
    fh.write("span %.1f ms, columns of the trace: %s\n" % ((t1 - t0) / 1e6, list(rows[0].keys())))
    for q, (n, us, names) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        fh.write("queue %s: %d kernels, %.1f ms busy\n" % (q, n, us / 1e3))
        for nm, u in names.most_common(8):
            fh.write("    %8.1f ms  %s\n" % (u / 1e3, nm))
PY
rm -rf $ROOT/gpurun_out/r06r/trace
