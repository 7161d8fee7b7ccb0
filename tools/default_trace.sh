#!/bin/bash
# kernel trace of the DEFAULT bench workload (4 096 registrations per step, 256 slots, 5 + 20 steps) without the extra legs, whose other
# launches of the same kernel (hard guesses, host-fed, single registrations) would mix into the per-kernel average:
#   gpurun_out/prof_default/{kernel_stats.csv,bench_trace.json}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_default
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o b -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras > $OUT/bench_trace.json 2> $OUT/trace.err
python - $OUT <<'PY'
import csv, glob, os, sqlite3, sys
out = sys.argv[1]
dbs = sorted(glob.glob(os.path.join(out, "trace", "**", "*.db"), recursive=True))
c = sqlite3.connect(dbs[-1])
rows = c.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by name order by 3 desc").fetchall()
tot = float(sum(r[2] for r in rows)) or 1.0
with open(os.path.join(out, "kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), int(r[4]), int(r[5]), round(100.0 * r[2] / tot, 2)])
PY
rm -rf $OUT/trace
head -5 $OUT/kernel_stats.csv
