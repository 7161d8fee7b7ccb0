#!/bin/bash
# Run ON THE MI355X BOX: kernel timeline of a short bench run -> the gaps between consecutive launches (developer tool).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/gaps
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace -d $OUT/trace -o b --output-format csv -- python $R/bench.py --no-cpu --no-extras --warmup 0 --steps 3 "$@" > $OUT/bench.json 2> $OUT/err
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY' > $OUT/gaps.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-28:]
    out.append((name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = e
# the last 60 launches
for n, s, d, g in out[-60:]:
    print("%-30s start %10.1f us  dur %8.1f us  gap %7.1f us" % (n, s, d, g))
PY
tail -60 $OUT/gaps.txt
