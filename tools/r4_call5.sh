#!/bin/bash
# round-4 GPU call 5: parity after the stale-slot fix, A/B of prebuilt library variants (previous-winner bound / gh-only stage 2), ordering kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/c5.pytest 2>&1; tail -4 gpurun_out/c5.pytest
for rep in 1 2; do
  for L in cur noprev noprev_nogh; do
    ELM_LIB=$PWD/build_ab/lib_$L.so python bench.py --no-cpu --no-extras > gpurun_out/c5_${L}_easy$rep.json 2> gpurun_out/c5_${L}_easy$rep.err || tail -3 gpurun_out/c5_${L}_easy$rep.err
  done
done
for L in cur noprev noprev_nogh; do
  ELM_LIB=$PWD/build_ab/lib_$L.so python bench.py --no-cpu --no-extras --guess hard --steps 6 > gpurun_out/c5_${L}_hard.json 2> gpurun_out/c5_${L}_hard.err || tail -3 gpurun_out/c5_${L}_hard.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/c5_*.json")):
    try:
        r = json.load(open(f)); ro = r["roofline"]
        print("%-40s value %8.0f  ms/step %.2f  avg %.4f ms  ps/unit %.2f" % (f, r["value"], r["ms_per_step"], ro["avg_launch_ms"], 1e9 * ro["avg_launch_ms"] / ro["units_per_launch"]))
    except Exception as e:
        print(f, "FAILED", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c5_order -o o -- python - <<PY > /dev/null 2> $R/gpurun_out/c5_order.err
import sys; sys.path.insert(0, "$R")
import numpy as np
from elimaloc_amd.registration import Context, Scan
c = Context(0)
rng = np.random.default_rng(1)
xyz = (rng.standard_normal((131072, 3)) * np.array([30.0, 30.0, 2.0])).astype(np.float32)
for _ in range(50):
    s = Scan(c, xyz)
PY
python - <<PY
import sqlite3, glob
for db in glob.glob("$R/gpurun_out/c5_order/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    for r in c.execute("select name, count(*), avg(end-start), min(end-start) from kernels group by name order by 3 desc"):
        print("%-90s n=%d avg %.1f us min %.1f us" % (r[0][:90], r[1], r[2]/1e3, r[3]/1e3))
PY
rm -rf $R/gpurun_out/c5_order
