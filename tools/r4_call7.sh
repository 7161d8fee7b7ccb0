#!/bin/bash
# round-4 GPU call 7: parity, distributed refill forms on the one-rank RCCL path, latency with the new early-stop placement, ordering kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/c7.pytest 2>&1; tail -4 gpurun_out/c7.pytest
for F in solve kernel solve kernel; do
  i=$((i+1))
  ELM_DIST_REFILL=$F python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$i bench.py --gpus 1 --no-cpu --no-extras > gpurun_out/c7_dist1_${F}_$i.json 2> gpurun_out/c7_dist1_${F}_$i.err
done
python bench.py --no-cpu --hostfed-batch 0 > gpurun_out/c7_lat.json 2> gpurun_out/c7_lat.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/c7_*.json")):
    try:
        r = json.load(open(f)); ro = r["roofline"]; ra = r.get("reference_api", {})
        print("%-38s value %8.0f  ms/step %.2f  launches %d avg %.4f ms | lat1 %s refapi %s" % (f, r["value"], r["ms_per_step"], ro["launches"], ro["avg_launch_ms"], r["config"].get("latency_ms_batch1"), ra.get("registrations_per_s")))
    except Exception as e:
        print(f, "FAILED", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c7_order -o o -- python - <<PY > /dev/null 2> $R/gpurun_out/c7_order.err
import sys; sys.path.insert(0, "$R")
import numpy as np
from elimaloc_amd.registration import Context, Scan
c = Context(0)
rng = np.random.default_rng(1)
xyz = (rng.standard_normal((131072, 3)) * np.array([30.0, 30.0, 2.0])).astype(np.float32)
for _ in range(50):
    s = Scan(c, xyz)
PY
python - <<PY | tee $R/gpurun_out/c7_order.txt
import sqlite3, glob
for db in glob.glob("$R/gpurun_out/c7_order/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    for r in c.execute("select name, count(*), avg(end-start), min(end-start) from kernels group by name order by 3 desc"):
        print("%-90s n=%d avg %.1f us min %.1f us" % (r[0][:90], r[1], r[2]/1e3, r[3]/1e3))
    rows = c.execute("select start, end, name from kernels order by start").fetchall()
    import statistics
    spans = []
    for i in range(0, len(rows) - 3):
        if "k_order_rank" in rows[i][2] and "k_order_scatter" in rows[i+3][2]:
            spans.append((rows[i+3][1] - rows[i][0]) / 1e3)
    if spans: print("k_order_rank start -> k_order_scatter end: median %.1f us (n = %d; under the tracer)" % (statistics.median(spans), len(spans)))
PY
rm -rf $R/gpurun_out/c7_order
