#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_3.pytest 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r3_3.pytest | tail -8
for M in 1 2 3; do
timeout 900 python bench.py --no-cpu --no-extras --batch 1024 --method $M > gpurun_out/r3_3_m$M.json 2> gpurun_out/r3_3_m$M.err || tail -20 gpurun_out/r3_3_m$M.err
python - $M <<'PY'
import json, sys
try:
    r = json.load(open("gpurun_out/r3_3_m%s.json" % sys.argv[1]))
    f = r["roofline"]
    print("method", sys.argv[1], "value %.0f reg/s  iters %.3f  launch %.4f ms  timed %.2fs" % (r["value"], r["config"]["iterations_mean"], f["avg_launch_ms"], f["timed_region_s"]))
except Exception as e:
    print("bench parse failed", e)
PY
done
