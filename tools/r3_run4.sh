#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r3_4.pytest 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r3_4.pytest | tail -8
# tiled vs dense on the bench workload
for G in dense tiled; do
ELM_GRID=$G timeout 900 python bench.py --no-cpu --no-extras --batch 1024 > gpurun_out/r3_4_$G.json 2> gpurun_out/r3_4_$G.err || tail -20 gpurun_out/r3_4_$G.err
python - $G <<'PY'
import json, sys
try:
    r = json.load(open("gpurun_out/r3_4_%s.json" % sys.argv[1]))
    f = r["roofline"]
    print(sys.argv[1], "value %.0f reg/s  iters %.3f  launch %.4f ms  index %.0f MB" % (r["value"], r["config"]["iterations_mean"], f["avg_launch_ms"], f["index_bytes"] / 1e6))
except Exception as e:
    print("bench parse failed", e)
PY
done
