#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r3_5.pytest 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r3_5.pytest | tail -6
ELM_FUSED_REDUCE=0 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_hostfed.py -m gpu -x -q > gpurun_out/r3_5u.pytest 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r3_5u.pytest | tail -4
for F in 1 0 1 0; do
ELM_FUSED_REDUCE=$F timeout 900 python bench.py --no-cpu --batch 2048 --hostfed-batch 0 > /tmp/b.json 2> /tmp/b.err || tail -5 /tmp/b.err
python - $F <<'PY'
import json, sys
r = json.load(open("/tmp/b.json")); f = r["roofline"]
print("fused", sys.argv[1], "value %.0f reg/s  launch %.4f ms  acc/step %.2f solve/step %.2f  lat %.4f  refapi %.0f hard %.0f" % (r["value"], f["avg_launch_ms"], f["accumulate_ms_per_step"], f["solve_ms_per_step"], r["config"].get("latency_ms_batch1", 0), r["reference_api"]["registrations_per_s"], r["hard_guess"]["value"]), flush=True)
PY
done
