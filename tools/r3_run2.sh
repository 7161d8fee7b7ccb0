#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_2.pytest 2>&1; grep -E "passed|failed|Error" gpurun_out/r3_2.pytest | tail -5
timeout 900 python bench.py --no-cpu --batch 1024 --hostfed-batch 512 > gpurun_out/r3_2_bench.json 2> gpurun_out/r3_2_bench.err || tail -20 gpurun_out/r3_2_bench.err
python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r3_2_bench.json"))
    f = r["roofline"]
    print("value %.0f reg/s  iters %.3f  launch %.4f ms  timed %.2fs" % (r["value"], r["config"]["iterations_mean"], f["avg_launch_ms"], f["timed_region_s"]))
    print("host_fed", r["host_fed"]["value"], r["host_fed"]["pcie_achieved_gbs"], r["host_fed"]["bit_identical_to_resident"])
    print("reference_api", json.dumps(r.get("reference_api")))
    print("lat", r["config"].get("latency_ms_batch1"))
except Exception as e:
    print("bench parse failed", e)
PY
timeout 600 python tools/stream_c5.py --native --scans 30 > gpurun_out/r3_2_c5.json 2> gpurun_out/r3_2_c5.err || tail -5 gpurun_out/r3_2_c5.err
cat gpurun_out/r3_2_c5.json
