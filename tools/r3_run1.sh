#!/bin/bash
# round-3 GPU call 1: full gpu suite, default bench (no CPU leg), FETCH_SIZE gather probe
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_1.pytest 2>&1; tail -15 gpurun_out/r3_1.pytest
timeout 900 python bench.py --no-cpu > gpurun_out/r3_1_bench.json 2> gpurun_out/r3_1_bench.err || tail -20 gpurun_out/r3_1_bench.err
python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r3_1_bench.json"))
    f = r["roofline"]
    print("value %.0f reg/s  iters %.3f  launch %.4f ms  acc/step %.2f solve/step %.2f  timed %.2fs bound %s frac %.3f" % (r["value"], r["config"]["iterations_mean"], f["avg_launch_ms"], f["accumulate_ms_per_step"], f["solve_ms_per_step"], f["timed_region_s"], f["bound"], f["frac"]))
    print("host_fed", json.dumps(r.get("host_fed")))
    print("reference_api", json.dumps(r.get("reference_api")))
    print("hard", json.dumps(r.get("hard_guess")))
    print("gen", r["config"]["input_gen_s"], "map", r["config"]["map_build_s"], "lat", r["config"].get("latency_ms_batch1"))
except Exception as e:
    print("bench parse failed", e)
PY
timeout 600 tools/probes/run_gather_probe.sh 2>&1 | tail -8
