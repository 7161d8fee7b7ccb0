#!/bin/bash
# Run ON THE MI355X BOX: parity subset + a short bench (developer loop).  tools/quick_gpu.sh <tag> [bench args]
TAG=${1:-q}; shift
python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu > gpurun_out/$TAG.pytest 2>&1; tail -3 gpurun_out/$TAG.pytest
python bench.py --no-cpu --no-extras "$@" > gpurun_out/$TAG.json 2> gpurun_out/$TAG.err || tail -5 gpurun_out/$TAG.err
python - <<PY
import json
r = json.load(open("gpurun_out/$TAG.json"))
f = r["roofline"]
print(open("gpurun_out/$TAG.pytest").read().strip().splitlines()[-1]); print("value %.0f reg/s  iters %.4f  launch %.4f ms  acc/step %.3f  solve/step %.3f  frac %.3f  tested %.2f" % (r["value"], r["config"]["iterations_mean"], f["avg_launch_ms"], f["accumulate_ms_per_step"], f["solve_ms_per_step"], f["frac"], f["tested_candidates_per_point"]))
PY
