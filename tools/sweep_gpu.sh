#!/bin/bash
# Run ON THE MI355X BOX: rebuild the library with each EXTRA flag set and run a short bench (developer A/B loop).
#   tools/sweep_gpu.sh "<bench args>" "<EXTRA 1>" "<EXTRA 2>" ...
BARGS=$1; shift
for X in "$@"; do
  make -s -C elimaloc_amd/csrc clean; make -s -C elimaloc_amd/csrc EXTRA="$X" 2>&1 | grep -E "error" | head
  timeout 300 python bench.py --no-cpu --no-extras $BARGS > gpurun_out/sweep.json 2> gpurun_out/sweep.err || tail -5 gpurun_out/sweep.err
  python - "$X" <<PY
import json, sys
r = json.load(open("gpurun_out/sweep.json")); f = r["roofline"]
print("%-44s value %.0f reg/s  iters %.4f  launch %.4f ms  acc/step %.3f  solve/step %.3f" % (sys.argv[1], r["value"], r["config"]["iterations_mean"], f["avg_launch_ms"], f["accumulate_ms_per_step"], f["solve_ms_per_step"]))
PY
done
make -s -C elimaloc_amd/csrc clean; make -s -C elimaloc_amd/csrc
