#!/bin/bash
# BASELINE config 4's sizes on ONE GPU (262 144-point scans, 50 M-point map), round-4 build
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
for m in ${C4_METHODS:-2 0}; do
  timeout 900 python bench.py --no-cpu --no-extras --method $m --scan-points 262144 --map-points 50000000 --batch 128 --slots 32 --steps 5 --warmup 1 > gpurun_out/c4size_m$m.json 2> gpurun_out/c4size_m$m.err || tail -3 gpurun_out/c4size_m$m.err
  python - gpurun_out/c4size_m$m.json <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); f = r["roofline"]
print(r["config"]["workload"][:60], "value %.0f" % r["value"], "iters %.2f" % r["config"]["iterations_mean"], "launch %.4f ms" % f["avg_launch_ms"], "map_build %.0f s" % r["config"]["map_build_s"], flush=True)
PY
done
