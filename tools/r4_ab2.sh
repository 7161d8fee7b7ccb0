#!/bin/bash
# A/B: columns requested in visiting order (new) against the permutation after the loads (cur = the shipped build before it)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/y.pytest 2>&1; tail -3 gpurun_out/y.pytest; grep -n "^FAILED" gpurun_out/y.pytest | head
one() { local L=$1 T=$2; shift 2
  ELM_LIB=$PWD/build_ab/lib_$L.so python bench.py --no-cpu --no-extras "$@" > gpurun_out/y_${L}_$T.json 2> gpurun_out/y_${L}_$T.err || tail -3 gpurun_out/y_${L}_$T.err
  python - $L $T gpurun_out/y_${L}_$T.json <<'PY'
import json, sys
r = json.load(open(sys.argv[3])); f = r["roofline"]
print("%-6s %-7s %8.0f reg/s  launch %.4f ms" % (sys.argv[1], sys.argv[2], r["value"], f["avg_launch_ms"]), flush=True)
PY
}
for L in cur new cur new; do one $L easy; done
for L in cur new; do one $L gicp --method 1; done
for L in cur new; do one $L hard --guess hard --steps 6; done
