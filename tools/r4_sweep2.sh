#!/bin/bash
# second re-tune sweep: 8-wave caps (64 VGPRs) for the grid and voxel-list kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
one() { # lib tag args...
  local L=$1 T=$2; shift 2
  ELM_LIB=$PWD/build_ab/lib_$L.so python bench.py --no-cpu --no-extras "$@" > gpurun_out/s2_${L}_$T.json 2> gpurun_out/s2_${L}_$T.err || tail -3 gpurun_out/s2_${L}_$T.err
  python - $L $T gpurun_out/s2_${L}_$T.json <<'PY'
import json, sys
r = json.load(open(sys.argv[3])); f = r["roofline"]
print("%-8s %-7s %8.0f reg/s  launch %.4f ms" % (sys.argv[1], sys.argv[2], r["value"], f["avg_launch_ms"]), flush=True)
PY
}
for L in cur w8g8 cur w8g8; do one $L easy; done
for L in cur w8g8; do one $L hard --guess hard --steps 6; done
for L in cur gw7 w8g8 w8g7; do one $L gicp --method 1; done
for L in cur v8 cur v8; do one $L vgicp --method 2; done
for L in cur v8; do one $L avgicp --method 3; done
ELM_LIB=$PWD/build_ab/lib_w8g7.so python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu > gpurun_out/s2_w8g7.pytest 2>&1; tail -2 gpurun_out/s2_w8g7.pytest
