#!/bin/bash
# A/B: GICP / VGICP compact pair trims (t2: weight by div_close, fused 0.8 w + 0.2, the fitness term from the fused n . e) against the build before (pf); gpu suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
one() { local L=$1 T=$2; shift 2
  ELM_LIB=$PWD/build_ab/lib_$L.so python bench.py --no-cpu --no-extras "$@" > gpurun_out/z_${L}_$T.json 2> gpurun_out/z_${L}_$T.err || tail -3 gpurun_out/z_${L}_$T.err
  python - $L $T gpurun_out/z_${L}_$T.json <<'PY'
import json, sys
r = json.load(open(sys.argv[3])); f = r["roofline"]
print("%-6s %-7s %8.0f reg/s  launch %.4f ms" % (sys.argv[1], sys.argv[2], r["value"], f["avg_launch_ms"]), flush=True)
PY
}
for L in pf t2 pf t2 pf t2; do one $L gicp --method 1; done
for L in pf t2 pf t2; do one $L vg --method 2; done
python -m pytest tests -q -m gpu > gpurun_out/z.pytest 2>&1; tail -3 gpurun_out/z.pytest; grep -n "^FAILED" gpurun_out/z.pytest | head
