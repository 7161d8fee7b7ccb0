#!/bin/bash
# AVGICP on a map WITH flagged voxels (200 000-point world, seed 1001): fused walk + fix-up launch (default) against the nine-entry walk
# with its in-line fallback (ELM_AVG_FIXUP=0); then the gpu suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
one() { local T=$1; shift
  timeout 300 python bench.py --no-cpu --no-extras --method 3 --map-points 200000 "$@" > gpurun_out/fx_$T.json 2> gpurun_out/fx_$T.err || tail -3 gpurun_out/fx_$T.err
  python - $T gpurun_out/fx_$T.json <<'PY'
import json, sys
r = json.load(open(sys.argv[2])); f = r["roofline"]
print("%-8s %8.0f reg/s  launch %.4f ms  launches %d" % (sys.argv[1], r["value"], f["avg_launch_ms"], f["launches"]), flush=True)
PY
}
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "flagged or rank_deficient or fused" 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2; do ELM_AVG_FIXUP=0 one inline$i; one fixup$i; done
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/z.pytest 2>&1; grep -E "passed|failed|error" gpurun_out/z.pytest | tail -2; grep -n "^FAILED" gpurun_out/z.pytest | head
