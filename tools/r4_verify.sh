#!/bin/bash
# round-4 verification call: the full gpu suite on the committed tree, a second sample of the default and per-method lines, the multi-rank
# launch sequence on the one-rank RCCL path
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/v.pytest 2>&1; tail -4 gpurun_out/v.pytest
python bench.py --no-cpu --no-extras > gpurun_out/v_default.json 2> gpurun_out/v_default.err
for m in 1 2 3; do python bench.py --method $m --no-cpu --no-extras > gpurun_out/v_m$m.json 2> gpurun_out/v_m$m.err; done
python bench.py --guess hard --steps 6 --no-cpu --no-extras > gpurun_out/v_hard.json 2> gpurun_out/v_hard.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/v_*.json")):
    try:
        r = json.load(open(f)); ro = r["roofline"]
        print("%-28s value %8.0f ms/step %.2f launches %d avg %.4f ms | frac %.3f %s (%s)" % (f, r["value"], r["ms_per_step"], ro["launches"], ro["avg_launch_ms"], ro["frac"], ro["bound"], ro.get("counters", {}).get("source")))
    except Exception as e:
        print(f, "FAILED", e)
PY
tools/trace_gaps_dist1.sh > gpurun_out/v_gaps.log 2>&1; tail -24 gpurun_out/v_gaps.log
