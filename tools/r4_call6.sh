#!/bin/bash
# round-4 GPU call 6: parity of the shipped build, resident-workgroup A/B, single-registration timeline, ordering kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/c6.pytest 2>&1; tail -4 gpurun_out/c6.pytest
for P in 0 1792 3584 7168; do
  ELM_PERSIST_WGS=$P python bench.py --no-cpu --no-extras > gpurun_out/c6_persist$P.json 2> gpurun_out/c6_persist$P.err || tail -3 gpurun_out/c6_persist$P.err
done
ELM_PERSIST_WGS=1792 python bench.py --no-cpu --no-extras --method 1 > gpurun_out/c6_gicp_persist1792.json 2> /dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/c6_*.json")):
    try:
        r = json.load(open(f)); ro = r["roofline"]
        print("%-40s value %8.0f  ms/step %.2f  avg %.4f ms" % (f, r["value"], r["ms_per_step"], ro["avg_launch_ms"]))
    except Exception as e:
        print(f, "FAILED", e)
PY
tools/trace_single.sh 2>&1 | tail -45
