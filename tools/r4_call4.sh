#!/bin/bash
# round-4 GPU call 4: gated previous-winner build, graph replay of single registrations (latency, reference API, A/B), wide ordering, full default line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/c4.pytest 2>&1; tail -4 gpurun_out/c4.pytest
python bench.py > gpurun_out/c4_bench.json 2> gpurun_out/c4_bench.err || tail -5 gpurun_out/c4_bench.err
ELM_GRAPH=0 python bench.py --no-cpu --hostfed-batch 0 > gpurun_out/c4_nograph.json 2> gpurun_out/c4_nograph.err
ELM_FUSED_REDUCE=1 python bench.py --no-cpu --hostfed-batch 0 > gpurun_out/c4_fused.json 2> gpurun_out/c4_fused.err
python bench.py --no-cpu --no-extras --guess hard --steps 6 > gpurun_out/c4_hard.json 2> gpurun_out/c4_hard.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/c4_*.json")):
    try:
        r = json.load(open(f)); ro = r["roofline"]
        ra = r.get("reference_api", {})
        print("%-30s value %8.0f ms/step %.2f avg %.4f ms | lat1 %s ms | refapi %s /s pinned %s /s | hard %s | hostfed %s | frac %.3f %s" % (
            f, r["value"], r["ms_per_step"], ro["avg_launch_ms"], r["config"].get("latency_ms_batch1"), ra.get("registrations_per_s"),
            ra.get("page_locked_source", {}).get("registrations_per_s"), r.get("hard_guess", {}).get("value"), r.get("host_fed", {}).get("value"), ro["frac"], ro["bound"]))
    except Exception as e:
        print(f, "FAILED", e)
PY
tools/default_trace.sh > gpurun_out/c4_trace.log 2>&1; head -12 gpurun_out/prof_default/kernel_stats.csv | cut -c1-160
tools/c5_c_harness.sh > gpurun_out/c4_c5.log 2>&1; tail -6 gpurun_out/c4_c5.log
