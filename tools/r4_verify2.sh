#!/bin/bash
# verification of the refill-launch rework (the solve saves + counts, the refill assigns by prefix count) and the AVGICP NaN fix
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/w.pytest 2>&1; tail -4 gpurun_out/w.pytest
for i in 1 2; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2954$i bench.py --gpus 1 --no-cpu --no-extras > gpurun_out/w_dist1_$i.json 2> gpurun_out/w_dist1_$i.err
  python bench.py --no-cpu --no-extras > gpurun_out/w_plain_$i.json 2> gpurun_out/w_plain_$i.err
done
python bench.py --method 3 --no-cpu --no-extras > gpurun_out/w_m3.json 2> gpurun_out/w_m3.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/w_*.json")):
    try:
        r = json.load(open(f)); ro = r["roofline"]
        print("%-28s value %8.0f ms/step %.2f launches %d avg %.4f ms solve/step %.2f rccl %s" % (f, r["value"], r["ms_per_step"], ro["launches"], ro["avg_launch_ms"], ro["solve_ms_per_step"], r.get("rccl_ranks")))
    except Exception as e:
        print(f, "FAILED", e)
PY
tools/trace_gaps_dist1.sh > gpurun_out/w_gaps.log 2>&1; tail -17 gpurun_out/w_gaps.log
