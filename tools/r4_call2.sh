#!/bin/bash
# round-4 GPU call 2: probe v2, full gpu suite on the STATS / half-set build, half-set A/B on one rank and on the one-rank collective path
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
tools/probes/run_valu_probe.sh > gpurun_out/valu_probe.log 2>&1
python -m pytest tests -x -q -m gpu > gpurun_out/c2.pytest 2>&1; tail -4 gpurun_out/c2.pytest
b() { # tag env... -- args
  local tag=$1; shift
  env "$@" python bench.py --no-cpu --no-extras > gpurun_out/c2_$tag.json 2> gpurun_out/c2_$tag.err || tail -3 gpurun_out/c2_$tag.err
}
b h0 ELM_HALF_SETS=0
b h1 ELM_HALF_SETS=1
ELM_HALF_SETS=1 python bench.py --no-cpu --no-extras --slots 512 > gpurun_out/c2_h1_s512.json 2> gpurun_out/c2_h1_s512.err
ELM_HALF_SETS=1 python bench.py --no-cpu --no-extras --slots 384 > gpurun_out/c2_h1_s384.json 2> gpurun_out/c2_h1_s384.err
for h in 0 1; do
  ELM_HALF_SETS=$h python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$h bench.py --gpus 1 --no-cpu --no-extras > gpurun_out/c2_dist1_h$h.json 2> gpurun_out/c2_dist1_h$h.err
done
ELM_HALF_SETS=1 ELM_FUSED_REDUCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --no-cpu --no-extras > gpurun_out/c2_dist1_h1_fused.json 2> gpurun_out/c2_dist1_h1_fused.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/c2_*.json")):
    try:
        r = json.load(open(f)); ro = r["roofline"]
        print("%-40s value %8.0f  ms/step %.2f  launches %d  avg %.4f ms  acc/step %.2f  solve/step %.2f  rccl %s" % (f, r["value"], r["ms_per_step"], ro["launches"], ro["avg_launch_ms"], ro["accumulate_ms_per_step"], ro["solve_ms_per_step"], r.get("rccl_ranks")))
    except Exception as e:
        print(f, "FAILED", e)
PY
