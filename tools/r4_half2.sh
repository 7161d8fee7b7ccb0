#!/bin/bash
# half-set streams again, now with the 256-thread / 80-VGPR solve (k_solve<256>) that fits beside a running accumulate launch
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "half_set or refill_forms or stream_equals or two_ranks or stream_many" > gpurun_out/h.pytest 2>&1; tail -3 gpurun_out/h.pytest
python bench.py --no-cpu --no-extras > gpurun_out/h_h0_s256.json 2> gpurun_out/h_h0_s256.err
python bench.py --no-cpu --no-extras --slots 512 > gpurun_out/h_h0_s512.json 2> gpurun_out/h_h0_s512.err
ELM_HALF_SETS=1 python bench.py --no-cpu --no-extras > gpurun_out/h_h1_s256.json 2> gpurun_out/h_h1_s256.err
ELM_HALF_SETS=1 python bench.py --no-cpu --no-extras --slots 512 > gpurun_out/h_h1_s512.json 2> gpurun_out/h_h1_s512.err
ELM_HALF_SETS=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 --no-cpu --no-extras --slots 512 > gpurun_out/h_dist1_h1_s512.json 2> gpurun_out/h_dist1_h1_s512.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/h_*.json")):
    try:
        r = json.load(open(f)); ro = r["roofline"]
        print("%-34s value %8.0f ms/step %.2f launches %d avg %.4f ms acc/step %.2f other/step %.2f" % (f, r["value"], r["ms_per_step"], ro["launches"], ro["avg_launch_ms"], ro["accumulate_ms_per_step"], ro["solve_ms_per_step"]))
    except Exception as e:
        print(f, "FAILED", e)
PY
