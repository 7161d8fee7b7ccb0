#!/bin/bash
# the routing of maps with asymmetric flagged covariances (layout bits 7 / 8): its tests, the fuzz block that holds case 813687, method lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
{
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "asymmetric or flagged or compact" 2>&1 | tail -15
echo "--seed0 810000 (5000 cases, default routing)"; timeout 420 python tools/fuzz_parity.py --cases 5000 --seed0 810000 2>&1 | grep -E "singular-system|MISMATCH|cases agree|pose error"
for m in 1 2 3; do timeout 200 python bench.py --method $m --no-cpu --no-extras > gpurun_out/asym_m$m.json 2> gpurun_out/asym_m$m.err; python -c "
import json; r=json.load(open('gpurun_out/asym_m$m.json')); print('method $m', round(r['value']), r['roofline']['avg_launch_ms'])"; done
} > gpurun_out/r4_asym.txt 2>&1
cat gpurun_out/r4_asym.txt
