#!/bin/bash
# closing call of round 4: gpu suite, default bench line, kernel trace of the default command, a long fuzz campaign on fresh seeds
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu > gpurun_out/k3.pytest 2>&1; grep -E "passed|failed|^FAILED|^E  " gpurun_out/k3.pytest | head -20
timeout 400 python bench.py > gpurun_out/k3_default.json 2> gpurun_out/k3_default.err; python -c "
import json; r=json.load(open('gpurun_out/k3_default.json')); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['cpu_baseline']['value'])"
bash tools/default_trace.sh 2>&1 | tail -4
cd $R
{
for s0 in 1000000 1010000 1020000 1030000; do echo "--seed0 $s0 (6000 cases)"; timeout 240 python tools/fuzz_parity.py --cases 6000 --seed0 $s0 2>&1 | grep -E "singular-system|MISMATCH|cases agree|pose error"; done
echo "ELM_GRID=tiled --seed0 1040000 (3000)"; ELM_GRID=tiled timeout 200 python tools/fuzz_parity.py --cases 3000 --seed0 1040000 2>&1 | grep -E "singular-system|MISMATCH|cases agree|pose error"
echo "ELM_PAIR_NINE=1 --seed0 1050000 (3000)"; ELM_PAIR_NINE=1 timeout 200 python tools/fuzz_parity.py --cases 3000 --seed0 1050000 2>&1 | grep -E "singular-system|MISMATCH|cases agree|pose error"
} > gpurun_out/r4_soak6.txt 2>&1
cat gpurun_out/r4_soak6.txt
