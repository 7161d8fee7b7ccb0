#!/bin/bash
# closing verification of the build with the asymmetric-covariance routing: gpu suite, fresh fuzz blocks, default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu > gpurun_out/k2.pytest 2>&1; grep -E "passed|failed|^FAILED|^E  " gpurun_out/k2.pytest | head -20
{
for s0 in 900000 910000; do echo "--seed0 $s0 (4000 cases)"; timeout 300 python tools/fuzz_parity.py --cases 4000 --seed0 $s0 2>&1 | grep -E "singular-system|MISMATCH|cases agree|pose error"; done
echo "ELM_GRID=tiled --seed0 920000 (1500)"; ELM_GRID=tiled timeout 200 python tools/fuzz_parity.py --cases 1500 --seed0 920000 2>&1 | grep -E "singular-system|MISMATCH|cases agree|pose error"
echo "--radar 1.0 --seed0 930000 (800)"; timeout 200 python tools/fuzz_parity.py --cases 800 --seed0 930000 --radar 1.0 2>&1 | grep -E "singular-system|MISMATCH|cases agree|singular-metric radar"
} > gpurun_out/r4_soak5.txt 2>&1
cat gpurun_out/r4_soak5.txt
timeout 400 python bench.py > gpurun_out/k2_default.json 2> gpurun_out/k2_default.err; python -c "
import json; r=json.load(open('gpurun_out/k2_default.json')); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['config']['map_layout_flags'], r['cpu_baseline']['value'])"
