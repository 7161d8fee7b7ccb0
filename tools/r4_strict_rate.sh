#!/bin/bash
# what the per-pair (strict) path costs at the BASELINE sizes: the routed path of maps with asymmetric flagged covariances, forced here by ELM_STRICT_PAIRS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
{
for m in 1 2 3; do ELM_STRICT_PAIRS=1 timeout 200 python bench.py --method $m --batch 256 --steps 2 --warmup 1 --no-cpu --no-extras > gpurun_out/strict_m$m.json 2> gpurun_out/strict_m$m.err; python -c "
import json; r=json.load(open('gpurun_out/strict_m$m.json')); print('ELM_STRICT_PAIRS=1 method $m', round(r['value']), 'registrations/s', r['ms_per_step'], 'ms per step of 256', r['config']['iterations_mean'])" || tail -3 gpurun_out/strict_m$m.err; done
} > gpurun_out/r4_strict_rate.txt 2>&1
cat gpurun_out/r4_strict_rate.txt
