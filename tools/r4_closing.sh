#!/bin/bash
# closing verification of the round-4 head: gpu suite, smoke, the driver's default bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu > gpurun_out/k.pytest 2>&1; tail -3 gpurun_out/k.pytest; grep -n "^FAILED\|^E  " gpurun_out/k.pytest | head -20
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 400 python bench.py > gpurun_out/k_default.json 2> gpurun_out/k_default.err; python -c "
import json; r=json.load(open('gpurun_out/k_default.json')); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['bound'], r['cpu_baseline']['value'])"
