#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/x.pytest 2>&1; tail -3 gpurun_out/x.pytest; grep -n "^FAILED\|^E  " gpurun_out/x.pytest | head -20
python bench.py --no-cpu --no-extras > gpurun_out/x_default.json 2> gpurun_out/x_default.err; python -c "
import json; r=json.load(open('gpurun_out/x_default.json')); print(r['value'], r['roofline']['frac'], r['roofline']['bound'], r['roofline']['counters'].get('source'))"
