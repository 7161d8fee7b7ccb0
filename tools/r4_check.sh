#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for i in 1 2; do python bench.py --method 3 --no-cpu --no-extras > gpurun_out/chk_m3_$i.json 2> gpurun_out/chk_m3_$i.err; python -c "
import json; r=json.load(open('gpurun_out/chk_m3_$i.json')); print('avgicp', round(r['value']), r['roofline']['avg_launch_ms'])"; done
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED" | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
