#!/bin/bash
# slots per GPU at the final kernel (bench parameter; 256 is the default)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
{
for sl in 256 384 512 192 256; do timeout 150 python bench.py --slots $sl --no-cpu --no-extras > gpurun_out/sl_$sl.json 2> gpurun_out/sl_$sl.err; python -c "
import json; r=json.load(open('gpurun_out/sl_$sl.json')); print('slots $sl', round(r['value']), 'reg/s  launch', round(r['roofline']['avg_launch_ms'],4), 'ms  launches', r['roofline']['launches'])"; done
} > gpurun_out/r4_slots.txt 2>&1
cat gpurun_out/r4_slots.txt
