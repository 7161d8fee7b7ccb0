#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
{
python tools/lat1.py 2>&1 | tail -1
ELM_FUSED_REDUCE=1 python tools/lat1.py 2>&1 | tail -1
ELM_GRAPH=1 python tools/lat1.py 2>&1 | tail -1
ELM_FUSED_REDUCE=1 ELM_GRAPH=1 python tools/lat1.py 2>&1 | tail -1
} > gpurun_out/lat1.txt 2>&1
cat gpurun_out/lat1.txt
