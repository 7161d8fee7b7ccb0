#!/bin/bash
# counter passes of the VGICP / AVGICP kernels after the fused compact pairs (tag i)
TAG=${1:-i}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
BENCH_ARGS="--method 2" PROF_NO_FINAL=1 PROF_STALLS=1 tools/collect_profiles.sh r04${TAG}_vgicp 2>&1 | tail -1 | cut -c1-200
BENCH_ARGS="--method 3" PROF_NO_FINAL=1 PROF_STALLS=1 tools/collect_profiles.sh r04${TAG}_avgicp 2>&1 | tail -1 | cut -c1-200
