#!/bin/bash
# round-3 profile collection: kernel trace + counter passes per method, then the default bench line with the CPU baseline
tools/collect_profiles.sh r03_p2p 2>&1 | tail -2
BENCH_ARGS="--method 1" tools/collect_profiles.sh r03_gicp 2>&1 | tail -2
BENCH_ARGS="--method 2" tools/collect_profiles.sh r03_vgicp 2>&1 | tail -2
BENCH_ARGS="--method 3" tools/collect_profiles.sh r03_avgicp 2>&1 | tail -2
