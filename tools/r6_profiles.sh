#!/bin/bash
# round 6: rocprofv3 kernel-trace + counter passes AT THE OPERATING POINT of every timed leg of the default bench line
#   tools/r6_profiles.sh [legs...]     legs: p2p gicp vgicp avgicp hard c4 field0 field1 field2 field3   (default: all)
# -> gpurun_out/prof_r06_<leg>/{kernel_stats.csv,pmc.json,bench_trace.json}; copy with tools/merge_pmc.py afterwards
set -u
LEGS=${@:-p2p gicp vgicp avgicp hard c4 field0 field1 field2 field3}
export PROF_NO_FINAL=1
for leg in $LEGS; do
  case $leg in
    p2p)    BENCH_ARGS="" PROF_BATCH=4096 tools/collect_profiles.sh r06_p2p ;;
    gicp)   BENCH_ARGS="--method 1" PROF_BATCH=4096 tools/collect_profiles.sh r06_gicp ;;
    vgicp)  BENCH_ARGS="--method 2" PROF_BATCH=4096 tools/collect_profiles.sh r06_vgicp ;;
    avgicp) BENCH_ARGS="--method 3" PROF_BATCH=4096 tools/collect_profiles.sh r06_avgicp ;;
    hard)   BENCH_ARGS="--guess hard" PROF_BATCH=4096 tools/collect_profiles.sh r06_hard ;;
    c4)     BENCH_ARGS="--method 2 --scan-points 32768 --shard-of 8 --map-points 50000000 --slots 256" PROF_BATCH=2048 tools/collect_profiles.sh r06_c4shard ;;
    field0|field1|field2|field3)  # the field world's legs: 1024 registrations per method (bench.py's `field_world`)
            BENCH_ARGS="--world field --method ${leg#field}" PROF_BATCH=1024 tools/collect_profiles.sh r06_field_m${leg#field} ;;
  esac | tail -1 | cut -c1-400
done
