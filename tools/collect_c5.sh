#!/bin/bash
# Run ON THE MI355X BOX: per-kernel trace of the config-5 node callback (deskew + downsample + VGICP + EKF, 131 072-point scans vs the
# 10 M-point map) -> gpurun_out/prof_<tag>/{kernel_stats.csv, c5.json}
TAG=${1:-r02_c5}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o b -- python $R/tools/stream_c5.py --native --scans 30 "$@" > $OUT/c5_trace.json 2> $OUT/trace.err
python $R/tools/rocpd_stats.py $(find $OUT/trace -name "*.db" | head -1) $OUT/kernel_stats.csv > /dev/null
python $R/tools/stream_c5.py --native --scans 40 "$@" > $OUT/c5.json 2> $OUT/c5.err
python $R/tools/stream_c5.py --native --scans 40 --ds 0.05 "$@" > $OUT/c5_full.json 2>> $OUT/c5.err
head -12 $OUT/kernel_stats.csv | cut -c1-160; cat $OUT/c5.json
