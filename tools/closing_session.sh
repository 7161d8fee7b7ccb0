#!/bin/bash
# Run ON THE MI355X BOX: the closing sequence of a round -- counter passes of every bench leg, merged into profiles/pmc_latest.json ON THE BOX,
# then the driver's default command (whose rooflines read those passes), its kernel trace, the one-rank RCCL line and the gpu suite.
#   tools/gpu.sh tools/closing_session.sh <tag> [legs...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
TAG=${1:-r05z}; shift
tools/r5_profiles.sh "$@"
python tools/merge_pmc.py gpurun_out/prof_r05_* > gpurun_out/merge.txt 2>&1; tail -1 gpurun_out/merge.txt | cut -c1-300
cp profiles/pmc_latest.json gpurun_out/pmc_latest.json
tools/gpu_session.sh $TAG bench dist1 suite
tools/default_trace.sh > gpurun_out/$TAG/default_trace.txt 2>&1; tail -3 gpurun_out/$TAG/default_trace.txt
