#!/bin/bash
# round 5, GPU session 4: full GPU suite (after the choose_path fix) + counter passes of every leg
set -u
mkdir -p gpurun_out/r5c4
O=gpurun_out/r5c4
( time python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1 ) 2> $O/pytest.time; tail -4 $O/pytest.txt; tail -3 $O/pytest.time
tools/r5_profiles.sh p2p gicp vgicp avgicp hard c4
