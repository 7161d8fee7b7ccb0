#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
mkdir -p gpurun_out/r5s
python -m pytest tests/test_correspondences.py tests/test_shim_compile.py -x -q -m gpu > gpurun_out/r5s/corr.txt 2>&1; tail -15 gpurun_out/r5s/corr.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size_properties" > gpurun_out/r5s/full.txt 2>&1; tail -15 gpurun_out/r5s/full.txt
