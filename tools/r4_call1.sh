#!/bin/bash
# round-4 GPU call 1: VALU probe, launcher checks on the GPU box, counter passes at the default operating point, hard-guess "before"
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
tools/probes/run_valu_probe.sh > gpurun_out/valu_probe.log 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_bench_launcher.py -x -q -m "gpu or not gpu" > gpurun_out/c1.pytest 2>&1; tail -3 gpurun_out/c1.pytest
python bench.py --gpus 2 > gpurun_out/c1_gpus2.out 2> gpurun_out/c1_gpus2.err; echo "bench --gpus 2 rc=$?" | tee -a gpurun_out/c1_gpus2.err
python bench.py > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err; tail -c 600 gpurun_out/c1_bench.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu > gpurun_out/c1_dist1.json 2> gpurun_out/c1_dist1.err; tail -c 300 gpurun_out/c1_dist1.json
PROF_NO_FINAL=1 tools/collect_profiles.sh r04_default 2>&1 | tail -2
BENCH_ARGS="--guess hard" PROF_STALLS=1 PROF_NO_FINAL=1 PROF_STEPS=2 tools/collect_profiles.sh r04_hard_before 2>&1 | tail -2
