#!/bin/bash
# round 6: the device-group path on the one-GPU box (two ranks on device 0, host-memory exchange) -- tests + bench --single-process
O=gpurun_out/${1:-r06h}
mkdir -p $O
timeout 600 python -m pytest tests/test_group.py tests/test_shim_compile.py -x -q -m gpu > $O/group_tests.log 2>&1; tail -5 $O/group_tests.log
timeout 600 python bench.py --gpus 2 --single-process --devices 0,0 --batch 512 --steps 3 --warmup 1 --dump-poses $O/poses_group.npz > $O/bench_group.json 2> $O/bench_group.err; tail -c 600 $O/bench_group.json; tail -3 $O/bench_group.err | cut -c1-300
