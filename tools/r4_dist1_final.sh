#!/bin/bash
# the driver's N > 1 launch form with one rank (RCCL communicator of one, collective path) on the final build
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu > gpurun_out/k4_dist1.json 2> gpurun_out/k4_dist1.err
python -c "
import json; r=json.load(open('gpurun_out/k4_dist1.json')); print(r['value'], r['n_gpus'], r.get('rccl_ranks'), r['ms_per_step'], r['config']['parallelism'])" || tail -20 gpurun_out/k4_dist1.err
timeout 100 python bench.py --gpus 2 > gpurun_out/k4_gpus2.txt 2>&1; echo "bench.py --gpus 2 on a 1-GPU box: rc=$?"; tail -2 gpurun_out/k4_gpus2.txt
