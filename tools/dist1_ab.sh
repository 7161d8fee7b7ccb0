#!/bin/bash
# one-rank collective path (RCCL communicator formed) with and without the fused reduction
for F in 0 1 0 1; do
  ELM_FUSED_REDUCE=$F python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$F bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu --no-extras > /tmp/d.json 2> /tmp/d.err || tail -3 /tmp/d.err
  python -c "
import json; r=json.load(open('/tmp/d.json')); print('fused=$F', round(r['value']), r['roofline']['avg_launch_ms'])"
done
