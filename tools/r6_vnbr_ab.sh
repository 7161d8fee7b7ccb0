#!/bin/bash
# round 6: the voxel-mean kernels after de-replicating the per-voxel record (run on the GPU box through gpurun)
#   tools/r6_vnbr_ab.sh <tag>  -> gpurun_out/<tag>/{gputest.log,vgicp.json,c4.json,avgicp.json}
TAG=${1:-r06b}
O=gpurun_out/$TAG
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; tail -3 $O/gputest.log
python bench.py --method 2 --no-cpu --no-extras --steps 5 --warmup 1 > $O/vgicp.json 2> $O/vgicp.err
python bench.py --method 2 --scan-points 32768 --shard-of 8 --map-points 50000000 --slots 256 --batch 2048 --no-cpu --no-extras --steps 5 --warmup 1 > $O/c4.json 2> $O/c4.err
python bench.py --method 3 --no-cpu --no-extras --steps 5 --warmup 1 > $O/avgicp.json 2> $O/avgicp.err
for f in vgicp c4 avgicp; do python - $O/$f.json <<'PY'
import json,sys
l=json.load(open(sys.argv[1])); r=l["roofline"]
print(sys.argv[1], "value", l["value"], "avg_launch_ms", r["avg_launch_ms"], "ps/unit", r.get("this_run_ps_per_unit") or 1e9*r["avg_launch_ms"]/r["units_per_launch"], "index_bytes", r["index_bytes"])
PY
done
