#!/bin/bash
# round 6: A/B of two prebuilt libraries on the voxel-mean kernels (run on the GPU box): ELM_LIB=build_ab/libA.so against the in-tree build
#   tools/r6_ab.sh <tag>   -> gpurun_out/<tag>/ab.txt
O=gpurun_out/${1:-r06ab}
mkdir -p $O
run() { # name lib args...
  local name=$1 lib=$2; shift 2
  ELM_LIB=$lib python bench.py "$@" --no-cpu --no-extras --steps 6 --warmup 2 > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY' >> $O/ab.txt
import json,sys
l=json.load(open(sys.argv[1])); r=l["roofline"]
print(f"{sys.argv[2]:28s} value {l['value']:10.1f}  avg_launch_ms {r['avg_launch_ms']:.5f}  ps/unit {1e9*r['avg_launch_ms']/r['units_per_launch']:.3f}")
PY
}
A=$(pwd)/build_ab/libA.so; B=$(pwd)/elimaloc_amd/libelimaloc_hip.so
for rep in 1 2; do
  run vgicp_A$rep $A --method 2
  run vgicp_B$rep $B --method 2
  run c4_A$rep $A --method 2 --scan-points 32768 --shard-of 8 --map-points 50000000 --slots 256 --batch 2048
  run c4_B$rep $B --method 2 --scan-points 32768 --shard-of 8 --map-points 50000000 --slots 256 --batch 2048
done
run fieldv_A $A --method 2 --world field --batch 1024
run fieldv_B $B --method 2 --world field --batch 1024
cat $O/ab.txt
