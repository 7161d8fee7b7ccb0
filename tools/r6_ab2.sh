#!/bin/bash
# A/B of two prebuilt libraries (build_ab/libA.so, libB.so) on the grid kernels: P2P easy + hard, GICP
O=gpurun_out/${1:-r06ab2}
mkdir -p $O
run() { local name=$1 lib=$2; shift 2
  ELM_LIB=$lib python bench.py "$@" --no-cpu --no-extras --steps 10 --warmup 3 > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY' >> $O/ab.txt
import json,sys
l=json.load(open(sys.argv[1])); r=l["roofline"]
print(f"{sys.argv[2]:16s} value {l['value']:10.1f}  avg_launch_ms {r['avg_launch_ms']:.5f}  ps/unit {1e9*r['avg_launch_ms']/r['units_per_launch']:.3f}")
PY
}
A=$(pwd)/build_ab/libA.so; B=$(pwd)/build_ab/libB.so
for rep in 1 2 3; do
  run p2p_A$rep $A; run p2p_B$rep $B
done
for rep in 1 2; do
  run gicp_A$rep $A --method 1; run gicp_B$rep $B --method 1
  run hard_A$rep $A --guess hard; run hard_B$rep $B --guess hard
done
cat $O/ab.txt
