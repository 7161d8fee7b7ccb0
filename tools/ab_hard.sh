#!/bin/bash
# A/B of prebuilt library variants on the hard initial-guess set: tools/ab_hard.sh build_ab/lib_a.so build_ab/lib_b.so ...
for L in "$@"; do
  for G in hard easy; do
    ELM_LIB=$PWD/$L timeout 600 python bench.py --no-cpu --no-extras --batch 512 --steps 6 --guess $G > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err
    python - "$L" $G <<'PY'
import json, sys
r = json.load(open("/tmp/ab.json")); f = r["roofline"]
print("%-28s %-5s %8.0f reg/s  iters %.3f  launch %.4f ms" % (sys.argv[1], sys.argv[2], r["value"], r["config"]["iterations_mean"], f["avg_launch_ms"]), flush=True)
PY
  done
done
