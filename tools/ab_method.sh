#!/bin/bash
# A/B of prebuilt library variants on one method: METHOD=1 tools/ab_method.sh build_ab/lib_a.so build_ab/lib_b.so ...  (each twice)
M=${METHOD:-1}
for rep in 1 2; do
for L in "$@"; do
    ELM_LIB=$PWD/$L timeout 600 python bench.py --no-cpu --no-extras --method $M --batch ${BATCH:-2048} --steps 6 > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err
    python - "$L" <<'PY'
import json, sys
r = json.load(open("/tmp/ab.json")); f = r["roofline"]
print("%-28s %8.0f reg/s  iters %.3f  launch %.4f ms" % (sys.argv[1], r["value"], r["config"]["iterations_mean"], f["avg_launch_ms"]), flush=True)
PY
done
done
