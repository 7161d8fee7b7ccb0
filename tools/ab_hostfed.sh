#!/bin/bash
for L in "$@"; do
    ELM_LIB=$PWD/$L timeout 600 python bench.py --no-cpu --batch 1024 --steps 5 --hostfed-batch 1024 --hostfed-steps 8 > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err
    python - "$L" <<'PY'
import json, sys
r = json.load(open("/tmp/ab.json")); h = r["host_fed"]
print("%-28s resident %8.0f  host_fed %8.0f reg/s  pcie %.1f of %.1f GB/s  identical %s" % (sys.argv[1], r["value"], h["value"], h["pcie_achieved_gbs"], h["pcie_h2d_probe_gbs"], h["bit_identical_to_resident"]), flush=True)
PY
done
