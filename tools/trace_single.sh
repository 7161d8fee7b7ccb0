#!/bin/bash
# Run ON THE MI355X BOX: timeline of resident single registrations (131 072-point scan vs the 10 M-point map, P2P): every launch and copy
# of the last few calls with its duration and the gap before it -> gpurun_out/single_timeline.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/single
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > $OUT/run.py <<PY
import sys, time; sys.path.insert(0, "$R")
import numpy as np
from elimaloc_amd import synth
from elimaloc_amd.registration import Context, VoxelHashMap, Registration, RegistrationConfig, IcpMethod, Scan
c = Context(0)
world = synth.make_world(10_000_000, seed=1001)
vm = VoxelHashMap(1.0, 30, c); vm.AddPoints(world)
reg = Registration(RegistrationConfig(icp_method=IcpMethod.P2P), c)
scans, T0s = [], []
for i in range(4):
    sc, Tt = synth.make_scan(world, 131072, seed=2002 + i)
    scans.append(Scan(c, sc)); T0s.append(synth.perturb(Tt, seed=3003 + i))
for r in range(3):
    for i in range(4):
        t = time.perf_counter(); out = reg.RunRegisterBatch([scans[i]], vm, [T0s[i]]); dt = time.perf_counter() - t
print("last call: %.1f us, iterations %d" % (dt * 1e6, out[0]["iterations"]))
PY
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/trace -o b --output-format csv -- python $OUT/run.py > $OUT/run.log 2> $OUT/err
python - $OUT <<'PY' > $R/gpurun_out/single_timeline.txt
import csv, glob, sys, os
out = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-34:]))
for f in glob.glob(os.path.join(out, "trace", "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
rows.sort()
print(open(os.path.join(out, "run.log")).read().strip())
prev = None
for s, e, n in rows[-40:]:
    print("%-40s dur %8.1f us  gap %7.1f us" % (n, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
PY
cat $R/gpurun_out/single_timeline.txt
