#!/bin/bash
# Run ON THE MI355X BOX: FETCH_SIZE calibration for the grid kernel's gather pattern -> gpurun_out/gather_probe.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/gather_probe
mkdir -p $OUT
[ -x $R/tools/probes/gather_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/probes/gather_probe.hip -o $R/tools/probes/gather_probe
cd /tmp && export TMPDIR=/tmp
$R/tools/probes/gather_probe > $OUT/expected.txt
GP_SKIP_HOST=1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/p1 -o g -- $R/tools/probes/gather_probe > /dev/null 2> $OUT/p1.err
GP_SKIP_HOST=1 timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum -d $OUT/p2 -o g -- $R/tools/probes/gather_probe > /dev/null 2> $OUT/p2.err
python3 - $OUT <<'PY'
import glob, os, sqlite3, sys
out = sys.argv[1]
exp = {}
for line in open(os.path.join(out, "expected.txt")):
    w = line.split()
    exp[w[0]] = {w[i]: float(w[i + 1]) for i in range(1, len(w), 2)}
cnt = {}
for db in sorted(glob.glob(os.path.join(out, "p*", "*.db"))):
    c = sqlite3.connect(db)
    for k, cn, v in c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
        cnt.setdefault(k.split("(")[0], {})[cn] = v
lines = ["# tools/probes/gather_probe on the MI355X box: rocprofv3 FETCH_SIZE (KB) against byte counts known from the address stream"]
for k, e in exp.items():
    c = next((v for kk, v in cnt.items() if k.split("<")[0] in kk and (("<" not in k) or k.split("<")[1].rstrip(">") in kk)), {})
    fs = c.get("FETCH_SIZE", 0.0) * 1024.0
    lines.append(f"{k}: FETCH_SIZE {fs:.4g} B  " + "  ".join(f"{n} {v:.4g} (FETCH/{n} = {fs / v:.3f})" for n, v in e.items()) +
                 "  " + "  ".join(f"{n} {v:.4g}" for n, v in c.items() if n != "FETCH_SIZE"))
open(os.path.join(out, "..", "gather_probe.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
