#!/bin/bash
# Run ON THE MI355X BOX: L1-hit gather rate by address pattern + what TCP_TOTAL_CACHE_ACCESSES / TA_TA_BUSY read for the same launches
#   -> gpurun_out/l1_probe.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/l1_probe
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/probes/l1_probe.hip -o /tmp/l1_probe || exit 1
cd /tmp && export TMPDIR=/tmp
/tmp/l1_probe > $OUT/timed.txt
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES -d $OUT/p1 -o g -- /tmp/l1_probe > /dev/null 2> $OUT/p1.err
python3 - $OUT <<'PY'
import glob, os, sqlite3, sys
out = sys.argv[1]
lines = ["# tools/probes/l1_probe on the MI355X box (tools/probes/run_l1_probe.sh): timed run, then the counters of the same launches (second launch of each mode)"]
lines += [l.rstrip() for l in open(os.path.join(out, "timed.txt"))]
for db in sorted(glob.glob(os.path.join(out, "p*", "*.db"))):
    c = sqlite3.connect(db)
    per = {}
    for k, did, cn, v in c.execute("select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection group by kernel_name, dispatch_id, counter_name"):
        per.setdefault(k.split("(")[0], {}).setdefault(cn, []).append(v)
    dur = {k.split("(")[0]: a for k, a in c.execute("select name, avg(end - start) from kernels group by name")}
    for k in sorted(per):
        m = {cn: v[-1] for cn, v in per[k].items()}
        cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # summed over the 8 XCDs
        s = f"{k[-12:]}  {dur.get(k, 0.0) / 1e3:9.1f} us  cycles/CU {cyc:.4g}"
        for cn, v in sorted(m.items()):
            if cn == "GRBM_GUI_ACTIVE": continue
            s += f"  {cn} {v:.4g} ({v / 256.0 / cyc if cyc else 0.0:.3f}/CU-cycle)"
        lines.append(s)
open(os.path.join(out, "..", "l1_probe.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/p1
