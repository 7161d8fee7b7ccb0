#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of the product library's device code (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/resource_table.py [filter-regex] [EXTRA compiler flags]"""
import os, re, subprocess, sys
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "elimaloc_amd", "csrc")
extra = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["make", "-s", "-C", csrc, "resource", f"EXTRA={extra}"], capture_output=True, text=True)
rows, cur = [], None
for line in (out.stdout + out.stderr).splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
flt = re.compile(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else None
print("%-72s %5s %5s %5s %6s %6s %7s %6s %10s" % ("kernel", "VGPR", "AGPR", "SGPR", "vspill", "sspill", "scratch", "LDS", "waves/SIMD"))
for r, n in zip(rows, names):
    n = re.sub(r"\(elm::DevMap.*", "", n).replace("void elm::", "")
    if flt and not flt.search(n):
        continue
    print("%-72s %5d %5d %5d %6d %6d %7d %6d %10d" % (n[:72], r.get("VGPRs", 0), r.get("AGPRs", 0), r.get("TotalSGPRs", 0), r.get("VGPRs Spill", 0),
                                                        r.get("SGPRs Spill", 0), r.get("ScratchSize", 0), r.get("LDS Size", 0), r.get("Occupancy", 0)))
