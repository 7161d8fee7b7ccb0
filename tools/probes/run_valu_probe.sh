#!/bin/bash
# Run ON THE MI355X BOX: VALU issue cost per instruction class + what the SQ counters read for it -> gpurun_out/valu_probe.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/valu_probe
mkdir -p $OUT
[ -x $R/tools/probes/valu_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/tools/probes/valu_probe.hip -o $R/tools/probes/valu_probe
cd /tmp && export TMPDIR=/tmp
$R/tools/probes/valu_probe > $OUT/timing.txt
for W in 1 2 8; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
      -d $OUT/w$W -o v -- $R/tools/probes/valu_probe $W > $OUT/w$W.txt 2> $OUT/w$W.err
done
# does a cycle-weighted VALU counter exist on this chip?  (a failed pass only loses its own columns)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/t8 -o v -- $R/tools/probes/valu_probe 8 > $OUT/t8.txt 2> $OUT/t8.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/i8 -o v -- $R/tools/probes/valu_probe 8 > $OUT/i8.txt 2> $OUT/i8.err
python3 - $OUT <<'PY'
import glob, os, sqlite3, sys
out = sys.argv[1]
lines = ["# tools/probes/valu_probe on the MI355X box (tools/probes/run_valu_probe.sh)", ""] + open(os.path.join(out, "timing.txt")).read().splitlines()
lines += ["", "# the same launches under rocprofv3 --pmc (the iters = 2000 launch of every kernel; counters summed over the chip):",
          "#   inst/wave = SQ_INSTS_VALU / SQ_WAVES;  active/inst = SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU (counter units per wave64 instruction);",
          "#   kernel_cyc = GRBM_GUI_ACTIVE / 8 XCDs;  x4 busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel_cyc) -- the figure bench.py quoted in round 3",
          "#   cyc/inst/SIMD = (kernel_cyc - kernel_cyc of the empty fill launch) / (W x inst/wave): shader cycles one wave64 instruction occupies its SIMD",
          f"{'kernel':14s} {'W':>2s} {'inst/wave':>10s} {'active/inst':>12s} {'kernel_cyc':>11s} {'x4 busy':>8s} {'wave_cyc/inst/wave':>19s} {'cyc/inst/SIMD':>14s}"]
for W in (1, 2, 8):
    for db in sorted(glob.glob(os.path.join(out, f"w{W}", "**", "*.db"), recursive=True)):
        c = sqlite3.connect(db)
        rows = {}
        for k, did, cn, v in c.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection"):
            rows.setdefault((k, did), {})[cn] = rows.setdefault((k, did), {}).get(cn, 0.0) + v
        best = {}
        for (k, did), v in rows.items():  # the long launch of every kernel = the one with the most instructions
            if k not in best or v.get("SQ_INSTS_VALU", 0) > best[k].get("SQ_INSTS_VALU", 0):
                best[k] = v
        base = min((v["GRBM_GUI_ACTIVE"] / 8.0 for k, v in best.items() if "k_valu" not in k and v.get("GRBM_GUI_ACTIVE")), default=0.0)
        for k in sorted(best, key=lambda s: int(s.split("<")[1].split(">")[0]) if "<" in s else 0):
            v = best[k]
            if not v.get("SQ_INSTS_VALU") or "k_valu" not in k:
                continue
            cyc = v["GRBM_GUI_ACTIVE"] / 8.0
            ipw = v["SQ_INSTS_VALU"] / v["SQ_WAVES"]
            lines.append(f"{k.split('(')[0][-14:]:14s} {W:2d} {ipw:10.0f} {v['SQ_ACTIVE_INST_VALU'] / v['SQ_INSTS_VALU']:12.3f} "
                         f"{cyc:11.0f} {4.0 * v['SQ_ACTIVE_INST_VALU'] / (1024.0 * cyc):8.3f} {v['SQ_WAVE_CYCLES'] / v['SQ_INSTS_VALU']:19.3f} {(cyc - base) / (W * ipw):14.3f}")
for tag, cn in (("t8", "SQ_THREAD_CYCLES_VALU"), ("i8", "SQ_INST_CYCLES_VALU")):
    for db in sorted(glob.glob(os.path.join(out, tag, "**", "*.db"), recursive=True)):
        c = sqlite3.connect(db)
        rows = {}
        for k, did, n, v in c.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection"):
            rows.setdefault((k, did), {})[n] = rows.setdefault((k, did), {}).get(n, 0.0) + v
        best = {}
        for (k, did), v in rows.items():
            if k not in best or v.get("SQ_INSTS_VALU", 0) > best[k].get("SQ_INSTS_VALU", 0):
                best[k] = v
        lines += ["", f"# {cn} at W = 8: counter / SQ_INSTS_VALU (per wave64 instruction) and counter / (1024 SIMDs x kernel cycles)"]
        for k in sorted(best, key=lambda s: int(s.split("<")[1].split(">")[0]) if "<" in s else 0):
            v = best[k]
            if v.get("SQ_INSTS_VALU") and cn in v:
                cyc = v["GRBM_GUI_ACTIVE"] / 8.0
                lines.append(f"{k.split('(')[0][-14:]:14s} {v[cn] / v['SQ_INSTS_VALU']:12.3f} {v[cn] / (1024.0 * cyc):12.3f}")
open(os.path.join(out, "..", "valu_probe.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
