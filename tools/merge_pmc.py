#!/usr/bin/env python3
"""Merge the pmc.json summaries of tools/collect_profiles.sh runs into profiles/pmc_latest.json (keyed by kernel name; what
bench.py reads for roofline.traffic) and copy the per-run summaries into profiles/ as <tag>_{kernel_stats.csv,pmc.json,bench*.json}.
    python tools/merge_pmc.py gpurun_out/prof_r02_p2p [gpurun_out/prof_r02_gicp ...]"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles", "pmc_latest.json")
cur = {}
if os.path.exists(dst):
    try:
        cur = json.load(open(dst))
        if "kernel" in cur:  # the round-1 single-kernel layout
            cur = {}
    except Exception:
        cur = {}
for d in sys.argv[1:]:
    tag = os.path.basename(d.rstrip("/")).replace("prof_", "")
    pm = json.load(open(os.path.join(d, "pmc.json")))
    small = {k: v for k, v in pm.items() if k != "counters"}
    if "kernel" in small:  # one entry per kernel, workload size and initial-guess set (bench.py: pmc_key; it checks batch / slots as well)
        key = small["kernel"]
        if int(small.get("scan_points", 131072)) != 131072 or int(small.get("map_points", 10_000_000)) != 10_000_000:
            key += f"@{int(small['scan_points'])}/{int(small['map_points'])}"
        cur[key + ("@hard" if small.get("guess") == "hard" else "") + ("" if small.get("world", "lattice") == "lattice" else "@" + small["world"])] = small
    for src, name in (("kernel_stats.csv", f"{tag}_kernel_stats.csv"), ("pmc.json", f"{tag}_pmc.json"), ("bench.json", f"{tag}_bench.json"),
                      ("bench_trace.json", f"{tag}_bench_under_rocprof.json")):
        if os.path.exists(os.path.join(d, src)):
            shutil.copy(os.path.join(d, src), os.path.join(ROOT, "profiles", name))
json.dump(cur, open(dst, "w"), indent=1)
print("profiles/pmc_latest.json:", list(cur))
