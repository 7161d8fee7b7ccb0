#!/usr/bin/env python3
"""Turn the rocprofv3 databases written by tools/collect_profiles.sh into small text/JSON summaries
(kernel_stats.csv, pmc.json) that are committed under profiles/."""
import json, os, sqlite3, sys

out = sys.argv[1]


def db(name):
    p = os.path.join(out, name, "b_results.db")
    return sqlite3.connect(p) if os.path.exists(p) else None


c = db("trace")
if c:
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                     "group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(os.path.join(out, "kernel_stats.csv"), "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage\n")
        for r in rows:
            f.write(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.1f},{r[4]},{r[5]},{100.0 * r[2] / tot:.2f}\n")
pmc = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_ta"):
    c = db(name)
    if not c:
        continue
    for k, cn, n, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                   "group by kernel_name, counter_name"):
        pmc.setdefault(k, {})[cn] = {"launches": n, "avg": avg}
summary = {"counters": pmc, "source": os.path.basename(out.rstrip("/"))}
try:
    b = json.load(open(os.path.join(out, "bench_trace.json")))
    kname = b["roofline"]["kernel"].split("<")[0]
    kidx = {"P2P": "0", "GICP": "1", "VGICP": "2", "AVGICP": "3"}[b["roofline"]["kernel"].split("<")[1].rstrip(">")]
    for k, v in pmc.items():
        kk = k.replace("(elm::IcpMethod)", "")
        if (kname + "<" + kidx + ">" in kk or kname + "<" + kidx + "," in kk) and "FETCH_SIZE" in v:
            fetch_kb = v["FETCH_SIZE"]["avg"]
            write_kb = v.get("WRITE_SIZE", {}).get("avg", 0.0)
            units = max(b["roofline"]["units_per_launch"], 1.0)
            hbm_view = b["roofline"].get("hbm", b["roofline"])
            # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports exactly half of the bytes
            # of a wide coalesced read stream -- but, calibrated with tools/probes/gather_probe (profiles/r03_gather_probe.txt), the
            # FULL bytes of this kernel's gathers (64-byte sectors of random 48-byte blocks / payload records: FETCH_SIZE = 1.008 x
            # the unique sectors touched).  So: x1 for the gathered bytes + the uncounted half of the coalesced scan-point stream
            # (12 B per unit, half of it missing = 6 B); the x2 figure of round 2 is kept as an upper bound.  WRITE_SIZE is
            # uncalibrated (tiny here).  FETCH_SIZE counts Infinity-Cache hits: fabric-side traffic, an upper bound on DRAM bytes.
            summary.update({"kernel": b["roofline"]["kernel"], "method": {"P2P": 0, "GICP": 1, "VGICP": 2, "AVGICP": 3}[b["roofline"]["kernel"].split("<")[1].rstrip(">")],
                            "batch": b["config"]["batch_per_gpu"], "scan_points": 131072,
                            "fetch_size_kb_avg_raw": fetch_kb, "write_size_kb_avg": write_kb,
                            "hbm_bytes_per_launch": (fetch_kb + write_kb) * 1024.0 + 6.0 * units,
                            "hbm_bytes_per_launch_x2_upper": (2.0 * fetch_kb + write_kb) * 1024.0,
                            "launches": v["FETCH_SIZE"]["launches"],
                            # traffic per unit of work (scan point x iteration): launch mixes differ between runs (early-exit
                            # launches move no data), so bench.py scales this by ITS units per launch
                            "units_per_launch_profiled": b["roofline"]["units_per_launch"],
                            "hbm_bytes_per_unit": (fetch_kb + write_kb) * 1024.0 / units + 6.0,
                            "hbm_bytes_per_unit_x2_upper": (2.0 * fetch_kb + write_kb) * 1024.0 / units,
                            "calibration": "profiles/r03_gather_probe.txt: FETCH_SIZE x1 for 64-byte gather sectors, x2 only for the coalesced scan stream",
                            "requested_bytes_per_unit": hbm_view.get("requested_bytes_per_unit")})
            if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v:
                cyc = v["GRBM_GUI_ACTIVE"]["avg"] / 8.0  # summed over the 8 XCDs
                summary["valu_busy"] = 4.0 * v["SQ_ACTIVE_INST_VALU"]["avg"] / (1024.0 * cyc)  # 4 cycles per wave64 instruction, 1024 SIMDs
                summary["valu_insts_per_wave"] = v["SQ_INSTS_VALU"]["avg"] / v["SQ_WAVES"]["avg"]
                summary["waves_per_launch"] = v["SQ_WAVES"]["avg"]
            if "TA_TA_BUSY_sum" in v and "GRBM_GUI_ACTIVE" in v:
                cyc = v["GRBM_GUI_ACTIVE"]["avg"] / 8.0
                summary["ta_busy"] = v["TA_TA_BUSY_sum"]["avg"] / 256.0 / cyc  # 256 CUs
                summary["l1_line_accesses_per_cu_cycle"] = v["TCP_TOTAL_CACHE_ACCESSES_sum"]["avg"] / 256.0 / cyc
                summary["l1_hit"] = 1.0 - v["TCP_TCC_READ_REQ_sum"]["avg"] / max(v["TCP_TOTAL_CACHE_ACCESSES_sum"]["avg"], 1.0)
            if "TCC_HIT_sum" in v:
                summary["l2_hit"] = v["TCC_HIT_sum"]["avg"] / (v["TCC_HIT_sum"]["avg"] + v["TCC_MISS_sum"]["avg"])
except Exception as e:  # noqa: BLE001
    summary["error"] = repr(e)
json.dump(summary, open(os.path.join(out, "pmc.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "counters"}))
