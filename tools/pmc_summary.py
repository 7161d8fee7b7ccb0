#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in one or more rocprofv3 rocpd databases."""
import sqlite3, sys
for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), avg(end-start) from counters_collection "
                     "group by kernel_name, counter_name order by kernel_name").fetchall()
    print("#", db)
    for r in rows:
        print(f"{r[0][:60]:60s} {r[1]:24s} n={r[2]:4d} avg={r[3]:.6g} avg_dur_us={r[4]/1e3:.1f}")
