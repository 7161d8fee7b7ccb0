#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace) as the per-kernel stats table
(name, calls, total/avg/min/max duration in us, % of GPU time)."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for r in rows:
        lines.append(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.1f},{r[4]},{r[5]},{100.0 * r[2] / tot:.2f}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main(*sys.argv[1:])
