#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace: per-kernel calls / total / avg / % -- the same
table `rocprofv3 --stats` prints.  usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                  "from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for name, n, tot, avg, mn, mx in rows[:40]:
    short = name if len(name) < 110 else name[:107] + "..."
    lines.append(f"| `{short}` | {n} | {tot/1e6:.3f} | {avg/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*tot/total:.1f} |")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
