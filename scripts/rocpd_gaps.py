#!/usr/bin/env python3
"""GPU idle time inside a rocprofv3 kernel trace (rocpd sqlite): union of all kernel intervals vs the span of the
last N seconds of the trace, plus the largest gaps and the kernels on either side.
usage: rocpd_gaps.py results.db [tail_ms]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tail_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rows = db.execute("select start, end, name from kernels order by start").fetchall()
anchor = [r[1] for r in rows if "depth_regress" in r[2]]  # the bench's last kernel of a step
t_end = max(anchor) if anchor else max(r[1] for r in rows)
rows = [r for r in rows if t_end - tail_ms * 1e6 <= r[0] <= t_end]
span = rows[-1][1] - rows[0][0]
busy, cur0, cur1 = 0, rows[0][0], rows[0][1]
gaps = []
prev_name = rows[0][2]
for s, e, n in rows[1:]:
    if s > cur1:
        gaps.append((s - cur1, prev_name, n))
        busy += cur1 - cur0
        cur0, cur1 = s, e
    else:
        cur1 = max(cur1, e)
    prev_name = n
busy += cur1 - cur0
print(f"kernels {len(rows)}  span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms  idle {(span-busy)/1e6:.3f} ms ({100*(span-busy)/span:.1f} %)")
print(f"gaps: {len(gaps)}  mean {sum(g[0] for g in gaps)/max(len(gaps),1)/1e3:.2f} us")
import collections
hist = collections.Counter(min(int(g[0] / 1e3) // 2 * 2, 40) for g in gaps)
print("gap histogram (us bucket: count):", sorted(hist.items()))
for g in sorted(gaps, reverse=True)[:12]:
    print(f"  {g[0]/1e3:8.1f} us  after {g[1][:60]}  before {g[2][:60]}")
