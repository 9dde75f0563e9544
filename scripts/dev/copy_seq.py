#!/usr/bin/env python3
"""Where do D2D copies / torch elementwise kernels sit inside a forward?  usage: copy_seq.py <dir with rocpd .db>"""
import glob
import sqlite3
import sys

db = sqlite3.connect(glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
names = [r[0] for r in rows]
# one steady-state forward: between the last two FeatureNet input permutes
marks = [i for i, n in enumerate(names) if 'elementwise_kernel_manual_unroll' in n]
a, b = marks[-6], marks[-5]
print(len(rows), "kernels; forward window", a, b, "=", b - a, "kernels")
for i in range(a, b):
    n, s, e = rows[i]
    tag = '   <<<' if ('copyBuffer' in n or 'elementwise' in n or 'fill' in n.lower() or 'index' in n.lower() or 'at::' in n) else ''
    print(f"{(e - s) / 1e3:8.1f}us  {n[:90]}{tag}")
