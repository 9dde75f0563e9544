import sqlite3, sys, glob
db = sqlite3.connect(glob.glob('/tmp/prof_seq/**/*.db', recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# take the last ~200 kernels (one forward is ~180)
names=[r[0][:60] for r in rows]
import collections
idx=[i for i,n in enumerate(names) if 'copyBuffer' in n or 'elementwise' in n or 'fill' in n.lower()]
print(len(rows), "kernels;", len(idx), "copies/elementwise")
tail=rows[-420:]
for i,(n,s,e) in enumerate(tail):
    if 'copyBuffer' in n or 'elementwise' in n or 'fill' in n.lower() or 'index' in n.lower():
        prev=tail[i-1][0][:50] if i else ''
        nxt=tail[i+1][0][:50] if i+1<len(tail) else ''
        print(f"{(e-s)/1e3:7.1f}us  {n[:70]}   after: {prev}   before: {nxt}")
