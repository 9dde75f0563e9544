#!/usr/bin/env python3
"""Per-kernel mean of one rocprofv3 PMC counter from a counter_collection CSV.  usage: pmc_summary.py csv [csv...]"""
import collections, csv, sys
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            k = (row["Kernel_Name"][:70], row["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(row["Counter_Value"])
    print("#", path)
    for (kn, cn), (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:400]:
        print(f"{cn:12s} n={n:5d} mean={tot/n:14.1f} total={tot:16.1f}  {kn}")
