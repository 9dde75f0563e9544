"""Development: VGPR / AGPR / LDS / spill counts of every kernel in a hipcc -S listing (stdin or file)."""
import re
import subprocess
import sys

text = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
keys = ("name", "vgpr_count", "agpr_count", "group_segment_fixed_size", "vgpr_spill_count")
cur = {}
rows = []
for line in text.splitlines():
    m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)", line)
    if m and m.group(1) in keys:
        cur[m.group(1)] = m.group(2)
        if len(cur) == len(keys):
            rows.append(cur)
            cur = {}
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*\)$", "", n)
    print(f"{n:64s} vgpr {r['vgpr_count']:>4s} agpr {r['agpr_count']:>3s} lds {r['group_segment_fixed_size']:>6s} spill {r['vgpr_spill_count']}")
