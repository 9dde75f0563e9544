#!/bin/bash
# the kernel sequence of ONE depth map in time order (rocprofv3 --kernel-trace, csv): which launches are not ours
export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/seq
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/seq -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aten-gpu-baseline --no-kernel-timing --no-live-traffic > /tmp/seq.log 2>&1)
csv=$(find /tmp/seq -name '*kernel_trace.csv' | head -1)
python - "$csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last depth map: from the last FeatureNet conv0.0 launch (first conv_mfma_kernel<16, 1, 1, 1, 3, 2 ...) backwards
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
idx = [i for i, n in enumerate(names) if "hyp_first" in n or "hypothesis" in n.lower()]
start = idx[-1] if idx else 0
# go back to the preceding elementwise / copy kernels that belong to this map's FeatureNet
k = start
while k > 0 and "conv_mfma_kernel<16, 1, 1, 1, 3, 2" not in names[k]:
    k -= 1
k = max(0, k - 4)
seq = rows[k:]
t0 = int(seq[0]["Start_Timestamp"])
prev_end = None
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else f"gap {1e-3 * (s - prev_end):7.1f} us"
    print(f"{1e-3 * (s - t0):9.1f} us  dur {1e-3 * (e - s):7.1f} us  {gap:16s} q{r.get('Queue_Id', '?'):>3s}  {short(r['Kernel_Name'])}")
    prev_end = max(prev_end or e, e)
PY
