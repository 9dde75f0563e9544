#!/usr/bin/env python3
"""Per-kernel totals of all counters in rocprofv3 counter_collection CSVs, for this repo's kernels only,
printed as ratios to SQ_WAVE_CYCLES / SQ_BUSY_CYCLES where present.  usage: pmc_ours.py csv..."""
import collections, csv, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            if not any(t in k for t in ("mfma_kernel", "wino_kernel", "warp_corr", "cout2", "depth_regress", "depth_select", "conv2d_c8", "conv0_fused", "coarse_kernel", "zmarch_kernel")):
                continue
            k = k.replace("(anonymous namespace)::", "").replace("void ", "")[:64]
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print(k)
    print("   " + "  ".join(f"{n.replace('SQ_','')}={v/wc:.3f}" if n.startswith("SQ_") and n != "SQ_WAVE_CYCLES" and "INSTS" not in n and n != "SQ_WAVES" else f"{n.replace('SQ_','')}={v:.3g}" for n, v in sorted(c.items())))
