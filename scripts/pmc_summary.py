#!/usr/bin/env python3
"""Per-kernel (and per-family) totals of one rocprofv3 PMC counter from counter_collection CSVs.

usage: pmc_summary.py [--launch-log log.json] csv [csv...]

Per kernel name: launches, mean, total (the counter's unit; FETCH_SIZE / WRITE_SIZE are KiB).
With --launch-log (written by `bench.py --launch-log`, the family of every K1..K4 launch in host order) the rows of
the dmvs kernels are also attributed to their FAMILY by dispatch order -- kernel names cannot tell a FeatureNet launch
of the MFMA conv kernel from a regularisation launch -- and printed as
    FAMILY <counter> family=<name> n=<launches> total=<value>
which is what bench.py's pmc_traffic() reads."""
import collections
import csv
import json
import sys

# every kernel whose launch ops.py records in the launch log (ops._log): keep in step with the kernels of dmvsnet_amd/csrc --
# a missing name makes the dispatch count differ from the log and drops every FAMILY line.
# tests/test_boundary.py::test_pmc_summary_knows_every_logged_kernel greps the .hip sources against this list.
DMVS_KERNELS = ("mfma_kernel", "wino_kernel", "warp_corr", "conv_cout2", "conv_direct_kernel", "deconv_direct_kernel", "depth_regress",
                "conv2d_c8_kernel", "conv0_fused_kernel", "coarse_kernel", "zmarch_kernel", "conv1_split_kernel", "depth_select_kernel")

args = sys.argv[1:]
log = None
if args and args[0] == "--launch-log":
    log = json.load(open(args[1]))
    args = args[2:]
for path in args:
    acc = collections.defaultdict(lambda: [0, 0.0])
    rows = []
    with open(path) as f:
        for row in csv.DictReader(f):
            k = (row["Kernel_Name"][:70], row["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(row["Counter_Value"])
            if log is not None and any(p in row["Kernel_Name"] for p in DMVS_KERNELS):
                rows.append((int(row["Dispatch_Id"]), row["Counter_Name"], float(row["Counter_Value"])))
    print("#", path)
    for (kn, cn), (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:400]:
        print(f"{cn:12s} n={n:5d} mean={tot/n:14.1f} total={tot:16.1f}  {kn}")
    if log is not None:
        per_counter = collections.defaultdict(list)
        for did, cn, val in sorted(rows):
            per_counter[cn].append(val)
        for cn, vals in per_counter.items():
            if len(vals) != len(log):
                print(f"# launch log has {len(log)} entries but {len(vals)} dmvs dispatches carry {cn}: no per-family lines")
                continue
            fam = collections.defaultdict(lambda: [0, 0.0])
            for name, v in zip(log, vals):
                fam[name][0] += 1
                fam[name][1] += v
            for name, (n, tot) in sorted(fam.items()):
                print(f"FAMILY {cn} family={name} n={n} total={tot:.1f}")
