#!/usr/bin/env python3
"""Per-depth-map instruction totals of the K1 (q4) kernel from a scripts/dev/k1_q4_pmc.sh counter file, and the two
issue-side fractions bench.py reports beside the HBM fraction of `warp_corr`:
  valu_useful_frac  = channel + weight FMAs the algorithm needs (samples x (4 C + 8) lane-FMAs) / SQ_INSTS_VALU
  valu_issue_floor_ms / lds_floor_ms = SQ_INSTS_VALU x 2 clk on 1024 SIMDs, SQ_INSTS_LDS x 4 clk (x the measured
                      bank-conflict factor) on 256 CUs, at 2.1 GHz -- the time the kernel's own instruction streams
                      need with perfect overlap
usage: k1_sq_summary.py profiles/rNN_x_k1_sq.txt LAUNCHES_PER_PASS > profiles/rNN_x_k1_sq_summary.json
(k1_q4_pmc.sh TAG 0 both --reps 2: every stage-pass is launched 8 times: (warm + 2 reps + 1 compare) x (smooth, real))"""
import json
import re
import sys

path, mult = sys.argv[1], float(sys.argv[2])
tot = {}
name = None
for line in open(path):
    if line.startswith("warp_corr_q4_kernel"):
        name = line.strip()
        continue
    if name is None:
        continue
    for k, v in re.findall(r"(\w+)=([0-9.e+-]+)", line):
        if k in ("INSTS_VALU", "INSTS_LDS", "WAVE_CYCLES", "FETCH_SIZE", "WRITE_SIZE"):
            tot.setdefault(k, {})[name] = float(v)
    for k, v in re.findall(r"(LDS_BANK_CONFLICT|LDS_IDX_ACTIVE)=([0-9.e+-]+)", line):
        tot.setdefault(k, {})[name] = float(v)   # fractions of that kernel's wave cycles
    name = None
valu = sum(tot["INSTS_VALU"].values()) / mult
lds = sum(tot["INSTS_LDS"].values()) / mult
# conflict factor: bank-conflict cycles over conflict-free LDS cycles, weighted by wave cycles
wc = tot["WAVE_CYCLES"]
conf = sum(tot["LDS_BANK_CONFLICT"][k] * wc.get(k, 0) for k in tot["LDS_BANK_CONFLICT"])
act = sum(tot["LDS_IDX_ACTIVE"][k] * wc.get(k, 0) for k in tot["LDS_IDX_ACTIVE"])
cf = act / max(act - conf, 1.0)
# config 2: samples per depth map and channel count
useful = (30.3e6 + 1.9e6) * (4 * 32 + 8) + (60.6e6 + 7.6e6) * (4 * 16 + 8) + (60.6e6 + 30.3e6) * (4 * 8 + 8)
out = {"source": path, "insts_valu_per_map": valu, "insts_lds_per_map": lds, "lds_conflict_factor": cf,
       "valu_useful_frac": useful / 64.0 / valu,
       "valu_issue_floor_ms": valu * 2.0 / 1024 / 2.1e9 * 1e3,
       "lds_floor_ms": lds * 4.0 * cf / 256 / 2.1e9 * 1e3,
       "fetch_write_bytes_per_map": (2.0 * sum(tot.get("FETCH_SIZE", {}).values()) + sum(tot.get("WRITE_SIZE", {}).values())) * 1024 / mult}
print(json.dumps(out, indent=1))
