#!/usr/bin/env python3
"""A/B of the K1 variants on the REAL inputs of every stage-pass (bench config, synthetic weights): hooks
ops.warp_corr during one forward, then replays each call with dmvs_tune("k1_variant", 1 | 2)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dmvsnet_amd import CostAgg, MVSNet, _lib, ops, synth  # noqa: E402

cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
net = MVSNet(cfg["ndepths"], cfg["ratios"], verbose=False)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
net = net.cuda()
net.return_prob_volume = False
CostAgg.autotune = False   # capture one call per pass, in the library default
imgs, proj, dv = synth.synth_inputs(cfg["H"], cfg["W"], cfg["V"], 0)
calls = []
orig = ops.warp_corr


def hook(ref, src, p12, depth, *a, **k):
    calls.append((ref, list(src), p12.clone(), depth))   # planes: a fresh tensor / AffinePlanes per pass
    return orig(ref, src, p12, depth, *a, **k)


ops.warp_corr = hook
net(imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, dv.cuda())
torch.cuda.synchronize()
ops.warp_corr = orig
lib = _lib.load()
tot = {1: 0.0, 2: 0.0, 3: 0.0, 4: 0.0}
for i, (ref, src, p12, depth) in enumerate(calls):
    row = []
    outs = {}
    for var in (1, 2, 3, 4):
        lib.dmvs_tune(b"k1_variant", var)
        outs[var] = ops.warp_corr(ref, src, p12, depth)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); ops.warp_corr(ref, src, p12, depth); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        row.append(ts[3])
        tot[var] += ts[3]
    diff = max((outs[1] - outs[v]).abs().max().item() for v in (2, 3, 4))
    print(f"pass {i} C={ref.shape[-1]} D={depth.shape[0]} {depth.shape[1]}x{depth.shape[2]}: lds {row[0]:.4f} ms  px {row[1]:.4f} ms  "
          f"px_big {row[2]:.4f} ms  lds_bc {row[3]:.4f} ms  max|diff| {diff:.2e}")
print(f"total: lds {tot[1]:.3f} ms  px {tot[2]:.3f} ms  px_big {tot[3]:.3f} ms  lds_bc {tot[4]:.3f} ms")
lib.dmvs_tune(b"k1_variant", 0)
