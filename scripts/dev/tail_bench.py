#!/usr/bin/env python3
"""The regularisation tail alone: ops.reg_tail (fused conv11 + skip + prob) vs the two separate kernels, on the six
stage-pass shapes of a config (one branch), HIP-event timed.   python scripts/dev/tail_bench.py [c2]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dmvsnet_amd import ops, synth  # noqa: E402

cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
w11 = torch.randn(16, 8, 3, 3, 3, generator=g) * 0.1
wp = torch.randn(2, 8, 3, 3, 3, generator=g) * 0.1
conv11 = ops.ConvLayer("c11", ops.DECONV_S2, 3, 16, 8, ops.pack_direct(w11, True).to(dev), ops.pack_mfma(w11, 16, 8, ops.DECONV_S2, 3).to(dev),
                       torch.ones(8, device=dev), torch.zeros(8, device=dev), True)
prob = ops.ConvLayer("p", ops.CONV_S1, 3, 8, 2, ops.pack_direct(wp, False).to(dev), None, None, None, False)


def timed(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


tot = [0.0, 0.0]
for s in range(3):
    sc = 2 ** (2 - s)
    H, W = cfg["H"] // sc, cfg["W"] // sc
    for D in (cfg["ndepths"][s], 4):
        x = torch.randn(16, D // 2, H // 2, W // 2, device=dev)
        skip = torch.randn(8, D, H, W, device=dev)
        out = torch.empty(2, D, H, W, device=dev)
        tf = timed(lambda: ops.reg_tail(x, skip, conv11, prob, out=out))
        t2 = timed(lambda: ops.conv3d(ops.conv3d(x, conv11, skip=skip, backend="mfma"), prob, out=out, backend="direct"))
        vox = D * H * W
        print(f"s{s+1} D={D:3d} {H}x{W}: fused {tf:.4f} ms ({48.0 * vox / tf / 1e6:7.1f} GB/s alg)   conv11 + prob {t2:.4f} ms", flush=True)
        tot[0] += tf; tot[1] += t2
print(f"one branch, six passes: fused {tot[0]:.3f} ms   separate {tot[1]:.3f} ms")
