#!/usr/bin/env python3
"""Does an intermediate that stays inside the 256 MB Infinity Cache make the full-resolution tail faster?
conv11 (16->8 transposed conv + skip) followed by prob (8->2) at the stage-2 main-pass shape, two ways:
  big     one launch pair over the whole volume (intermediate 485 MB: written to and re-read from HBM)
  slabs   the same voxels as K y-slabs with DIFFERENT inputs / skips / outputs per slab (HBM-cold) but ONE reused
          intermediate buffer per slab size (<= 61 MB: cache resident)
Halo rows are ignored (timing probe, not a correct decomposition)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dmvsnet_amd import ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device="cpu").manual_seed(0)


def layer(cin, cout, mode, w_shape, bn):
    w = torch.randn(*w_shape, generator=g) * 0.1
    tr = mode == ops.DECONV_S2
    wm = ops.pack_mfma(w, cin, cout, mode, 3)
    sc = torch.rand(cout, generator=g).add(0.5).to(dev) if bn else None
    sh = torch.randn(cout, generator=g).mul(0.1).to(dev) if bn else None
    return ops.ConvLayer("t", mode, 3, cin, cout, ops.pack_direct(w, tr).to(dev), None if wm is None else wm.to(dev), sc, sh, bn)


conv11 = layer(16, 8, ops.DECONV_S2, (16, 8, 3, 3, 3), True)
prob = layer(8, 2, ops.CONV_S1, (2, 8, 3, 3, 3), False)
D, H, W = 16, 296, 400        # conv11 input grid of stage 2 main (output 32 x 592 x 800)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


x = torch.randn(16, D, H, W, device=dev)
skip = torch.randn(8, 2 * D, 2 * H, 2 * W, device=dev)
y = torch.empty(8, 2 * D, 2 * H, 2 * W, device=dev)
out = torch.empty(2, 2 * D, 2 * H, 2 * W, device=dev)


def big():
    ops.conv3d(x, conv11, skip=skip, out=y)
    ops.conv3d(y, prob, out=out)


t_big = timeit(big)
t_c11 = timeit(lambda: ops.conv3d(x, conv11, skip=skip, out=y))
t_prob = timeit(lambda: ops.conv3d(y, prob, out=out))
print(f"big: conv11+prob {t_big:.3f} ms (conv11 {t_c11:.3f}, prob {t_prob:.3f})")
for K in (2, 4, 8, 16):
    hs = H // K // 4 * 4
    xs = [torch.randn(16, D, hs, W, device=dev) for _ in range(K)]
    sk = [torch.randn(8, 2 * D, 2 * hs, 2 * W, device=dev) for _ in range(K)]
    os_ = [torch.empty(2, 2 * D, 2 * hs, 2 * W, device=dev) for _ in range(K)]
    for nbuf, label in ((1, "one reused intermediate"), (K, "one intermediate per slab (HBM)")):
        ys = [torch.empty(8, 2 * D, 2 * hs, 2 * W, device=dev) for _ in range(nbuf)]

        def slabs():
            for k in range(K):
                ops.conv3d(xs[k], conv11, skip=sk[k], out=ys[k % nbuf])
                ops.conv3d(ys[k % nbuf], prob, out=os_[k])
        t = timeit(slabs)
        vox = K * hs / H
        print(f"K={K:2d} slabs of {hs} rows ({ys[0].numel() * 4 / 1e6:.0f} MB intermediate), {label}: {t:.3f} ms "
              f"= {t / vox:.3f} ms per full volume")
