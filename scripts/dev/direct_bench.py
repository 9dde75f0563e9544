#!/usr/bin/env python3
"""Times direct-form K3 layers of config 2 alone under dmvs_tune settings (development).
    python scripts/dev/direct_bench.py [--tune k3_single_buf_min_blocks=1073741824]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dmvsnet_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tune", default="")
ap.add_argument("--reps", type=int, default=9)
args = ap.parse_args()
lib = _lib.load()
for kv in filter(None, args.tune.split(",")):
    n, v = kv.split("=")
    assert lib.dmvs_tune(n.encode(), int(v)) == 0, kv
dev = "cuda:0"
SHAPES = [  # (name, cin, cout, mode, kd, D, H, W, skip)
    ("s2.conv1", 8, 16, ops.CONV_S2, 3, 32, 592, 800, False), ("s3.conv1", 8, 16, ops.CONV_S2, 3, 8, 1184, 1600, False),
    ("s2.conv3", 16, 32, ops.CONV_S2, 3, 16, 296, 400, False), ("s2.conv5", 32, 64, ops.CONV_S2, 3, 8, 148, 200, False),
    ("s2.conv7", 64, 32, ops.DECONV_S2, 3, 4, 74, 100, True), ("s2.conv9", 32, 16, ops.DECONV_S2, 3, 8, 148, 200, True),
    ("s2.conv11", 16, 8, ops.DECONV_S2, 3, 16, 296, 400, True), ("s3.conv11", 16, 8, ops.DECONV_S2, 3, 4, 592, 800, True),
]


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


for name, cin, cout, mode, kd, D, H, W, skip in SHAPES:
    tr = mode == ops.DECONV_S2
    w = torch.randn(((cin, cout) if tr else (cout, cin)) + (3, 3, 3)) * 0.05
    layer = ops.ConvLayer(name, mode, kd, cin, cout, None, ops.pack_mfma(w, cin, cout, mode, kd).to(dev), torch.ones(cout, device=dev),
                          torch.zeros(cout, device=dev), True)
    x = torch.randn(cin, D, H, W, device=dev)
    Do, Ho, Wo = layer.out_shape(D, H, W)
    out = torch.empty(cout, Do, Ho, Wo, device=dev)
    sk = torch.randn(cout, Do, Ho, Wo, device=dev) if skip else None
    t = timed(lambda: ops.conv3d(x, layer, skip=sk, out=out, backend="mfma"))
    vox = D * H * W if tr else Do * Ho * Wo
    gf = 2.0 * 27 * cin * cout * vox / 1e9
    gb = 4.0 * (x.numel() + out.numel() * (2 if skip else 1)) / 1e9
    print(f"{name:10s} {t:.3f} ms  {gf / t:6.1f} TF/s  {gb / t * 1e3:6.0f} GB/s", flush=True)
