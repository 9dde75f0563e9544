#!/usr/bin/env python3
"""Times the Winograd conv kernel (K3w) on the layer shapes of config 2, against the direct-form kernel, with the
dmvs_tune("wino_stages") knob; with DMVS_LIB pointing at a knock-out build (scripts/dev/wino_ko.sh) the same shapes
give the phase knock-out table.
    python scripts/dev/wino_bench.py [--stages 0,1,2] [--direct]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dmvsnet_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--stages", default="0")
ap.add_argument("--direct", action="store_true")
ap.add_argument("--reps", type=int, default=9)
ap.add_argument("--tune", default="", help="name=value,... passed to dmvs_tune")
ap.add_argument("--only", default=None, help="comma-separated layer names")
args = ap.parse_args()
lib = _lib.load()
dev = "cuda:0"
for kv in filter(None, args.tune.split(",")):
    n, v = kv.split("=")
    assert lib.dmvs_tune(n.encode(), int(v)) == 0, kv
SHAPES = [  # (name, cin, cout, kd, D, H, W)
    ("s1.conv0", 2, 16, 3, 64, 296, 400), ("s2.conv0", 2, 16, 3, 32, 592, 800), ("s3.conv0", 2, 16, 3, 8, 1184, 1600),
    ("s3r.conv0", 2, 16, 3, 4, 1184, 1600),
    ("s2.conv2", 16, 16, 3, 16, 296, 400), ("s3.conv2", 16, 16, 3, 4, 592, 800), ("s1.conv2", 16, 16, 3, 32, 148, 200),
    ("s2.conv4", 32, 32, 3, 8, 148, 200), ("s3.conv4", 32, 32, 3, 2, 296, 400), ("s2.conv6", 64, 64, 3, 4, 74, 100),
    ("s3.conv6@d1", 64, 64, 1, 1, 148, 200), ("f.conv1.1", 16, 16, 1, 5, 592, 800), ("f.conv2.1", 32, 32, 1, 5, 296, 400),
    ("f.out2", 32, 32, 1, 5, 592, 800), ("f.out3", 32, 16, 1, 5, 1184, 1600),
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


for name, cin, cout, kd, D, H, W in SHAPES:
    if args.only and name not in args.only.split(","):
        continue
    w = torch.randn((cout, cin) + ((3, 3, 3) if kd == 3 else (3, 3))) * 0.05
    wm = ops.pack_mfma(w, cin, cout, ops.CONV_S1, kd)
    layer = ops.ConvLayer(name, ops.CONV_S1, kd, cin, cout, None, None if wm is None else wm.to(dev), torch.ones(cout, device=dev),
                          torch.zeros(cout, device=dev), True, ops.pack_wino(w, cin, cout, kd).to(dev))
    x = torch.randn(cin, D, H, W, device=dev)
    out = torch.empty(cout, D, H, W, device=dev)
    gf = 2.0 * 9 * kd * cin * cout * D * H * W / 1e9
    row = []
    if args.direct and wm is not None:
        t = timed(lambda: ops.conv3d(x, layer, out=out, backend="mfma"))
        row.append(f"direct {t:.3f} ms {gf / t:6.1f} TF/s")
    for st in [int(v) for v in args.stages.split(",")]:
        lib.dmvs_tune(b"wino_stages", st)
        t = timed(lambda: ops.conv3d(x, layer, out=out, backend="wino"))
        row.append(f"wino[st{st}] {t:.3f} ms {gf / t:6.1f} TF/s-eq")
    print(f"{name:12s} {cin}>{cout} kd{kd} {D}x{H}x{W}: " + "   ".join(row), flush=True)
