import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, "tests")
import test_gpu_parity as T
from dmvsnet_amd import ops
cin, cout, mode, kd, D, H, W = 16, 8, ops.DECONV_S2, 3, 2, 3, 5
w = T.rnd(cin, cout, 3, 3, 3, seed=1, scale=0.1)
layer, scale, shift = T._layer(w, mode, kd, bn=True, seed=3)
x = T.rnd(cin, D, H, W, seed=1)
for use_skip in (False, True):
    skip = T.rnd(cout, 2*D, 2*H, 2*W, seed=2) if use_skip else None
    want = T._conv_ref(x, w, mode, kd, scale, shift, skip)
    got = ops.conv3d(T.cu(x), layer, skip=None if skip is None else T.cu(skip), backend="mfma").cpu()
    d = (got - want).abs()
    bad = (d > 1e-4)
    print("skip", use_skip, "bad frac", bad.float().mean().item(), "max", d.max().item())
    idx = bad.nonzero()
    print(" bad by x parity:", [(bad[..., p::2].float().mean().item()) for p in (0, 1)], "by y parity", [(bad[:, :, p::2].float().mean().item()) for p in (0, 1)], "by z", [(bad[:, p::2].float().mean().item()) for p in (0, 1)], "by co", [bad[c].float().mean().item() for c in range(cout)])
    print(" sample got/want", got[0,0,0,:6].tolist(), want[0,0,0,:6].tolist())
