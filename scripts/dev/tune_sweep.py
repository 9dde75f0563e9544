#!/usr/bin/env python3
"""Sweep one dmvs_tune knob over values on the bench config: depth-maps/s per value (short runs, same process).
    python scripts/dev/tune_sweep.py k3_single_buf_min_blocks 0 512 1024 2048 100000"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dmvsnet_amd import MVSNet, _lib, synth  # noqa: E402

name, values = sys.argv[1], [int(v) for v in sys.argv[2:]]
cfg = synth.CONFIGS["c2"]
net = MVSNet(cfg["ndepths"], cfg["ratios"], verbose=False)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
net = net.cuda()
net.return_prob_volume = False
imgs, proj, dv = synth.synth_inputs(cfg["H"], cfg["W"], cfg["V"], 0)
args = (imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, dv.cuda())
lib = _lib.load()
for _ in range(5):
    net(*args)
torch.cuda.synchronize()
for rnd in range(2):
    for v in values:
        _lib.check(lib.dmvs_tune(name.encode(), v), "tune")
        for _ in range(3):
            net(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for i in range(n):
            net(*args)
            if i % 2 == 1:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name}={v}: {n / dt:.2f} maps/s ({1e3 * dt / n:.3f} ms)", flush=True)
