#!/usr/bin/env python3
"""Do two DIFFERENT kernels of the path gain from running concurrently on two HIP streams?  (dev; r04)
Times K1 (s2.main shape) and conv0x2 (s2.main) alone, back to back, and concurrently on two streams with independent buffers;
the same for conv11 + prob of different branches.  What would be gained by pipelining K1 and conv0 over depth slabs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dmvsnet_amd import MVSNet, ops, synth  # noqa: E402
dev = torch.device("cuda:0")
cfg = synth.CONFIGS["c2"]
net = MVSNet(cfg["ndepths"], cfg["ratios"], verbose=False)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0)); net = net.to(dev); net.prepare(dev)
H, W, V = cfg["H"], cfg["W"], cfg["V"]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=9):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


def pair(name, fa, fb):
    ta, tb = timed(fa), timed(fb)
    def both():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1): fa()
        with torch.cuda.stream(s2): fb()
        cur.wait_stream(s1); cur.wait_stream(s2)
    tc = timed(both)
    print(f"{name}: A {ta:.3f}  B {tb:.3f}  sum {ta + tb:.3f}  concurrent {tc:.3f}  ({tc / (ta + tb):.2f} of the sum, max {max(ta, tb):.3f})")


for stage in (1, 2, 3):
    sc = 2 ** (3 - stage); h, w, C, D = H // sc, W // sc, (32, 16, 8)[stage - 1], cfg["ndepths"][stage - 1]
    g = torch.Generator().manual_seed(0)
    feats = [ops.hwc_to_q4(torch.randn(h, w, C, generator=g).to(dev)) for _ in range(V)]
    cams = synth.synth_cameras(H, W, V)
    p12 = ops.relative_proj(cams[f"stage{stage}"][0].to(dev).contiguous())
    dv = synth.synth_depth_values().to(dev)
    hyp, _ = ops.hypotheses_first(dv, D, h, w, False, True)
    sim_out = torch.empty((2, D, h, w), device=dev)
    conv0, small, huge = net.cost_regularization[stage - 1]._packed
    x = torch.randn(2, D, h, w, device=dev); c0 = torch.empty((16, D, h, w), device=dev)
    pair(f"s{stage}.main K1 + conv0x2", lambda: ops.warp_corr(feats[0], feats[1:], p12, hyp, out=sim_out), lambda: ops.conv3d(x, conv0, out=c0))
    y = torch.randn(16, D // 2, h // 2, w // 2, device=dev); sk = torch.randn(8, D, h, w, device=dev); t = torch.empty(8, D, h, w, device=dev)
    t2 = torch.randn(8, D, h, w, device=dev); lo = torch.empty(2, D, h, w, device=dev)
    pair(f"s{stage}.main conv11 + prob", lambda: ops.conv3d(y, small["conv11"], skip=sk, out=t), lambda: ops.conv3d(t2, huge["prob"], out=lo))
    pair(f"s{stage}.main conv11 + conv11", lambda: ops.conv3d(y, small["conv11"], skip=sk, out=t), lambda: ops.conv3d(y, huge["conv11"], skip=sk, out=t2))
    pair(f"s{stage}.main K1 + prob", lambda: ops.warp_corr(feats[0], feats[1:], p12, hyp, out=sim_out), lambda: ops.conv3d(t2, huge["prob"], out=lo))
