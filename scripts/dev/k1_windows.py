#!/usr/bin/env python3
"""Staging-window statistics of K1 on the REAL pipeline (bench config, synthetic weights): hooks ops.warp_corr, projects
every hypothesis plane of every stage-pass with torch and reduces per (tile, plane chunk, view) bounding boxes for a
few tile shapes.  Output: fraction of windows that fit a given LDS budget, mean / p95 window bytes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dmvsnet_amd import CostAgg, MVSNet, ops, synth  # noqa: E402

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
    calls.append((ref.shape[-1], p12.clone(), depth.volume() if isinstance(depth, ops.AffinePlanes) else depth.clone()))
    return orig(ref, src, p12, depth, *a, **k)


ops.warp_corr = hook
net(imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, dv.cuda())
torch.cuda.synchronize()


def stats(C, p12, depth, TW, TH, DC, pad_b):
    D, H, W = depth.shape
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda", dtype=torch.float32), torch.arange(W, device="cuda", dtype=torch.float32), indexing="ij")
    nty, ntx = -(-H // TH), -(-W // TW)
    sizes = []
    for v in range(p12.shape[0]):
        P = p12[v]
        rx = P[0] * xs + P[1] * ys + P[2]; ry = P[3] * xs + P[4] * ys + P[5]; rz = P[6] * xs + P[7] * ys + P[8]
        for d0 in range(0, D, DC):
            dd = depth[d0:d0 + DC]
            pz = rz * dd + P[11]
            ix = ((rx * dd + P[9]) / pz).clamp(-1, W); iy = ((ry * dd + P[10]) / pz).clamp(-1, H)

            def red(a, fn):
                a = fn(a, 0)
                a = torch.nn.functional.pad(a[None, None], (0, ntx * TW - W, 0, nty * TH - H), mode="replicate")[0, 0]
                return fn(fn(a.view(nty, TH, ntx, TW), 3), 1)
            mn = lambda a, d: a.min(d).values  # noqa: E731
            mx = lambda a, d: a.max(d).values  # noqa: E731
            x0 = red(ix, mn).floor().clamp(0, W - 1); x1 = (red(ix, mx).floor() + 1).clamp(0, W - 1)
            y0 = red(iy, mn).floor().clamp(0, H - 1); y1 = (red(iy, mx).floor() + 1).clamp(0, H - 1)
            sizes.append(((x1 - x0 + 1) * (y1 - y0 + 1)).flatten())
    kb = torch.cat(sizes) * pad_b / 1024
    fits = " ".join(f"<={c}KB:{(kb <= c).float().mean().item() * 100:.0f}%" for c in (8, 12, 16, 20, 24, 32, 40))
    return f"mean {kb.mean().item():.1f} KB p95 {kb.quantile(0.95).item():.1f} | {fits}"


for i, (C, p12, depth) in enumerate(calls):
    pad_b = {32: 80, 16: 80, 8: 48}[C]
    for TW, TH in ((32, 8), (16, 8), (32, 4), (16, 4), (8, 8)):
        for DC in ((4,) if depth.shape[0] <= 4 else (4, 8)):
            print(f"pass {i} C={C} D={depth.shape[0]} {depth.shape[1]}x{depth.shape[2]} tile {TW}x{TH} DC={DC}: {stats(C, p12, depth, TW, TH, DC, pad_b)}")
