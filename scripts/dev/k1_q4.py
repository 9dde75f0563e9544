#!/usr/bin/env python3
"""A/B of the K1 kernels: the generic pixel-major kernel (--hwc) and the quad-planar kernel's launch variants, (a) on smooth
synthetic planes (what scripts/k1_bench.py times) and (b) on the REAL inputs of every stage-pass of the bench
configuration (random-weight network: incoherent hypotheses).  (r03's A/B against r02's four LDS-window kernels, since
removed, is profiles/r03_a_k1_ab.txt.)
    python scripts/dev/k1_q4.py [c2] [--q4 0,8,16,2,3] [--hwc]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dmvsnet_amd import MVSNet, _lib, ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("config", nargs="?", default="c2")
ap.add_argument("--q4", default="0,8,2,10,3,11")
ap.add_argument("--no-old", action="store_true", help="(kept for old command lines; the r02 kernels are gone)")
ap.add_argument("--hwc", action="store_true", help="also time the generic pixel-major kernel")
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--part", default="both", choices=["both", "smooth", "real"])
args = ap.parse_args()
cfg = synth.CONFIGS[args.config]
lib = _lib.load()
q4_vars = [int(v) for v in args.q4.split(",")]
old_vars = [0] if args.hwc else []
names = {0: "hwc_generic"}


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


def ab(label, calls):
    tot = {}
    for i, (ref, src, p12, depth) in enumerate(calls):
        rq, sq = ops.hwc_to_q4(ref), [ops.hwc_to_q4(s) for s in src]
        row, base = [], None
        for var in old_vars:
            fn = lambda: ops.warp_corr(ref, src, p12, depth, layout="hwc")  # noqa: E731
            t = timed(fn)
            if base is None:
                base = fn()
            tot[names[var]] = tot.get(names[var], 0.0) + t
            row.append(f"{names[var]} {t:.4f}")
        if base is None:
            base = ops.warp_corr(ref, src, p12, depth, layout="hwc")
        diff = 0.0
        # ORDER-BALANCED (r06): the first kernel timed on a set of inputs runs 5-15 % slower than the same kernel a few launches later
        # (clocks / caches settle; profiles/r06_u_k1_stagger_balanced.txt) -- a sweep that times its variants one after the other
        # credits that drift to whatever comes later in the list.  So: one untimed round over all variants, then the variants forwards,
        # backwards, forwards, backwards; a variant's time is the median of its four medians.
        fns = {var: (lambda var=var: ops.warp_corr(rq, sq, p12, depth, layout="q4", variant=var)) for var in dict.fromkeys(q4_vars)}
        for fn in fns.values():
            fn(); fn()
        torch.cuda.synchronize()
        seen = {var: [] for var in fns}
        for rnd_ in range(4):
            for var in (list(fns) if rnd_ % 2 == 0 else list(fns)[::-1]):
                seen[var].append(timed(fns[var]))
        for var, fn in fns.items():
            ts = sorted(seen[var])
            t = 0.5 * (ts[1] + ts[2])
            diff = max(diff, (fn() - base).abs().max().item())
            tot[f"q4.{var}"] = tot.get(f"q4.{var}", 0.0) + t
            row.append(f"q4.{var} {t:.4f}")
        D, H, W = depth.shape
        print(f"[{label}] pass {i} C={ref.shape[-1]} D={D} {H}x{W}: " + "  ".join(row) + f"  max|q4-hwc| {diff:.2e}", flush=True)
    print(f"[{label}] total ms: " + "  ".join(f"{k} {v:.3f}" for k, v in tot.items()), flush=True)


# (a) smooth planes, as scripts/k1_bench.py
H, W, V = cfg["H"], cfg["W"], cfg["V"]
dev = "cuda:0"
cams = synth.synth_cameras(H, W, V)
dv = synth.synth_depth_values().to(dev)
g = torch.Generator(device="cpu").manual_seed(0)
calls, last = [], None
for s in range(3):
    sc = 2 ** (2 - s)
    h, w, C, D = H // sc, W // sc, (32, 16, 8)[s], cfg["ndepths"][s]
    feats = [torch.randn(h, w, C, generator=g).to(dev) for _ in range(V)]
    p12 = ops.relative_proj(cams[f"stage{s + 1}"][0].to(dev).contiguous())
    if s == 0:
        hyp, _ = ops.hypotheses_first(dv, D, h, w, False, True)
    else:
        hyp, _ = ops.hypotheses_next(last, dv, float(cfg["ratios"][s]), D, False, True)
    yy, xx = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
    last = (650.0 + 100.0 * torch.sin(xx / w * 6.0) + 50.0 * torch.cos(yy / h * 4.0)).float().contiguous()
    spread = (8.0, 4.0, 2.0)[s]
    hyp_c = (last[None] + (torch.arange(4, device=dev).view(4, 1, 1) - 1.5) * spread).contiguous()
    calls.append((feats[0], feats[1:], p12, hyp))
    calls.append((feats[0], feats[1:], p12, hyp_c))
if args.part != "real":
    ab("smooth", calls)
del calls, feats
if args.part == "smooth":
    sys.exit(0)

# (b) the real pipeline's inputs
net = MVSNet(cfg["ndepths"], cfg["ratios"], verbose=False)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
net = net.cuda()
net.return_prob_volume = False
imgs, proj, dv = synth.synth_inputs(cfg["H"], cfg["W"], cfg["V"], 0)
calls = []
orig = ops.warp_corr


def hook(ref, src, p12, depth, *a, **k):
    calls.append((ref, list(src), p12.clone(), depth))
    return orig(ref, src, p12, depth, *a, **k)


ops.warp_corr = hook
net(imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, dv.cuda())
torch.cuda.synchronize()
ops.warp_corr = orig
if calls and calls[0][0].dim() == 4:   # the product already runs quad-planar features: back to pixel-major for the A/B
    q2h = lambda t: t.permute(1, 2, 0, 3).reshape(t.shape[1], t.shape[2], -1).contiguous()  # noqa: E731
    calls = [(q2h(r), [q2h(x) for x in s], p, d) for r, s, p, d in calls]
ab("real", calls)
