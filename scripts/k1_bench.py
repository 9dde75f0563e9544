#!/usr/bin/env python3
"""K1 alone: the six stage-passes of a config with synthetic features and plausible (smooth) hypothesis planes, HIP-event
timed (median of --reps), algorithmic GB/s per pass and for the depth map.  --layout hwc times the generic pixel-major
kernel instead of the product's quad-planar one; --variant = the q4 kernel's launch knob (dmvs_warp_corr_q4).
    python scripts/k1_bench.py [--config c2] [--reps 9] [--layout q4|hwc] [--variant 0] [--save out.pt]"""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmvsnet_amd import ops, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--save", default=None)
    ap.add_argument("--layout", default="q4", choices=["q4", "hwc"])
    ap.add_argument("--variant", type=int, default=0)
    args = ap.parse_args()
    cfg = synth.CONFIGS[args.config]
    H, W, V = cfg["H"], cfg["W"], cfg["V"]
    dev = "cuda:0"
    cams = synth.synth_cameras(H, W, V)
    dv = synth.synth_depth_values().to(dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    res, outs = [], {}
    tot_ms = tot_b = 0.0
    last = None
    for s in range(3):
        sc = 2 ** (2 - s)
        h, w, C, D = H // sc, W // sc, (32, 16, 8)[s], cfg["ndepths"][s]
        feats = [torch.randn(h, w, C, generator=g).to(dev) for _ in range(V)]
        if args.layout == "q4":
            feats = [ops.hwc_to_q4(f) for f in feats]
        p12 = ops.relative_proj(cams[f"stage{s + 1}"][0].to(dev).contiguous())
        if s == 0:
            hyp, _ = ops.hypotheses_first(dv, D, h, w, False, True)
        else:
            hyp, _ = ops.hypotheses_next(last, dv, float(cfg["ratios"][s]), D, False, True)
        yy, xx = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
        last = (650.0 + 100.0 * torch.sin(xx / w * 6.0) + 50.0 * torch.cos(yy / h * 4.0)).float().contiguous()
        spread = (8.0, 4.0, 2.0)[s]
        hyp_c = (last[None] + (torch.arange(4, device=dev).view(4, 1, 1) - 1.5) * spread).contiguous()
        for name, hy in (("main", hyp), ("refine", hyp_c)):
            Dp = hy.shape[0]
            fn = lambda: ops.warp_corr(feats[0], feats[1:], p12, hy, layout=args.layout, variant=args.variant)  # noqa: E731
            out = fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(args.reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fn(); b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            ts.sort()
            ms = ts[len(ts) // 2]
            nbytes = 4.0 * (V * C * h * w + 2 * Dp * h * w + (h * w if isinstance(hy, ops.AffinePlanes) else Dp * h * w))
            res.append({"pass": f"s{s + 1}.{name}", "C": C, "D": Dp, "HxW": f"{h}x{w}", "ms": round(ms, 4),
                        "GBps": round(nbytes / ms / 1e6, 1), "finite": bool(torch.isfinite(out).all())})
            tot_ms += ms
            tot_b += nbytes
            outs[f"s{s + 1}.{name}"] = out.cpu()
    print(json.dumps({"kernel": args.layout, "variant": args.variant, "passes": res, "ms_per_map": round(tot_ms, 4),
                      "GBps": round(tot_b / tot_ms / 1e6, 1), "hbm_frac": round(tot_b / tot_ms / 1e6 / 8000.0, 4)}))
    if args.save:
        torch.save(outs, args.save)


if __name__ == "__main__":
    main()
