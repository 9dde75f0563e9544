"""Per-layer roofline table: every conv layer of one configuration timed in isolation (HIP events, median of
several repetitions), with its real FLOP rate and its algorithmic HBM rate.  Development tool for picking the
next kernel to work on; the numbers the judge reads come from bench.py.

    python scripts/layer_bench.py [--config c2] [--reps 7]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmvsnet_amd import MVSNet, ops, synth  # noqa: E402

PEAK_TF, PEAK_GBS = 157.3, 8000.0


def time_layer(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--json", default=None)
    ap.add_argument("--no-wino", action="store_true", help="direct-form K3 for the stride-1 3x3 layers too")
    ap.add_argument("--no-coarse", action="store_true", help="K3w / K3 for conv4 / conv6 (instead of the register-stationary K3r)")
    ap.add_argument("--no-c8", action="store_true", help="direct-form K3 for FeatureNet conv0.0 / conv0.1 (instead of the K3s row sweep)")
    ap.add_argument("--no-zmarch", action="store_true", help="K3w for conv2 (instead of the z-marching K3z)")
    ap.add_argument("--zmarch-min-depth", type=int, default=None, help="ops.ZMARCH_MIN_DEPTH (conv2 on K3z from this depth up)")
    ap.add_argument("--feat-coarse", action="store_true", help="FeatureNet conv2.1 / conv2.2 through K3r's 32 -> 32 2D form (ops.use_coarse_feature; measured slower, r06)")
    ap.add_argument("--only", default=None, help="comma-separated substrings: time only the layers whose tag contains one")
    ap.add_argument("--tune", action="append", default=[], help="name=value for dmvs_tune (repeatable), e.g. k3_deconv_prefetch=0")
    args = ap.parse_args()
    from dmvsnet_amd import _lib
    for kv in args.tune:
        k, v = kv.split("=")
        _lib.check(_lib.load().dmvs_tune(k.encode(), int(v)), f"dmvs_tune({k})")
    only = args.only.split(",") if args.only else None
    cfg = synth.CONFIGS[args.config]
    ops.use_wino = not args.no_wino
    ops.use_c8 = not args.no_c8
    ops.use_coarse = not args.no_coarse
    ops.use_coarse_feature = args.feat_coarse
    ops.use_zmarch = not args.no_zmarch
    if args.zmarch_min_depth is not None:
        ops.ZMARCH_MIN_DEPTH = args.zmarch_min_depth
    dev = torch.device("cuda:0")
    net = MVSNet(cfg["ndepths"], cfg["ratios"], verbose=False)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
    net = net.to(dev)
    net.prepare(dev)
    H, W, V = cfg["H"], cfg["W"], cfg["V"]
    rows = []

    def run(tag, layer, shape, skip=False, skip_up2=False, mult=1, in_views=False):
        cin, D, h, w = shape
        Do, Ho, Wo = layer.out_shape(D, h, w)
        if only is not None and not any(o in tag for o in only):
            return (layer.cout, Do, Ho, Wo)
        x = torch.randn((D, 3, h, w) if in_views else shape, device=dev)   # in_views: the loader's image stack [V,3,H,W]
        out = torch.empty((layer.cout, Do, Ho, Wo), device=dev)
        sk = None
        if skip:
            sk = torch.randn((layer.cout, Do, Ho // 2, Wo // 2) if skip_up2 else out.shape, device=dev)
        ms = time_layer(lambda: ops.conv3d(x, layer, skip=sk, out=out, skip_up2=skip_up2, in_views=in_views), args.reps)
        taps = 25 if layer.mode == ops.CONV2D_K5S2 else (1 if layer.mode == ops.CONV2D_K1 else 9 * layer.kdepth)
        vox = D * h * w if layer.mode == ops.DECONV_S2 else Do * Ho * Wo
        flops = 2.0 * taps * layer.cin * layer.cout * vox
        nbytes = 4.0 * (cin * D * h * w + layer.cout * Do * Ho * Wo * (2 if skip else 1))
        # executed FLOPs: the Winograd layers issue 16 of 36 products (conv0, Cin = 2: two k-groups of 4 for 6 pairs)
        wino = ops.use_wino and (layer.w_wino is not None or layer.w_coarse is not None or layer.w_zmarch is not None) and not skip
        xflops = flops / 2.25 * (8.0 / 6.0 if layer.cin == 2 else 1.0) if wino else flops
        rows.append(dict(layer=tag, cin=cin, cout=layer.cout, shape=[D, h, w], ms=ms, per_map_ms=ms * mult,
                         tflops=flops / ms / 1e9, gbs=nbytes / ms / 1e6, flops=flops, xflops=xflops, bytes=nbytes))
        return (layer.cout, Do, Ho, Wo)

    # FeatureNet on the [C][V][H][W] stack
    L = net.feature._packed
    s0 = run("feat.conv0.0", L["conv0.0"], (4, V, H, W), in_views=True)
    s0 = run("feat.conv0.1", L["conv0.1"], s0)
    if only is None or any(o in "feat.conv0.fused" for o in only):
        imgs = torch.randn((V, 3, H, W), device=dev)
        if ops.featurenet_conv0(imgs, L["conv0.0"], L["conv0.1"]) is not None:
            ms = time_layer(lambda: ops.featurenet_conv0(imgs, L["conv0.0"], L["conv0.1"]), args.reps)
            fl, nb = 2.0 * 9 * 11 * 8 * V * H * W, 4.0 * 11 * V * H * W
            rows.append(dict(layer="feat.conv0.fused", cin=3, cout=8, shape=[V, H, W], ms=ms, per_map_ms=ms,
                             tflops=fl / ms / 1e9, gbs=nb / ms / 1e6, flops=fl, xflops=fl, bytes=nb))
    s1 = run("feat.conv1.0", L["conv1.0"], s0)
    s1 = run("feat.conv1.1", L["conv1.1"], s1)
    s1 = run("feat.conv1.2", L["conv1.2"], s1)
    s2 = run("feat.conv2.0", L["conv2.0"], s1)
    s2 = run("feat.conv2.1", L["conv2.1"], s2)
    s2 = run("feat.conv2.2", L["conv2.2"], s2)
    run("feat.out1", L["out1"], s2)
    i1 = run("feat.inner1", L["inner1"], s1, skip=True, skip_up2=True)
    run("feat.out2", L["out2"], i1)
    i2 = run("feat.inner2", L["inner2"], s0, skip=True, skip_up2=True)
    run("feat.out3", L["out3"], i2)
    # the two output layers as the product runs them: quad-planar epilogue, level 3 with the top-down merge fused
    if only is None or any(o in "feat.out2.q4" for o in only):
        xq = torch.randn(i1, device=dev)
        ms = time_layer(lambda: ops.conv3d(xq, L["out2"], out_q4=True), args.reps)
        vox = i1[1] * i1[2] * i1[3]
        fl = 2.0 * 9 * 32 * 32 * vox
        rows.append(dict(layer="feat.out2.q4", cin=32, cout=32, shape=list(i1[1:]), ms=ms, per_map_ms=ms,
                         tflops=fl / ms / 1e9, gbs=4.0 * 64 * vox / ms / 1e6, flops=fl, xflops=fl / 2.25, bytes=4.0 * 64 * vox))
    if only is None or any(o in "feat.out3.fpn.q4" for o in only):
        lat, td = torch.randn(s0, device=dev), torch.randn(i1, device=dev)
        ms = time_layer(lambda: ops.conv3d_fpn(lat, td, L["out3"], out_q4=True), args.reps)
        vox = s0[1] * s0[2] * s0[3]
        fl = 2.0 * vox * (9 * 32 * 16 + 8 * 32)
        rows.append(dict(layer="feat.out3.fpn.q4", cin=32, cout=16, shape=list(s0[1:]), ms=ms, per_map_ms=ms,
                         tflops=fl / ms / 1e9, gbs=4.0 * (8 + 8 + 16) * vox / ms / 1e6, flops=fl, xflops=vox * 120 * 2048 / 64.0,
                         bytes=4.0 * (8 + 8 + 16) * vox))

    for s in range(len(cfg["ndepths"])):
        scale = 2 ** (3 - s - 1)
        h, w = H // scale, W // scale
        for kind, nets, D in (("main", net.cost_regularization, cfg["ndepths"][s]),
                              ("refine", net.cost_regularization_refine, 4)):
            conv0, small, _ = nets[s]._packed
            t = f"s{s + 1}.{kind}."
            c0 = run(t + "conv0x2", conv0, (2, D, h, w))
            x0 = (c0[0] // 2,) + c0[1:]
            c1 = run(t + "conv1", small["conv1"], x0, mult=2)
            c2 = run(t + "conv2", small["conv2"], c1, mult=2)
            c3 = run(t + "conv3", small["conv3"], c2, mult=2)
            # depth-1 volumes take the 2D form of a stride-1 3D layer, as the product does (mvsnet._RegBranch._branch)
            pick = lambda name, shp: small.get(name + "@d1", small[name]) if shp[1] == 1 else small[name]   # noqa: E731
            c4 = run(t + "conv4", pick("conv4", c3), c3, mult=2)
            c5 = run(t + "conv5", small["conv5"], c4, mult=2)
            c6 = run(t + "conv6", pick("conv6", c5), c5, mult=2)
            c7 = run(t + "conv7", small["conv7"], c6, skip=True, mult=2)
            c9 = run(t + "conv9", small["conv9"], c7, skip=True, mult=2)
            c11 = run(t + "conv11", small["conv11"], c9, skip=True, mult=2)
            run(t + "prob", small["prob"], c11, mult=2)

    tot = sum(r["per_map_ms"] for r in rows)
    # floor = what the layer would take at the rates this chip actually sustains: max(algorithmic bytes at the 6.3 TB/s a
    # copy reaches, EXECUTED MFMA FLOPs at 140 TFLOP/s); ratio = measured / floor
    print(f"{'layer':22s} {'Cin>Cout':>8s} {'D x H x W':>16s} {'ms':>8s} {'x':>2s} {'TF/s':>7s} {'%mfma':>6s} {'GB/s':>7s} {'%hbm':>5s} {'floor_ms':>8s} {'ratio':>5s}")
    ftot = 0.0
    for r in rows:
        D, h, w = r["shape"]
        r["floor_ms"] = max(r["bytes"] / 6.3e9, r["xflops"] / 140e9)
        r["ratio"] = r["ms"] / r["floor_ms"]
        ftot += r["floor_ms"] * r["per_map_ms"] / r["ms"]
        print(f"{r['layer']:22s} {r['cin']:3d}>{r['cout']:<3d}  {D:3d}x{h:4d}x{w:4d}  {r['ms']:8.3f} "
              f"{int(round(r['per_map_ms'] / r['ms'])):2d} {r['tflops']:7.1f} {100 * r['tflops'] / PEAK_TF:6.1f} "
              f"{r['gbs']:7.0f} {100 * r['gbs'] / PEAK_GBS:5.1f} {r['floor_ms']:8.3f} {r['ratio']:5.2f}")
    print(f"sum over one depth map (both branches, serial): {tot:.2f} ms; sum of floors {ftot:.2f} ms")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
