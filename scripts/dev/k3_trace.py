#!/usr/bin/env python3
"""Phase timeline of the K3 conv / deconv kernels (dev): needs the trace build, DMVS_LIB=/tmp/libdmvs_k3trace.so
(scripts/dev/k3_trace.sh).  For every chosen layer of config 2: per-workgroup medians of the phases (s_memtime ticks) and,
per CU, how the lives of its workgroups overlap -- what VERDICT r03 item 1 asks for: do the loads run UNDER the MFMAs of
co-resident workgroups, or do the phases add up?

    scripts/dev/k3_trace.sh [layer ...]        layers: s2.conv1 s2.conv11 s2.conv9 s2.conv3 s3.conv1 s3.conv11 f.conv0.1 ...
"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dmvsnet_amd import MVSNet, _lib, ops, synth  # noqa: E402

lib = _lib.load()
lib.dmvs_dev_trace_k3.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
cfg = synth.CONFIGS["c2"]
net = MVSNet(cfg["ndepths"], cfg["ratios"], verbose=False)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
net = net.to(dev)
net.prepare(dev)
H, W, V = cfg["H"], cfg["W"], cfg["V"]
NWG = 65536
trace = torch.zeros(NWG * 16, dtype=torch.int64, device=dev)


def shapes(stage):
    sc = 2 ** (3 - stage)
    D, h, w = cfg["ndepths"][stage - 1], H // sc, W // sc
    return {"conv1": (8, D, h, w), "conv3": (16, D // 2, h // 2, w // 2), "conv5": (32, D // 4, h // 4, w // 4),
            "conv7": (64, D // 8, h // 8, w // 8), "conv9": (32, D // 4, h // 4, w // 4), "conv11": (16, D // 2, h // 2, w // 2)}


def run(tag):
    if tag.startswith("f."):
        layer = net.feature._packed[tag[2:]]
        shape = {"conv0.0": (4, V, H, W), "conv0.1": (8, V, H, W), "conv1.0": (8, V, H, W), "conv2.0": (16, V, H // 2, W // 2)}[tag[2:]]
        skip = False
    else:
        st, name = tag.split(".")
        stage = int(st[1])
        layer = net.cost_regularization[stage - 1]._packed[1][name]
        shape = shapes(stage)[name]
        skip = layer.mode == ops.DECONV_S2
    x = torch.randn(shape, device=dev)
    Do, Ho, Wo = layer.out_shape(*shape[1:])
    out = torch.empty((layer.cout, Do, Ho, Wo), device=dev)
    sk = torch.randn_like(out) if skip else None
    saved = ops.use_wino
    ops.use_wino = False
    try:
        for _ in range(3):
            ops.conv3d(x, layer, skip=sk, out=out)
        torch.cuda.synchronize()
        trace.zero_()
        lib.dmvs_dev_trace_k3(ctypes.c_void_p(trace.data_ptr()))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.conv3d(x, layer, skip=sk, out=out); b.record()
        torch.cuda.synchronize()
        lib.dmvs_dev_trace_k3(None)
    finally:
        ops.use_wino = saved
    tr = trace.view(-1, 16).cpu()
    wg = torch.arange(tr.shape[0])
    ok = (tr[:, 0] > 0) & (tr[:, 4] > 0)
    tr, wg = tr[ok], wg[ok]
    n = len(tr)
    # every XCD has its own s_memtime counter: workgroup id % 8 is the XCD (common.h), shift each XCD's stamps to its first
    xcd = (tr[:, 14] & 0xf) if (tr[:, 14] != 0).any() else wg % 8   # XCC_ID register (trace slot 14)
    print("   workgroups per XCC_ID:", [int((xcd == k).sum()) for k in range(8)], " agreement with id % 8:", float((xcd == wg % 8).float().mean()))
    base = torch.zeros(8, dtype=torch.int64)
    for k in range(8):
        if (xcd == k).any():
            base[k] = tr[xcd == k, 0].min()
    tr = tr.clone()
    tr[:, 0:5] -= base[xcd][:, None] * (tr[:, 0:5] > 0)
    tr[:, 0] += 1
    t0, t1, t2, t3, t4 = (tr[:, i].double() for i in range(5))
    wait, mfma = tr[:, 8].double(), tr[:, 9].double()
    life = t4 - t0
    hw = tr[:, 15]
    # HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (+ xcc via XCC_ID elsewhere): CU key within an XCD
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 0x7) << 5)
    med = lambda v: float(v.median())
    print(f"{tag}: {tuple(shape)} -> {layer.cout} ch, {a.elapsed_time(b):.3f} ms, {n} workgroups traced")
    print(f"   per workgroup (median ticks): life {med(life):.0f} | first chunk lands {med(t1 - t0):.0f} | chunk loop {med(t2 - t1):.0f} "
          f"(of it: waits+barriers {med(wait) - med(t1 - t0):.0f}, MFMA sections {med(mfma):.0f}) | epilogue until stores issued "
          f"{med(torch.where(t3 > 0, t3, t2) - t2):.0f} | stores retire {med(t4 - torch.where(t3 > 0, t3, t2)):.0f}")
    # (no cross-workgroup analysis: the s_memtime counters of different CUs / XCDs are not synchronised -- measured r04:
    # "spans" of 1e7 .. 1e10 ticks for a 0.2 ms kernel -- so only differences inside one workgroup mean anything)
    print(f"   share of a workgroup's life: waits at the chunk barriers {float(wait.sum() / life.sum()):.2f}, MFMA sections "
          f"{float(mfma.sum() / life.sum()):.2f}, epilogue + store retirement {float((t4 - t2).sum() / life.sum()):.2f}")
    return tr


tags = [a for a in sys.argv[1:] if "=" not in a]
for kv in (a for a in sys.argv[1:] if "=" in a):      # name=value: dmvs_tune knobs
    k, v = kv.split("=")
    _lib.check(lib.dmvs_tune(k.encode(), int(v)), f"dmvs_tune({k})")
    print("tune", k, v)
for tag in (tags or ["s2.conv1", "s2.conv11", "s2.conv9", "s2.conv3"]):
    run(tag)
