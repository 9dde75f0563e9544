#!/usr/bin/env python3
"""Phase timeline of the q4 K1 kernel (dev): needs the trace build, DMVS_LIB=/tmp/libdmvs_trace.so (scripts/dev/k1_trace.sh).
Per stage-pass of the smooth benchmark: median s_memtime deltas between the kernel's phases over its workgroups."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dmvsnet_amd import _lib, ops, synth  # noqa: E402
lib = _lib.load()
lib.dmvs_dev_trace.argtypes = [ctypes.c_void_p]
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = synth.CONFIGS["c2"]
H, W, V = cfg["H"], cfg["W"], cfg["V"]
dev = "cuda:0"
cams = synth.synth_cameras(H, W, V)
dv = synth.synth_depth_values().to(dev)
g = torch.Generator(device="cpu").manual_seed(0)
last = None
trace = torch.zeros(16384 * 32, dtype=torch.int64, device=dev)
names = ["prologue->bar", "table->barA", "stage issue", "tapinfo", "wait vmcnt", "barrier B", "sample v0", "view1", "view2", "view3", "tail", "store"]
for s in range(3):
    sc = 2 ** (2 - s)
    h, w, C, D = H // sc, W // sc, (32, 16, 8)[s], cfg["ndepths"][s]
    feats = [ops.hwc_to_q4(torch.randn(h, w, C, generator=g).to(dev)) for _ in range(V)]
    p12 = ops.relative_proj(cams[f"stage{s + 1}"][0].to(dev).contiguous())
    if s == 0:
        hyp, _ = ops.hypotheses_first(dv, D, h, w, False, True)
    else:
        hyp, _ = ops.hypotheses_next(last, dv, float(cfg["ratios"][s]), D, False, True)
    yy, xx = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
    last = (650.0 + 100.0 * torch.sin(xx / w * 6.0) + 50.0 * torch.cos(yy / h * 4.0)).float().contiguous()
    for _ in range(2):
        ops.warp_corr(feats[0], feats[1:], p12, hyp, variant=variant)
    torch.cuda.synchronize()
    trace.zero_()
    lib.dmvs_dev_trace(ctypes.c_void_p(trace.data_ptr()))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.warp_corr(feats[0], feats[1:], p12, hyp, variant=variant); b.record()
    torch.cuda.synchronize()
    lib.dmvs_dev_trace(None)
    tr = trace.view(-1, 32).cpu()
    ok = (tr[:, 0] > 0) & (tr[:, 12] > 0)
    tr = tr[ok]
    d = (tr[:, 1:13] - tr[:, 0:12]).double()
    tot = (tr[:, 12] - tr[:, 0]).double()
    span = (tr[:, 12].max() - tr[:, 0].min()).item()
    print(f"pass s{s+1}.main C={C} D={D}: {a.elapsed_time(b):.3f} ms, {len(tr)} workgroups traced, kernel span {span} ticks; per-workgroup total median {tot.median().item():.0f} ticks")
    print("   " + "  ".join(f"{n}: {d[:, i].median().item():.0f}" for i, n in enumerate(names)))
    hw = tr[:, 31]
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 0x7) << 5)
    print("   distinct (se,sh,cu) ids seen:", len(set(cu.tolist())))
