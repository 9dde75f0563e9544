#!/usr/bin/env python3
"""Where a K3r (csrc/conv3d_coarse.hip) launch spends its time (dev; needs the trace build, scripts/dev/k3r_trace.sh):
per-wave sums of s_memtime ticks in the phases of the stage pipeline, averaged over the waves that had work.

    scripts/dev/k3r_trace.sh [cin,kd,D,H,W ...]      default: the config-2 coarse shapes
"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dmvsnet_amd import _lib, ops  # noqa: E402

lib = _lib.load()
lib.dmvs_dev_trace_k3r.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
NAMES = ["wait+barrier", "issue", "finish", "compute", "partial", "kernel", "prologue", "stages"]
cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [
    (64, 3, 8, 37, 50), (64, 3, 4, 74, 100), (64, 1, 1, 148, 200), (64, 1, 1, 37, 50), (32, 3, 16, 74, 100), (32, 3, 8, 148, 200),
    (32, 3, 2, 296, 400), (32, 1, 1, 296, 400), (32, 1, 1, 74, 100)]
for cin, kd, D, H, W in cases:
    w = torch.randn((cin, cin) + ((3, 3, 3) if kd == 3 else (3, 3))) * 0.05
    layer = ops.ConvLayer("t", ops.CONV_S1, kd, cin, cin, None, None, torch.ones(cin, device=dev), torch.zeros(cin, device=dev), True,
                          w_coarse=ops.pack_coarse(w, cin, cin, kd).to(dev))
    x = torch.randn(cin, D, H, W, device=dev)
    out = torch.empty_like(x)
    for _ in range(3):
        ops.conv3d(x, layer, out=out, backend="coarse")
    torch.cuda.synchronize()
    trace = torch.zeros(4096 * 64, dtype=torch.int64, device=dev)
    lib.dmvs_dev_trace_k3r(ctypes.c_void_p(trace.data_ptr()))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.conv3d(x, layer, out=out, backend="coarse"); b.record()
    torch.cuda.synchronize()
    lib.dmvs_dev_trace_k3r(None)
    t = trace.view(4096, 8, 8)[:256].cpu().double()
    busy = t[:, :, 5] > 0
    n = int(busy[:, 0].sum())
    m = t[busy].mean(0)
    mx = t[busy].max(0).values
    print(f"{cin}>{cin} kd={kd} {D}x{H}x{W}: {a.elapsed_time(b) * 1e3:.1f} us, {n} workgroups with work, stages/wave mean {m[7]:.1f} max {mx[7]:.0f}")
    print("   mean ticks per wave: " + "  ".join(f"{NAMES[i]}={m[i]:.0f}" for i in range(7)))
    print("   per stage:           " + "  ".join(f"{NAMES[i]}={m[i] / max(m[7], 1):.0f}" for i in range(5)))
    print(f"   longest wave: kernel={mx[5]:.0f} ticks")
