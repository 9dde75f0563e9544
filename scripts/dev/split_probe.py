"""The bf16-split probe: conv1 (8 -> 16, stride 2) at config 2's stage-2 main-pass shape through the fp32 MFMA
kernel (K3), the six-term and the three-term split kernels (csrc/conv3d_split.hip): layer time, error against a float64 convolution
and against ATen fp32, then the whole forward with every conv1 swapped (depth rel-L1 against the fp32 product, depth-maps/s).
    python scripts/dev/split_probe.py [--no-e2e]"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dmvsnet_amd import MVSNet, ops, synth  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=9):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


cfg = synth.CONFIGS["c2"]
net = MVSNet(cfg["ndepths"], cfg["ratios"], verbose=False)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
net = net.to(dev)
net.prepare(dev)
layer = net.cost_regularization[1]._packed[1]["conv1"]        # stage 2, `small` branch
w = net.cost_regularization[1].cosR_small.conv1.conv.weight.detach().cpu()

# accuracy on a volume small enough for a float64 reference on the CPU, activations of a realistic scale (post-ReLU inputs)
g = np.random.Generator(np.random.PCG64(3))
x = torch.from_numpy(np.maximum(g.standard_normal((8, 16, 148, 200), dtype=np.float32), 0))
y64 = F.conv3d(x[None].double(), w.double(), None, 2, 1)[0]
sc, sh = layer.scale.double().cpu().view(-1, 1, 1, 1), layer.shift.double().cpu().view(-1, 1, 1, 1)
truth = torch.relu(y64 * sc + sh)
aten = torch.relu(F.conv3d(x[None], w, None, 2, 1)[0] * sc.float() + sh.float())
scale = truth.abs().mean().item()
print(f"accuracy, 8x16x148x200 input, output mean |y| = {scale:.4f}")
for name, be in (("fp32 MFMA (K3)", "mfma"), ("bf16 split, 6 terms", "split6"), ("bf16 split, 3 terms", "split3")):
    y = ops.conv3d(x.to(dev), layer, backend=be).double().cpu()
    print(f"  {name:22s} vs float64: max abs {float((y - truth).abs().max()):.3e}  rel-L1 {float((y - truth).abs().mean() / truth.abs().mean()):.3e}"
          f"   vs ATen fp32: max abs {float((y - aten.double()).abs().max()):.3e}")
print(f"  {'ATen fp32 (CPU)':22s} vs float64: max abs {float((aten.double() - truth).abs().max()):.3e}  rel-L1 {float((aten.double() - truth).abs().mean() / truth.abs().mean()):.3e}")

# layer time at the stage-2 main-pass shape
xs = torch.randn((8, 32, 592, 800), device=dev)
out = torch.empty((16, 16, 296, 400), device=dev)
fl = 2.0 * 27 * 8 * 16 * 16 * 296 * 400
print("layer time, s2.main conv1 (8 -> 16, 32x592x800 -> 16x296x400):")
for name, be in (("fp32 MFMA (K3)", "mfma"), ("bf16 split, 6 terms", "split6"), ("bf16 split, 3 terms", "split3")):
    ms = timeit(lambda: ops.conv3d(xs, layer, backend=be, out=out))
    print(f"  {name:22s} {ms:.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s-equivalent (direct-form fp32 FLOPs / time)")
del xs, out

if "--no-e2e" not in sys.argv:
    imgs, proj, dv = synth.synth_inputs(cfg["H"], cfg["W"], cfg["V"], 0)
    imgs, dv, proj = imgs.to(dev), dv.to(dev), {k: v.to(dev) for k, v in proj.items()}
    net.return_prob_volume = net.return_depth_values = False
    res = {}
    for terms in (0, 6, 3):
        ops.split_probe = terms
        for _ in range(3):
            o = net(imgs, proj, dv)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            o = net(imgs, proj, dv)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        res[terms] = ([o[f"stage{s + 1}"]["depth"].double().cpu() for s in range(3)], 1.0 / dt)
    ops.split_probe = 0
    print("whole forward, config 2, every conv1 (12 launches per depth map) swapped:")
    for terms in (0, 6, 3):
        d, rate = res[terms]
        rel = [float((a - b).abs().mean() / b.abs().mean()) for a, b in zip(d, res[0][0])]
        print(f"  terms {terms}: {rate:6.2f} depth-maps/s   depth rel-L1 vs the fp32 product per stage: " + " ".join(f"{r:.2e}" for r in rel))
