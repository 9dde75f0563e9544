import sys, os, torch
sys.path.insert(0, os.getcwd())
from dmvsnet_amd import MVSNet, ops, synth
cfg = synth.CONFIGS["c2"]
net = MVSNet(cfg["ndepths"], cfg["ratios"], verbose=False)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
net = net.cuda(); net.prepare(torch.device("cuda:0"))
L = net.feature._packed
V, H, W = 5, 1184, 1600
c0 = torch.randn(8, V, H, W, device="cuda"); intra2 = torch.randn(32, V, H // 2, W // 2, device="cuda")
def t(fn, n=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[n // 2]
def sep():
    i3 = ops.conv3d(c0, L["inner2"], skip=intra2, skip_up2=True)
    return ops.conv3d(i3, L["out3"], out_q4=True)
def fused():
    return ops.conv3d_fpn(c0, intra2, net.feature._inner2_w, net.feature._inner2_b, L["out3"], out_q4=True)
print("separate %.3f ms   fused %.3f ms" % (t(sep), t(fused)))
a, b = sep(), fused()
print("max abs diff", (a - b).abs().max().item())
