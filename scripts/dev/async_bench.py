import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from dmvsnet_amd import MVSNet, synth
cfg = synth.CONFIGS["c2"]
net = MVSNet(cfg["ndepths"], cfg["ratios"], verbose=False)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0)); net = net.cuda(); net.return_prob_volume = False
imgs, proj, dv = synth.synth_inputs(cfg["H"], cfg["W"], cfg["V"], 0)
imgs, dv = imgs.cuda(), dv.cuda(); proj = {k: v.cuda() for k, v in proj.items()}
def run(n):
    for _ in range(n): net(imgs, proj, dv)
    torch.cuda.synchronize()
for mode in (False, True, False, True):
    net.feature_async_topdown = mode
    run(5); t = time.perf_counter(); run(20); dt = time.perf_counter() - t
    print("async_topdown", mode, "%.2f maps/s %.3f ms" % (20 / dt, dt / 20 * 1e3))
