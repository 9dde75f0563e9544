import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from dmvsnet_amd import MVSNet, synth
cfg = synth.CONFIGS["c2"]
net = MVSNet(cfg["ndepths"], cfg["ratios"], verbose=False)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0)); net = net.cuda(); net.return_prob_volume = False
imgs, proj, dv = synth.synth_inputs(cfg["H"], cfg["W"], cfg["V"], 0)
imgs, dv = imgs.cuda(), dv.cuda(); proj = {k: v.cuda() for k, v in proj.items()}
torch.cuda.synchronize()
ts = []
for i in range(80):
    t = time.perf_counter(); net(imgs, proj, dv); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print(" ".join(f"{x:.1f}" for x in ts))
