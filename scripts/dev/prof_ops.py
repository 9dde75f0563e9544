import sys, os, torch
sys.path.insert(0, os.getcwd())
from dmvsnet_amd import MVSNet, synth
cfg = synth.CONFIGS["c2"]
net = MVSNet(cfg["ndepths"], cfg["ratios"], verbose=False)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0)); net = net.cuda(); net.return_prob_volume = False
imgs, proj, dv = synth.synth_inputs(cfg["H"], cfg["W"], cfg["V"], 0)
imgs, dv = imgs.cuda(), dv.cuda(); proj = {k: v.cuda() for k, v in proj.items()}
for _ in range(3): net(imgs, proj, dv)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=False) as p:
    net(imgs, proj, dv); torch.cuda.synchronize()
rows = [(e.key, e.count, e.cpu_time_total) for e in p.key_averages() if e.key.startswith("aten::")]
for k, c, t in sorted(rows, key=lambda r: -r[2])[:25]: print(f"{k:40s} {c:5d} {t:10.0f} us")
