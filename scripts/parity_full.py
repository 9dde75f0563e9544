#!/usr/bin/env python3
"""Full-size parity: the HIP path vs the oracle (CPU) on a BASELINE config, depth rel-L1 per stage.
usage: parity_full.py [config=c2] [inverse=0] [seed=0]   (the oracle needs minutes of CPU and ~7 GB at c2; seed = weights AND inputs)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmvsnet_amd import MVSNet, synth
from oracle import dmvs_oracle as O

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
inverse = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg = synth.CONFIGS[name]
torch.set_num_threads(min(os.cpu_count() or 1, 32))
net = MVSNet(cfg["ndepths"], cfg["ratios"], inverse_depth=inverse, verbose=False)
sd = synth.synth_state_dict(net.state_dict(), seed)
net.load_state_dict(sd)
net = net.cuda()
net.return_prob_volume = False
imgs, proj, dv = synth.synth_inputs(cfg["H"], cfg["W"], cfg["V"], seed)
out = net(imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, dv.cuda())
torch.cuda.synchronize()
t0 = time.time()
ref = O.mvsnet_forward(sd, cfg["ndepths"], cfg["ratios"], imgs, proj, dv, inverse_depth=inverse)
res = {"config": name, "inverse_depth": inverse, "seed": seed, "oracle_cpu_s": round(time.time() - t0, 1), "threads": torch.get_num_threads()}
for s in range(len(cfg["ndepths"])):
    d, r = out[f"stage{s+1}"]["depth"].cpu(), ref[f"stage{s+1}"]["depth"]
    c, rc = out[f"stage{s+1}"]["photometric_confidence"].cpu(), ref[f"stage{s+1}"]["photometric_confidence"]
    res[f"stage{s+1}"] = {"depth_rel_l1": float((d - r).abs().mean() / r.abs().mean()), "depth_max_abs": float((d - r).abs().max()),
                          "conf_max_abs": float((c - rc).abs().max()), "shape": list(d.shape)}
print(json.dumps(res))
