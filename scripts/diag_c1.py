"""Diagnostic: config-c1 intermediates, HIP path vs oracle (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dmvsnet_amd import MVSNet, ops, synth
from oracle import dmvs_oracle as O

cfgname = sys.argv[1] if len(sys.argv) > 1 else "c1"
cfg = synth.CONFIGS[cfgname]
seed = 0
net = MVSNet(cfg["ndepths"], cfg["ratios"], verbose=False)
sd = synth.synth_state_dict(net.state_dict(), seed)
net.load_state_dict(sd); net = net.cuda()
imgs, proj, dv = synth.synth_inputs(cfg["H"], cfg["W"], cfg["V"], seed)
ref = O.mvsnet_forward(sd, cfg["ndepths"], cfg["ratios"], imgs, proj, dv, keep_intermediates=True)
out = net(imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, dv.cuda())

def cmp(name, a, b):
    a = a.detach().cpu().float(); b = b.detach().cpu().float()
    a = a.reshape(b.shape)
    d = (a - b).abs()
    print(f"{name:34s} max|d| {d.max():.3e} mean|d| {d.mean():.3e} rel-L1 {d.mean()/b.abs().mean():.3e}  (ref mean|.| {b.abs().mean():.3e})")

# features
feats_ref = ref["_features"]
f_gpu = net.feature(imgs[:, 1].cuda())
for s in range(3):
    full = torch.cat((feats_ref[1][f"stage{s+1}"], feats_ref[1][f"stage{s+1}_c"]), 1)
    cmp(f"feature view1 stage{s+1}", f_gpu[s], full)
for s in range(len(cfg["ndepths"])):
    st, rs = out[f"stage{s+1}"], ref[f"stage{s+1}"]
    for k in ("depth_values", "depth_sub_plus", "depth_values_c", "photometric_confidence", "depth_sub_plus_refine", "depth", "photometric_confidence_refine"):
        cmp(f"stage{s+1}.{k}", st[k], rs[k])
    # re-run the pieces on the oracle's inputs to isolate each kernel
    mid = ref["_intermediates"][f"stage{s+1}"]
    net.prepare(torch.device("cuda:0"))
    lg = net.cost_regularization[s].run(mid["sim"][0].cuda().contiguous(), "direct")
    cmp(f"stage{s+1} costreg(direct) on ref sim", lg, mid["logits"][0])
    lgc = net.cost_regularization_refine[s].run(mid["sim_c"][0].cuda().contiguous(), "direct")
    cmp(f"stage{s+1} costreg_refine on ref sim_c", lgc, mid["logits_c"][0])
    C = feats_ref[0][f"stage{s+1}"].shape[1]
    hwc = lambda f: ops.hwc_to_q4(f[0].permute(1, 2, 0).contiguous().cuda())
    p12 = ops.relative_proj(proj[f"stage{s+1}"][0].cuda().contiguous())
    sim = ops.warp_corr(hwc(feats_ref[0][f"stage{s+1}"]), [hwc(feats_ref[v][f"stage{s+1}"]) for v in range(1, cfg["V"])], p12, rs["depth_values"][0].cuda().contiguous())
    cmp(f"stage{s+1} warp_corr on ref feats", sim, mid["sim"][0])
    dsp, hy, conf, _ = ops.depth_regress(mid["logits"][0].cuda().contiguous(), rs["depth_values"][0].cuda().contiguous(), rs["interval"].cuda(), 1.0, 0, False)
    cmp(f"stage{s+1} depth_regress on ref logits", dsp, rs["depth_sub_plus"][0])
    cmp(f"stage{s+1}   hyps", hy, rs["depth_values_c"][0])
