#!/usr/bin/env python3
"""Which window mode every (tile, plane chunk, view) of the q4 K1 kernel takes on the REAL inputs of the bench configuration
(makes the LDS conflict factor attributable).  Needs the trace build: scripts/dev/k1_trace.sh modes."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dmvsnet_amd import MVSNet, _lib, ops, synth  # noqa: E402
lib = _lib.load()
lib.dmvs_dev_modes.argtypes = [ctypes.c_void_p]
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = synth.CONFIGS[name]
net = MVSNet(cfg["ndepths"], cfg["ratios"], inverse_depth=cfg.get("inverse", False), verbose=False)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
net = net.cuda()
net.return_prob_volume = False
imgs, proj, dv = synth.synth_inputs(cfg["H"], cfg["W"], cfg["V"], 0)
counts = torch.zeros(16, dtype=torch.int64, device="cuda")
orig = ops.warp_corr
rows = []


def hook(ref, src, p12, depth, *a, **k):
    torch.cuda.synchronize()
    counts.zero_()
    lib.dmvs_dev_modes(ctypes.c_void_p(counts.data_ptr()))
    out = orig(ref, src, p12, depth, *a, **k)
    torch.cuda.synchronize()
    lib.dmvs_dev_modes(None)
    D, H, W = depth.shape
    rows.append((4 * ref.shape[0], D, H, W, len(src), counts.cpu().tolist()))
    return out


ops.warp_corr = hook
net(imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, dv.cuda())
ops.warp_corr = orig
print(f"# K1 window modes on the real inputs of config {name} (random-weight network), per stage-pass: share of the (tile, plane chunk,")
print("# view) triples per mode -- mode m stages the window in 2^m channel slabs (m = 0: all quad planes at once) -- and the mean")
print("# window size in 16-byte pieces per quad plane; 'global' = exact global-tap path (corner bound failed or window > LDS)")
for C, D, H, W, nsrc, c in rows:
    tot = sum(c[:5]) or 1
    parts = []
    for m, lab in enumerate(("mode0", "mode1", "mode2", "mode3", "global")):
        if c[m]:
            parts.append(f"{lab} {100.0 * c[m] / tot:5.1f} % (mean {c[8 + m] / c[m]:6.0f} pieces)")
    print(f"C={C:2d} D={D:2d} {H}x{W} views={nsrc}: {tot} triples: " + "  ".join(parts))
