#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports /root/reference/networks (read-only, never copied) and evaluates it on the
deterministic synthetic inputs/weights of dmvsnet_amd.synth.  Only data (inputs'
checksums + expected outputs) is written to tests/golden/*.npz.  The GPU box has
no /root/reference: tests there regenerate the inputs from the seeds and compare
against these files.
"""
import contextlib
import io
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")

from dmvsnet_amd import synth  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    from networks import module as ref_module  # noqa: E402
    from networks import mvsnet as ref_mvsnet  # noqa: E402

torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def checksum(t):
    a = npy(t).astype(np.float64)
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum()], dtype=np.float64)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KB, keys={len(arrs)}")


def build_ref(ndepths, ratios, seed, inverse=False):
    with contextlib.redirect_stdout(io.StringIO()):
        net = ref_mvsnet.MVSNet(ndepths, ratios, inverse_depth=inverse)
    sd = synth.synth_state_dict(net.state_dict(), seed)
    net.load_state_dict(sd, strict=True)
    net.eval()
    return net, sd


STAGE_KEYS = ["depth", "photometric_confidence", "photometric_confidence_refine", "depth_sub_plus",
              "depth_sub_plus_refine", "depth_values_c", "interval"]


@torch.no_grad()
def e2e(name, H, W, V, ndepths, ratios, seed, inverse=False, keep_hyps=True):
    net, _ = build_ref(ndepths, ratios, seed, inverse)
    imgs, proj, dv = synth.synth_inputs(H, W, V, seed)
    out = net(imgs, proj, dv)
    arrs = {"cfg": np.array([H, W, V, seed, int(inverse)] + list(ndepths) + list(ratios)),
            "imgs_checksum": checksum(imgs)}
    for s in range(len(ndepths)):
        st = out[f"stage{s + 1}"]
        for k in STAGE_KEYS:
            arrs[f"stage{s + 1}.{k}"] = st[k]
        if keep_hyps and s == 0:
            arrs[f"stage{s + 1}.depth_values"] = st["depth_values"]
    d = out["depth"]
    print(f"  {name}: depth mean {d.mean():.2f} std {d.std():.2f}; conf mean "
          f"{out['photometric_confidence'].mean():.3f}; softmax peak {out['prob_volume'].max(2)[0].mean():.3f}")
    save(name, **arrs)


@torch.no_grad()
def per_op(seed=7):
    g = np.random.Generator(np.random.PCG64(seed))

    def rnd(*shape, scale=1.0):
        return torch.from_numpy((g.standard_normal(shape, dtype=np.float32) * np.float32(scale)))

    # ---- homo_warping: C=8, D=3, 6x10; includes out-of-bounds and z<0 planes
    C, D, H, W = 8, 3, 6, 10
    src = rnd(1, C, H, W)
    ref_proj = torch.eye(4).unsqueeze(0)
    ref_proj[0, 0, 0] = ref_proj[0, 1, 1] = 20.0
    ref_proj[0, 0, 2], ref_proj[0, 1, 2] = 5.0, 3.0
    src_proj = ref_proj.clone()
    a = 0.05
    R = torch.tensor([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=torch.float32)
    E = torch.eye(4)
    E[:3, :3] = R
    E[:3, 3] = torch.tensor([-3.0, 0.5, 0.2])
    src_proj[0, :3, :4] = ref_proj[0, :3, :3] @ E[:3, :4]
    depth = torch.stack([torch.full((H, W), 8.0), torch.full((H, W), 40.0) + rnd(H, W), torch.full((H, W), -5.0)])[None]
    warped, grid = ref_module.homo_warping(src, src_proj, ref_proj, depth)
    save("op_homo_warping.npz", src=src, src_proj=src_proj, ref_proj=ref_proj, depth=depth, warped=warped, grid=grid)

    # ---- CostAgg with 2 source views (C=8, D=4, 8x12) using synth cameras
    C, D, H, W, V = 8, 4, 8, 12, 3
    feats = [rnd(1, C, H, W) for _ in range(V)]
    cams = synth.synth_cameras(H * 4, W * 4, V)["stage1"]
    dvals = (500.0 + 60.0 * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1) + rnd(1, D, H, W, scale=5.0))
    agg = ref_mvsnet.CostAgg("variance")
    agg.eval()
    sim = agg(feats, cams, dvals, 0)
    save("op_costagg.npz", **{f"feat{v}": feats[v] for v in range(V)}, proj=cams, depth=dvals, sim=sim)

    # ---- CostRegNet_part / _refine (D=8 / 4, 16x16)
    net, sd = build_ref([8], [4], seed)
    x = rnd(1, 2, 8, 16, 16, scale=0.3)
    y = net.cost_regularization[0].cosR_small(x)
    yfull = net.cost_regularization[0](x)
    xr = rnd(1, 2, 4, 16, 16, scale=0.3)
    yr = net.cost_regularization_refine[0].cosR_huge(xr)
    yrfull = net.cost_regularization_refine[0](xr)
    save("op_costreg.npz", seed=np.array(seed), x=x, y_small=y, y_full=yfull, xr=xr, yr_huge=yr, yr_full=yrfull)

    # ---- DepthNet.forward / .refine: 8x6 grid covers all (row%4, col%2) cases
    Dn, H, W = 8, 8, 6
    logits = rnd(1, 4, Dn, H, W, scale=2.0)
    dv = 500.0 + 10.0 * torch.arange(Dn, dtype=torch.float32).view(1, Dn, 1, 1) + rnd(1, Dn, H, W)
    itv = torch.tensor(10.0)
    dn = ref_mvsnet.DepthNet()
    o = dn(logits, dv, Dn, itv)
    logits_c = rnd(1, 4, 4, H, W, scale=1.0)
    o2 = dn.refine(logits_c, o["depth_values_c"], 4, itv)
    save("op_depthnet.npz", logits=logits, depth_values=dv, interval=itv, conf=o["photometric_confidence"],
         prob=o["prob_volume"], dsp=o["depth_sub_plus"], hyps=o["depth_values_c"], logits_c=logits_c,
         depth=o2["depth"], conf_refine=o2["photometric_confidence_refine"], dsp_refine=o2["depth_sub_plus_refine"])

    # ---- get_depth_range_samples: linear/inverse x first-stage/later-stage
    dvv = synth.synth_depth_values()
    pix = (dvv[0, -1] - dvv[0, 0]) / dvv.size(1) * 2
    last = 600.0 + rnd(1, 6, 8, scale=30.0)
    arrs = {"last": last, "pix": pix}
    for inv in (False, True):
        s, i = ref_module.get_depth_range_samples(dvv, 8, pix, shape=[6, 8], inverse=inv)
        arrs[f"first_inv{int(inv)}"] = s
        arrs[f"first_inv{int(inv)}_itv"] = i
        s, i = ref_module.get_depth_range_samples(last, 8, pix, shape=[12, 16], inverse=inv)
        arrs[f"later_inv{int(inv)}"] = s
        arrs[f"later_inv{int(inv)}_itv"] = i
        arrs[f"later_inv{int(inv)}_up"] = torch.nn.functional.interpolate(s, [12, 16], mode="bilinear", align_corners=False)
    save("op_hypotheses.npz", **arrs)

    # ---- FeatureNet on a 32x32 image
    img = torch.from_numpy(g.random((1, 3, 32, 32), dtype=np.float32))
    f = net.feature(img)
    save("op_featurenet.npz", seed=np.array(seed), img=img, **{k: v for k, v in f.items()})


def state_dict_keys():
    """Key -> shape of the reference's 3-stage network: the checkpoint contract of the boundary."""
    import json
    net, _ = build_ref([48, 32, 8], [4, 2, 1], 0)
    keys = {k: list(v.shape) for k, v in net.state_dict().items()}
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    print("state_dict_keys.json:", len(keys), "tensors,", sum(int(np.prod(s)) if s else 1 for s in keys.values()), "elements")


if __name__ == "__main__":
    state_dict_keys()
    if "--keys-only" in sys.argv:
        sys.exit(0)
    per_op()
    e2e("e2e_c1.npz", seed=0, **synth.CONFIGS["c1"])
    e2e("e2e_small3.npz", 64, 96, 3, [16, 8, 8], [3, 2, 1], seed=1)
    e2e("e2e_small3_inv.npz", 64, 96, 3, [16, 8, 8], [4, 2, 1], seed=2, inverse=True)
