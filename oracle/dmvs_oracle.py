"""ORACLE -- test infrastructure only.  NOT part of the shipped product path.

CPU restatement of DMVSNet's per-stage hot path (plane-sweep warp + 2-group
correlation + dual 3D U-Net regularisation + dual-depth regression) written
from the behaviour of /root/reference/networks/{mvsnet,module}.py.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this file; ``dmvsnet_amd`` never does (tests/test_boundary.py
enforces that).

Pinning status: the reference ships NO tests, golden vectors or fixtures
(SURVEY.md section 4), so parity is pinned by golden vectors generated in the
build container by importing the reference itself
(tests/golden/make_golden.py -> tests/golden/*.npz); tests/test_oracle.py checks
every function below against them.

Third-party arithmetic: every hot op of the reference executes inside PyTorch
ATen (authors pin 1.8.1, README.md:68; this image has 2.10.0): grid_sample,
conv3d, conv_transpose3d, batch_norm, softmax, inverse, interpolate.  This file
calls the same ATen CPU ops; ``oracle/ref_ops.c`` restates their published
algorithms in plain C (no ATen) and is cross-checked against this file in
tests/test_oracle.py.

The model is expressed functionally over a plain ``state_dict`` whose keys are the
reference's (SURVEY.md section 8b "Weights").
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm default, module.py:50,144


# --------------------------------------------------------------------------- helpers
def _bn(sd, p, x):
    """Eval-mode BatchNorm (module.py:59-60,153-154): (x-mean)/sqrt(var+eps)*w+b."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def _cbr2(sd, p, x, stride, pad):
    """Conv2d + BN + ReLU wrapper, module.py:57-63."""
    return F.relu(_bn(sd, p + ".bn", F.conv2d(x, sd[p + ".conv.weight"], None, stride, pad)))


def _cbr3(sd, p, x, stride):
    """Conv3d(k3,p1) + BN + ReLU wrapper, module.py:151-157."""
    return F.relu(_bn(sd, p + ".bn", F.conv3d(x, sd[p + ".conv.weight"], None, stride, 1)))


def _dbr3(sd, p, x):
    """ConvTranspose3d(k3,s2,p1,op1) + BN + ReLU, module.py:196-202,372-376."""
    y = F.conv_transpose3d(x, sd[p + ".conv.weight"], None, 2, 1, 1)
    return F.relu(_bn(sd, p + ".bn", y))


def _dbr2(sd, p, x):
    """ConvTranspose2d(k3,s2,p1,op1) + crop + BN + ReLU, module.py:102-111,414."""
    y = F.conv_transpose2d(x, sd[p + ".conv.weight"], None, 2, 1, 1)
    h, w = x.shape[2:]
    y = y[:, :, : 2 * h, : 2 * w]
    return F.relu(_bn(sd, p + ".bn", y))


def _parity_mask(h, w, device):
    """True where row%2 == col%2 (module.py:573-577; 'ij' meshgrid: first axis = rows)."""
    r = torch.arange(h, device=device).view(h, 1)
    c = torch.arange(w, device=device).view(1, w)
    return (r % 2) == (c % 2)


# --------------------------------------------------------------------------- FeatureNet
def feature_net(sd: Dict[str, torch.Tensor], img: torch.Tensor, p: str = "feature") -> Dict[str, torch.Tensor]:
    """2D FPN, module.py:274-340.  Returns stageK / stageK_c halves (split at 326,331,336)."""
    c0 = _cbr2(sd, f"{p}.conv0.0", img, 1, 1)
    c0 = _cbr2(sd, f"{p}.conv0.1", c0, 1, 1)
    c1 = _cbr2(sd, f"{p}.conv1.0", c0, 2, 2)
    c1 = _cbr2(sd, f"{p}.conv1.1", c1, 1, 1)
    c1 = _cbr2(sd, f"{p}.conv1.2", c1, 1, 1)
    c2 = _cbr2(sd, f"{p}.conv2.0", c1, 2, 2)
    c2 = _cbr2(sd, f"{p}.conv2.1", c2, 1, 1)
    c2 = _cbr2(sd, f"{p}.conv2.2", c2, 1, 1)
    out = {}
    intra = c2
    o = F.conv2d(intra, sd[f"{p}.out1.weight"])
    out["stage1"], out["stage1_c"] = o.split(o.shape[1] // 2, 1)
    intra = F.interpolate(intra, scale_factor=2, mode="nearest") + F.conv2d(c1, sd[f"{p}.inner1.weight"], sd[f"{p}.inner1.bias"])
    o = F.conv2d(intra, sd[f"{p}.out2.weight"], None, 1, 1)
    out["stage2"], out["stage2_c"] = o.split(o.shape[1] // 2, 1)
    intra = F.interpolate(intra, scale_factor=2, mode="nearest") + F.conv2d(c0, sd[f"{p}.inner2.weight"], sd[f"{p}.inner2.bias"])
    o = F.conv2d(intra, sd[f"{p}.out3.weight"], None, 1, 1)
    out["stage3"], out["stage3_c"] = o.split(o.shape[1] // 2, 1)
    return out


# --------------------------------------------------------------------------- a1 hypotheses
def depth_hypotheses(last_depth, ndepth: int, pix_interval, shape, inverse: bool = False):
    """get_depth_range_samples, module.py:556-649 (+ helpers 476-507, 525-554).

    last_depth: [B,n] (first stage; only [:,0] and [:,-1] are read) or [B,h,w].
    Returns (samples [B,D,h,w] at *last_depth's* resolution for stage>1, interval 0-dim).
    The x2 bilinear upsample of mvsnet.py:232-233 is applied by the caller.
    """
    D = ndepth
    if last_depth.dim() == 2:
        dmin, dmax = last_depth[:, 0], last_depth[:, -1]
        step = (dmax - dmin) / (D - 1)
        itv = step[0]
        H, W = shape
        mask = _parity_mask(H, W, last_depth.device)
        ar = torch.arange(D, device=last_depth.device, dtype=last_depth.dtype)
        if not inverse:  # module.py:560-579
            planes = dmin[:, None] + ar[None] * step[:, None]
            vol = planes[:, :, None, None].repeat(1, 1, H, W)
            return torch.where(mask, vol - itv, vol + itv), itv
        # inverse, first stage: module.py:598-634
        lo, hi = dmin - itv, dmax - itv
        itv = ((hi - lo) / (D - 1))[0]
        vn = torch.stack([torch.linspace(1 / a, 1 / b, D, device=last_depth.device) for a, b in zip(lo, hi)])
        lo, hi = dmin + itv, dmax + itv
        itv = ((hi - lo) / (D - 1))[0]
        vp = torch.stack([torch.linspace(1 / a, 1 / b, D, device=last_depth.device) for a, b in zip(lo, hi)])
        vn = (1 / vn)[:, :, None, None].repeat(1, 1, H, W)
        vp = (1 / vp)[:, :, None, None].repeat(1, 1, H, W)
        return torch.where(mask, vn, vp).float(), itv.float()

    # later stages: per-pixel ranges around last_depth, module.py:582-594 / 636-648
    h, w = last_depth.shape[-2:]
    mask = _parity_mask(h, w, last_depth.device)
    ar = torch.arange(D, device=last_depth.device, dtype=last_depth.dtype).view(1, D, 1, 1)

    def span(below, above):
        lo = last_depth - below / 2 * pix_interval
        hi = last_depth + above / 2 * pix_interval
        if not inverse:  # module.py:476-507
            return lo.unsqueeze(1) + ar * ((hi - lo) / (D - 1)).unsqueeze(1)
        ilo, ihi = 1 / lo, 1 / hi  # module.py:525-554
        return 1 / (ilo.unsqueeze(1) + ar * ((ihi - ilo) / (D - 1)).unsqueeze(1))

    vn = span(D + 2, D - 2)
    vp = span(D - 2, D + 2)
    itv = (D * pix_interval) / (D - 1)
    out = torch.where(mask, vn, vp)
    if inverse:
        return out.float(), itv.float()
    return out, itv


# --------------------------------------------------------------------------- a2+a3 warp + correlation
def compose_projection(proj_pair: torch.Tensor) -> torch.Tensor:
    """K[:3,:3] @ E[:3,:4] written into a copy of E (mvsnet.py:133-136). proj_pair: [B,2,4,4]."""
    P = proj_pair[:, 0].clone()
    P[:, :3, :4] = torch.matmul(proj_pair[:, 1, :3, :3], proj_pair[:, 0, :3, :4])
    return P


def relative_projection(src_pair: torch.Tensor, ref_pair: torch.Tensor):
    """rot [B,3,3], trans [B,3] of src_proj @ inverse(ref_proj) (module.py:223-225)."""
    proj = torch.matmul(compose_projection(src_pair), torch.inverse(compose_projection(ref_pair)))
    return proj[:, :3, :3], proj[:, :3, 3]


def warp_source(src_fea, rot, trans, depth_values):
    """homo_warping, module.py:212-251: [B,C,H,W] -> [B,C,D,H,W]."""
    B, C, H, W = src_fea.shape
    D = depth_values.shape[1]
    dev = src_fea.device
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=dev),
                            torch.arange(W, dtype=torch.float32, device=dev), indexing="ij")
    xyz = torch.stack((xx.reshape(-1), yy.reshape(-1), torch.ones(H * W, device=dev)))  # [3,HW]
    rot_xyz = torch.matmul(rot, xyz.unsqueeze(0).expand(B, 3, H * W))  # [B,3,HW]
    pts = rot_xyz.unsqueeze(2) * depth_values.reshape(B, 1, D, H * W) + trans.view(B, 3, 1, 1)
    z = pts[:, 2]
    z = torch.where(z == 0, z + 1e-5, z)  # module.py:237
    gx = pts[:, 0] / z / ((W - 1) / 2) - 1
    gy = pts[:, 1] / z / ((H - 1) / 2) - 1
    grid = torch.stack((gx, gy), dim=3).view(B, D * H, W, 2)
    out = F.grid_sample(src_fea, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    return out.view(B, C, D, H, W)


def warp_corr(features: Sequence[torch.Tensor], proj_matrices: torch.Tensor, depth_values: torch.Tensor,
              views: Sequence[int] | None = None):
    """CostAgg.forward (variance mode), mvsnet.py:111-153.

    features: V tensors [B,C,H,W]; proj_matrices [B,V,2,4,4]; depth_values [B,D,H,W].
    ``views`` optionally restricts the source views that are summed (1-based indices into
    ``features``) -- the multi-GPU view shard; the reference always sums all of them.
    Output [B,2,D,H,W]: group k = mean over g of warped[2g+k]*ref[2g+k] (mvsnet.py:139),
    summed (not averaged) over source views (mvsnet.py:146).
    """
    ref = features[0]
    B, C, H, W = ref.shape
    D = depth_values.shape[1]
    total = torch.zeros(B, 2, D, H, W, dtype=ref.dtype, device=ref.device)
    idx = range(1, len(features)) if views is None else views
    for v in idx:
        rot, trans = relative_projection(proj_matrices[:, v], proj_matrices[:, 0])
        warped = warp_source(features[v], rot, trans, depth_values)
        prod = warped.view(B, C // 2, 2, D, H, W) * ref.view(B, C // 2, 2, 1, H, W)
        total += prod.mean(1)
    return total


# --------------------------------------------------------------------------- a4/a5 regularisation
def cost_reg_branch(sd, p: str, x: torch.Tensor) -> torch.Tensor:
    """CostRegNet_part.forward, module.py:389-398."""
    c0 = _cbr3(sd, p + ".conv0", x, 1)
    c2 = _cbr3(sd, p + ".conv2", _cbr3(sd, p + ".conv1", c0, 2), 1)
    c4 = _cbr3(sd, p + ".conv4", _cbr3(sd, p + ".conv3", c2, 2), 1)
    y = _cbr3(sd, p + ".conv6", _cbr3(sd, p + ".conv5", c4, 2), 1)
    y = c4 + _dbr3(sd, p + ".conv7", y)
    y = c2 + _dbr3(sd, p + ".conv9", y)
    y = c0 + _dbr3(sd, p + ".conv11", y)
    return F.conv3d(y, sd[p + ".prob.weight"], None, 1, 1)


def cost_reg_branch_refine(sd, p: str, x: torch.Tensor) -> torch.Tensor:
    """CostRegNet_part_refine.forward, module.py:426-436 (D=4 -> 2 -> 1, 2D bottleneck)."""
    c0 = _cbr3(sd, p + ".conv0", x, 1)
    c2 = _cbr3(sd, p + ".conv2", _cbr3(sd, p + ".conv1", c0, 2), 1)
    c4 = _cbr3(sd, p + ".conv4", _cbr3(sd, p + ".conv3", c2, 2), 1).squeeze(2)
    y = _cbr2(sd, p + ".conv6", _cbr2(sd, p + ".conv5", c4, 2, 1), 1, 1)
    y = c4 + _dbr2(sd, p + ".conv7", y)
    y = y.unsqueeze(2)
    y = c2 + _dbr3(sd, p + ".conv9", y)
    y = c0 + _dbr3(sd, p + ".conv11", y)
    return F.conv3d(y, sd[p + ".prob.weight"], None, 1, 1)


def cost_reg(sd, p: str, x: torch.Tensor, refine: bool = False) -> torch.Tensor:
    """CostRegNet / CostRegNet_refine: cat(small, huge) on channels, module.py:342-357."""
    f = cost_reg_branch_refine if refine else cost_reg_branch
    return torch.cat((f(sd, p + ".cosR_small", x), f(sd, p + ".cosR_huge", x)), dim=1)


# --------------------------------------------------------------------------- a6/a7 regression
def _expectation(logits, depth_values, alpha: float = 1.0):
    prob = F.softmax(logits * alpha, dim=2) if alpha != 1.0 else F.softmax(logits, dim=2)
    return prob, torch.sum(prob * depth_values.unsqueeze(1), dim=2)  # module.py:454-460


def _confidence(dsp, interval):
    """2*(sigmoid(interval/(std_pop+1e-5))-0.5), mvsnet.py:61-62,96-97."""
    return 2 * (torch.sigmoid(interval / (dsp.var(1, unbiased=False).sqrt() + 1e-5)) - 0.5)


def _six(m, M):
    return torch.stack((3 * m - 2 * M, 2 * m - M, m, M, 2 * M - m, 3 * M - 2 * m), 1)  # mvsnet.py:42-45


def depth_regress_main(logits, depth_values, interval):
    """DepthNet.forward, mvsnet.py:15-66."""
    prob, dsp = _expectation(logits, depth_values)
    small, huge = dsp[:, :2], dsp[:, 2:]
    sm, sM = small.min(1)[0], small.max(1)[0]
    hm, hM = huge.min(1)[0], huge.max(1)[0]
    stacks = [_six(sm, sM), _six(hm, hM),
              _six(2 * sm - sM, 2 * sM - sm), _six(2 * hm - hM, 2 * hM - hm)]  # rows%4 = 0,1,2,3
    B, _, H, W = dsp.shape
    r = torch.arange(H, device=dsp.device).view(1, 1, H, 1)
    c = torch.arange(W, device=dsp.device).view(1, 1, 1, W)
    hyps = torch.zeros_like(dsp)
    for q in range(4):
        lo, hi = stacks[q][:, :4], stacks[q][:, 2:]
        # (row%4,col%2): (0,0)->[0:4] (0,1)->[2:6] (1,0)->[2:6] (1,1)->[0:4] ... mvsnet.py:49-56
        pick_hi = ((r + c) % 2) == 1
        hyps = torch.where((r % 4) == q, torch.where(pick_hi, hi, lo), hyps)
    return {"photometric_confidence": _confidence(dsp, interval), "prob_volume": prob,
            "depth_sub_plus": dsp, "depth_values_c": hyps, "depth_values": depth_values, "interval": interval}


def depth_regress_refine(logits, depth_values, interval, alpha: float = 5.0):
    """DepthNet.refine, mvsnet.py:67-100."""
    _, dsp = _expectation(logits, depth_values, alpha)
    small, huge = dsp[:, :2], dsp[:, 2:]
    sm, sM = small.min(1)[0], small.max(1)[0]
    hm, hM = huge.min(1)[0], huge.max(1)[0]
    B, _, H, W = dsp.shape
    r = torch.arange(H, device=dsp.device).view(1, H, 1) % 2
    c = torch.arange(W, device=dsp.device).view(1, 1, W) % 2
    depth = torch.where(r == 0, torch.where(c == 0, sm, sM), torch.where(c == 0, hM, hm))  # mvsnet.py:88-91
    return {"depth": depth, "photometric_confidence_refine": _confidence(dsp, interval),
            "depth_sub_plus_refine": dsp}


# --------------------------------------------------------------------------- a8 stage loop
def stage_level(stage_idx: int, num_stage: int) -> int:
    """Feature-pyramid level (0 = 1/4 resolution, 2 = full) of a stage.  The reference is hard-wired to <= 3 stages,
    where the level IS the stage index (mvsnet.py:214: scale = 2 ** (3 - stage_idx - 1)); the declared EXTENSION for
    deeper pyramids (BASELINE configs[4]: 4 stages -- the reference raises KeyError 'stage4' there) keeps
    the three FPN levels and spends the extra stages at the coarsest one: levels 0, 0, 1, 2 for 4 stages."""
    return stage_idx if num_stage <= 3 else max(0, stage_idx - (num_stage - 3))


def stage_pass(sd, stage_idx: int, feats: List[Dict[str, torch.Tensor]], proj_stage, hyps_volume, interval,
               views=None, reduce_fn=None, level=None):
    """One coarse-to-fine stage: main pass then refine pass (mvsnet.py:236-254)."""
    k = (stage_idx if level is None else level) + 1
    sim = warp_corr([f[f"stage{k}"] for f in feats], proj_stage, hyps_volume, views)
    if reduce_fn is not None:
        sim = reduce_fn(sim)
    logits = cost_reg(sd, f"cost_regularization.{stage_idx}", sim, refine=False)
    main = depth_regress_main(logits, hyps_volume, interval)
    sim_c = warp_corr([f[f"stage{k}_c"] for f in feats], proj_stage, main["depth_values_c"], views)
    if reduce_fn is not None:
        sim_c = reduce_fn(sim_c)
    logits_c = cost_reg(sd, f"cost_regularization_refine.{stage_idx}", sim_c, refine=True)
    ref = depth_regress_refine(logits_c, main["depth_values_c"], interval)
    return {**ref, **main}, {"sim": sim, "logits": logits, "sim_c": sim_c, "logits_c": logits_c}


@torch.no_grad()
def mvsnet_forward(sd, ndepths, ratios, imgs, proj_matrices, depth_values, inverse_depth=False,
                   views=None, reduce_fn=None, keep_intermediates=False):
    """MVSNet.forward, mvsnet.py:188-260.  Returns the reference's output dict."""
    depth_interval = (depth_values[0, -1] - depth_values[0, 0]) / depth_values.size(1)  # mvsnet.py:196
    feats = [feature_net(sd, imgs[:, v]) for v in range(imgs.size(1))]
    H, W = imgs.shape[-2:]
    outputs, inter = {}, {}
    last = None
    S = len(ndepths)
    for s in range(S):
        level = stage_level(s, S)
        scale = 2 ** (3 - level - 1)
        shape = (H // scale, W // scale)
        if s == 0:
            hyp, itv = depth_hypotheses(depth_values, ndepths[s], ratios[s] * depth_interval, shape, inverse_depth)
        else:
            hyp, itv = depth_hypotheses(last, ndepths[s], ratios[s] * depth_interval, shape, inverse_depth)
            hyp = F.interpolate(hyp, shape, mode="bilinear", align_corners=False)  # mvsnet.py:233 (identity at equal size)
        # (extension, S > 3: a loader that emits the reference's three projection scales serves the stages by level)
        per_stage = S <= 3 or f"stage{S}" in proj_matrices
        proj = proj_matrices[f"stage{s + 1}" if per_stage else f"stage{level + 1}"]
        out, mid = stage_pass(sd, s, feats, proj, hyp, itv, views, reduce_fn, level)
        last = out["depth"]
        outputs[f"stage{s + 1}"] = out
        outputs.update(out)
        if keep_intermediates:
            inter[f"stage{s + 1}"] = mid
    if keep_intermediates:
        outputs["_intermediates"] = inter
        outputs["_features"] = feats
    return outputs
