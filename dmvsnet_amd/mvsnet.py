"""Host-side mirror of the reference's network interface for the cost-volume hot path.

``MVSNet`` keeps the constructor arguments, ``forward(imgs, proj_matrices, depth_values) -> dict``
signature, output keys and the state_dict key layout of /root/reference/networks/mvsnet.py:156-260, so
a DMVSNet checkpoint loads unchanged (model.py:59-70) and the module drops into ``Model.test()``
(model.py:336).  Below that boundary nothing of the reference's implementation is reused:

  stage loop (mvsnet.py:208-258)  ->  ops.hypotheses_*  (hypothesis planes, one kernel)
                                      ops.warp_corr     (K1: warp + group correlation + view sum)
                                      ops.conv3d        (K2 direct / K3 MFMA, BN+ReLU+skip fused)
                                      ops.depth_regress (K4: softmax, expectation, selection, confidence)

The parameter-holder sub-modules (``feature``, ``cost_regularization``...) exist to own the weights under
the reference's names; the 3D ones are never called -- their tensors are folded (BatchNorm -> scale/shift)
and re-packed once per device into kernel layouts.  FeatureNet (module.py:274-340) runs on the same MFMA conv
kernels with kdepth = 1 (the V views are the depth slices of a [C][V][H][W] stack) and writes its outputs
quad-planar ([C/4][H][W][4]), the layout the warp kernel samples.

Inference only: the module refuses ``train()`` mode and CPU tensors (there is no CPU fallback; the CPU
restatement of this path lives in oracle/ and is test infrastructure).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import ops
from ._lib import DmvsError

BN_EPS = 1e-5


# ----------------------------------------------------------------------------------- parameter holders
class _ConvBn(nn.Module):
    """Holder for ``<name>.conv.weight`` + ``<name>.bn.*`` (module.py:28-208 wrappers)."""

    def __init__(self, cin, cout, k, dims, transposed=False):
        super().__init__()
        if dims == 3:
            self.conv = (nn.ConvTranspose3d if transposed else nn.Conv3d)(cin, cout, k, bias=False)
            self.bn = nn.BatchNorm3d(cout)
        else:
            self.conv = (nn.ConvTranspose2d if transposed else nn.Conv2d)(cin, cout, k, bias=False)
            self.bn = nn.BatchNorm2d(cout)

    def folded(self):
        """(scale, shift) of eval-mode BatchNorm: y = x*scale + shift."""
        scale = self.bn.weight / torch.sqrt(self.bn.running_var + BN_EPS)
        return scale, self.bn.bias - self.bn.running_mean * scale


class FeatureNet(nn.Module):
    """2D FPN (module.py:274-340); same attribute tree => same state_dict keys."""

    def __init__(self, base_channels=8):
        super().__init__()
        b = base_channels
        self.conv0 = nn.Sequential(_ConvBn(3, b, 3, 2), _ConvBn(b, b, 3, 2))
        self.conv1 = nn.Sequential(_ConvBn(b, 2 * b, 5, 2), _ConvBn(2 * b, 2 * b, 3, 2), _ConvBn(2 * b, 2 * b, 3, 2))
        self.conv2 = nn.Sequential(_ConvBn(2 * b, 4 * b, 5, 2), _ConvBn(4 * b, 4 * b, 3, 2), _ConvBn(4 * b, 4 * b, 3, 2))
        self.out1 = nn.Conv2d(4 * b, 8 * b, 1, bias=False)
        self.inner1 = nn.Conv2d(2 * b, 4 * b, 1, bias=True)
        self.inner2 = nn.Conv2d(b, 4 * b, 1, bias=True)
        self.out2 = nn.Conv2d(4 * b, 4 * b, 3, padding=1, bias=False)
        self.out3 = nn.Conv2d(4 * b, 2 * b, 3, padding=1, bias=False)
        self.out_channels = [4 * b, 2 * b, b]
        self._packed = None
        self.fuse_topdown = True   # level-3 lateral conv + upsample-add fused into out3's input staging

    # -- K3 path ------------------------------------------------------------------------------------
    def pack(self):
        """Fold BatchNorm and pack every layer for the MFMA conv kernel (kdepth = 1: views are depth slices)."""
        def layer(name, w, mode, scale, shift, relu):
            cout, cin = w.shape[0], w.shape[1]
            wm = ops.pack_mfma(w, cin, cout, mode, 1)
            if wm is None:
                raise DmvsError(f"FeatureNet layer {name} ({cin}->{cout}) is not covered by the MFMA kernel")
            ww = ops.pack_wino(w, cin, cout, 1) if (mode == ops.CONV_S1 and w.shape[-1] == 3) else None
            # conv2.1 / conv2.2 are the 32 -> 32 2D shape K3r compiles (ops.use_coarse_feature)
            wr = ops.pack_coarse(w, cin, cout, 1) if (ops.use_coarse_feature and mode == ops.CONV_S1 and w.shape[-1] == 3) else None
            return ops.ConvLayer("feature." + name, mode, 1, cin, cout, None, wm.to(w.device),
                                 None if scale is None else scale.detach().contiguous(),
                                 None if shift is None else shift.detach().contiguous(), relu,
                                 None if ww is None else ww.to(w.device), w_coarse=None if wr is None else wr.to(w.device))

        L = {}
        spec = (("conv0.0", self.conv0[0], ops.CONV_S1), ("conv0.1", self.conv0[1], ops.CONV_S1),
                ("conv1.0", self.conv1[0], ops.CONV2D_K5S2), ("conv1.1", self.conv1[1], ops.CONV_S1),
                ("conv1.2", self.conv1[2], ops.CONV_S1), ("conv2.0", self.conv2[0], ops.CONV2D_K5S2),
                ("conv2.1", self.conv2[1], ops.CONV_S1), ("conv2.2", self.conv2[2], ops.CONV_S1))
        for name, m, mode in spec:
            w = m.conv.weight.detach()
            if w.shape[1] == 3:  # RGB + one zero channel: the MFMA k-group is 4 channels wide
                w = torch.cat((w, torch.zeros_like(w[:, :1])), 1).contiguous()
            scale, shift = m.folded()
            L[name] = layer(name, w, mode, scale, shift, True)
            if name.startswith("conv0."):   # the two 8-channel full-resolution layers: K3s (row sweep on the 4x4x1 MFMA)
                wc = ops.pack_c8(m.conv.weight.detach())
                L[name].w_c8 = None if wc is None else wc.to(w.device)
        one = lambda n: torch.ones(n, device=self.out1.weight.device)
        L["out1"] = layer("out1", self.out1.weight.detach(), ops.CONV2D_K1, None, None, False)
        L["inner1"] = layer("inner1", self.inner1.weight.detach(), ops.CONV2D_K1, one(self.inner1.out_channels),
                            self.inner1.bias, False)
        L["inner2"] = layer("inner2", self.inner2.weight.detach(), ops.CONV2D_K1, one(self.inner2.out_channels),
                            self.inner2.bias, False)
        self._inner2_w = self.inner2.weight.detach().reshape(self.inner2.out_channels, -1).contiguous()   # [32, 8]
        self._inner2_b = self.inner2.bias.detach().contiguous()
        L["out2"] = layer("out2", self.out2.weight.detach(), ops.CONV_S1, None, None, False)
        L["out3"] = layer("out3", self.out3.weight.detach(), ops.CONV_S1, None, None, False)
        wf = ops.pack_wino_fpn(self.out3.weight.detach(), self._inner2_w, self._inner2_b)   # inner2 folded into out3
        L["out3"].w_wino_fpn = None if wf is None else wf.to(self.out3.weight.device)
        self._packed = L

    def run(self, imgs_v, side=None):
        """imgs_v [V,3,H,W] -> three outputs [2,V,C/4,h,w,4]: the stageK / stageK_c channel halves (module.py:326-336)
        of every view, QUAD-PLANAR -- the layout the warp kernel samples -- written directly by the output
        layers' epilogue.  conv+BN+ReLU are single kernels; the FPN's nearest x2 upsample + add
        (module.py:328,333) is the 1x1 lateral conv's epilogue."""
        V, _, H, W = imgs_v.shape
        L = self._packed
        f = lambda t, n, **kw: ops.conv3d(t, L[n], family="feature_mfma", **kw)
        # conv0.0 reads the loader's [V,3,H,W] images in place (DMVS_IN_VIEWS; r01-r03 first copied them into a planar
        # [4,V,H,W] stack with a zero channel: two torch kernels, 55 us per depth map)
        c0 = ops.featurenet_conv0(imgs_v, L["conv0.0"], L["conv0.1"], family="feature_mfma")   # one sweep, no intermediate
        if c0 is None:
            c0 = f(f(imgs_v, "conv0.0", in_views=True), "conv0.1")
        c1 = f(f(f(c0, "conv1.0"), "conv1.1"), "conv1.2")
        c2 = f(f(f(c1, "conv2.0"), "conv2.1"), "conv2.2")
        o1 = f(c2, "out1", out_q4=True)

        def topdown():
            intra = f(c1, "inner1", skip=c2, skip_up2=True)
            o2 = f(intra, "out2", out_q4=True)
            # level 3: inner2 + upsample-add + out3 in ONE kernel (the 32-channel full-resolution tensor is never stored)
            o3 = None
            if self.fuse_topdown:
                o3 = ops.conv3d_fpn(c0, intra, L["out3"], out_q4=True, family="feature_mfma")
            if o3 is None:
                intra = f(c0, "inner2", skip=intra, skip_up2=True)
                o3 = f(intra, "out3", out_q4=True)
            return o2, o3

        if side is None:
            o2, o3 = topdown()
            return o1, o2, o3
        # The level-2 / level-3 outputs are first needed by stage 2: run the top-down path on a side stream under the
        # stage-1 kernels; `done` is waited for before stage 2 (MVSNet.forward)
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            o2, o3 = topdown()
            done = torch.cuda.Event()
            done.record(side)
        for t in (c0, c1, c2):       # allocated on main, read on side
            t.record_stream(side)
        for t in (o2, o3):           # allocated on side, read on main (after `done`)
            t.record_stream(main)
        self._topdown_done = done
        return o1, o2, o3

    def forward(self, x):
        raise DmvsError("FeatureNet is a parameter holder; MVSNet.forward runs it through FeatureNet.run (K3 kernels)")


class _RegBranch(nn.Module):
    """CostRegNet_part / CostRegNet_part_refine parameter tree (module.py:358-436)."""

    def __init__(self, cin, b, refine):
        super().__init__()
        self.refine = refine
        self.conv0 = _ConvBn(cin, b, 3, 3)
        self.conv1 = _ConvBn(b, 2 * b, 3, 3)
        self.conv2 = _ConvBn(2 * b, 2 * b, 3, 3)
        self.conv3 = _ConvBn(2 * b, 4 * b, 3, 3)
        self.conv4 = _ConvBn(4 * b, 4 * b, 3, 3)
        d = 2 if refine else 3  # refine: D has collapsed to 1 -> 2D bottleneck (module.py:411-414)
        self.conv5 = _ConvBn(4 * b, 8 * b, 3, d)
        self.conv6 = _ConvBn(8 * b, 8 * b, 3, d)
        self.conv7 = _ConvBn(8 * b, 4 * b, 3, d, transposed=True)
        self.conv9 = _ConvBn(4 * b, 2 * b, 3, 3, transposed=True)
        self.conv11 = _ConvBn(2 * b, b, 3, 3, transposed=True)
        self.prob = nn.Conv3d(b, 2, 3, stride=1, padding=1, bias=False)

    _SPEC = (("conv1", ops.CONV_S2), ("conv2", ops.CONV_S1), ("conv3", ops.CONV_S2), ("conv4", ops.CONV_S1),
             ("conv5", ops.CONV_S2), ("conv6", ops.CONV_S1), ("conv7", ops.DECONV_S2), ("conv9", ops.DECONV_S2),
             ("conv11", ops.DECONV_S2))

    def pack(self, tag) -> Dict[str, ops.ConvLayer]:
        layers = {}
        for name, mode in self._SPEC:
            m: _ConvBn = getattr(self, name)
            w = m.conv.weight.detach()
            kd = 1 if w.dim() == 4 else 3
            tr = mode == ops.DECONV_S2
            cin, cout = (w.shape[0], w.shape[1]) if tr else (w.shape[1], w.shape[0])
            scale, shift = m.folded()
            wm = ops.pack_mfma(w, cin, cout, mode, kd)
            ww = ops.pack_wino(w, cin, cout, kd) if mode == ops.CONV_S1 else None
            wr = ops.pack_coarse(w, cin, cout, kd) if mode == ops.CONV_S1 else None
            wz = ops.pack_zmarch(w, cin, cout, kd) if mode == ops.CONV_S1 else None
            layers[name] = ops.ConvLayer(f"{tag}.{name}", mode, kd, cin, cout, ops.pack_direct(w, tr),
                                         None if wm is None else wm.to(w.device), scale.detach().contiguous(),
                                         shift.detach().contiguous(), True, None if ww is None else ww.to(w.device),
                                         w_coarse=None if wr is None else wr.to(w.device),
                                         w_zmarch=None if wz is None else wz.to(w.device))
            if name == "conv1":   # the bf16-split PROBE's operands (ops.split_probe; off in the product)
                ws = ops.pack_split(w)
                layers[name].w_split = None if ws is None else ws.to(w.device)
            if kd == 3 and mode == ops.CONV_S1:
                # On a volume of depth 1 the outer depth taps only ever meet zero padding: the middle 3x3 slice as a
                # per-slice 2D conv gives the same sums with a third of the MFMA work (refine conv4, stage-3 conv6)
                w2 = w[:, :, 1].contiguous()
                wm2 = ops.pack_mfma(w2, cin, cout, mode, 1)
                if wm2 is not None:
                    ww2 = ops.pack_wino(w2, cin, cout, 1)
                    wr2 = ops.pack_coarse(w2, cin, cout, 1)
                    layers[name + "@d1"] = ops.ConvLayer(f"{tag}.{name}@d1", mode, 1, cin, cout, None, wm2.to(w.device),
                                                         scale.detach().contiguous(), shift.detach().contiguous(), True,
                                                         None if ww2 is None else ww2.to(w.device),
                                                         w_coarse=None if wr2 is None else wr2.to(w.device))
        w = self.prob.weight.detach()
        layers["prob"] = ops.ConvLayer(f"{tag}.prob", ops.CONV_S1, 3, w.shape[1], 2, ops.pack_direct(w, False), None,
                                       None, None, False)
        return layers


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class CostRegNet(nn.Module):
    """Two independent U-Nets on the same input (module.py:342-357)."""

    def __init__(self, in_channels, base_channels, refine=False):
        super().__init__()
        self.cosR_small = _RegBranch(in_channels, base_channels, refine)
        self.cosR_huge = _RegBranch(in_channels, base_channels, refine)
        self.refine = refine
        self._packed = None

    def pack(self, tag):
        s, h = self.cosR_small, self.cosR_huge
        # conv0 of both branches shares its input: one 2 -> 2b conv (channels [0:b] small, [b:2b] huge)
        w = torch.cat((s.conv0.conv.weight.detach(), h.conv0.conv.weight.detach()), 0)
        sc_s, sh_s = s.conv0.folded()
        sc_h, sh_h = h.conv0.folded()
        wm = ops.pack_mfma(w, w.shape[1], w.shape[0], ops.CONV_S1, 3)
        ww = ops.pack_wino(w, w.shape[1], w.shape[0], 3)
        conv0 = ops.ConvLayer(f"{tag}.conv0x2", ops.CONV_S1, 3, w.shape[1], w.shape[0], ops.pack_direct(w, False),
                              None if wm is None else wm.to(w.device), torch.cat((sc_s, sc_h)).detach().contiguous(),
                              torch.cat((sh_s, sh_h)).detach().contiguous(), True, None if ww is None else ww.to(w.device))
        self._packed = (conv0, s.pack(tag + ".small"), h.pack(tag + ".huge"))

    def run(self, sim: torch.Tensor, backend: str, side: Optional[torch.cuda.Stream] = None, regress=None) -> torch.Tensor:
        """sim [2,D,H,W] -> logits [4,D,H,W] (cat(small, huge), module.py:348,356).  ``side``: HIP stream for the
        `huge` branch (None: both branches back to back on the current stream).
        ``regress`` = (hypotheses, interval, alpha): the `prob` heads regress their own channels (ops.prob_regress) and the result is
        depth_sub_plus [4,H,W] instead of the logits -- only for shapes ops.prob_fusable accepts; a head whose shape the kernel
        declines (returns False) makes the whole call fall back to logits, signalled by a 4-D result."""
        conv0, small, huge = self._packed
        b = conv0.cout // 2
        c0 = ops.conv3d(sim, conv0, backend=backend)
        if regress is not None:
            dsp = torch.empty((4,) + tuple(sim.shape[2:]), dtype=torch.float32, device=sim.device)
            main = torch.cuda.current_stream()
            if side is not None:
                side.wait_stream(main)
            ok = True
            for i, L in enumerate((small, huge)):
                ctx = torch.cuda.stream(side) if (side is not None and i == 1) else _NullCtx()
                with ctx:
                    ok = self._branch(c0[i * b:(i + 1) * b], L, dsp[2 * i:2 * i + 2], backend, regress) and ok
            if side is not None:
                main.wait_stream(side)
                for t in (c0, dsp):
                    t.record_stream(side)
            if ok:
                return dsp
            return self.run(sim, backend, side)     # (never at the product's shapes: W % 4 and alignment are checked up front)
        logits = torch.empty((4,) + tuple(sim.shape[1:]), dtype=torch.float32, device=sim.device)
        # The two U-Nets are independent (module.py:347-348): the `huge` branch runs on a second HIP stream so
        # its kernels fill the load / epilogue stalls of the `small` branch's kernels (and vice versa).
        main = torch.cuda.current_stream()
        if side is not None:
            side.wait_stream(main)
        for i, L in enumerate((small, huge)):
            ctx = torch.cuda.stream(side) if (side is not None and i == 1) else _NullCtx()
            with ctx:
                self._branch(c0[i * b:(i + 1) * b], L, logits[2 * i:2 * i + 2], backend)
        if side is not None:
            main.wait_stream(side)
            for t in (c0, logits):
                t.record_stream(side)
        return logits

    @staticmethod
    def _branch(x0, L, out, backend, regress=None):
        """One U-Net (CostRegNet_part.forward, module.py:389-398; _part_refine 426-436) on its conv0 slice."""
        def conv(x, name):
            # depth-1 volumes take the 2D form of a stride-1 3D layer (see pack)
            d1 = L.get(name + "@d1")
            use = d1 if (d1 is not None and x.shape[1] == 1 and backend != "direct") else L[name]
            return ops.conv3d(x, use, backend=backend)
        c2 = conv(ops.conv3d(x0, L["conv1"], backend=backend), "conv2")
        c4 = conv(ops.conv3d(c2, L["conv3"], backend=backend), "conv4")
        y = conv(ops.conv3d(c4, L["conv5"], backend=backend), "conv6")
        y = ops.conv3d(y, L["conv7"], skip=c4, backend=backend)   # conv4 + deconv(...)  module.py:394,431
        y = ops.conv3d(y, L["conv9"], skip=c2, backend=backend)
        # (r03 built the tail conv11 + skip + prob as ONE depth-marching kernel: parity-green, 1.65x slower than these two
        # launches -- profiles/r03_c_tail_fusion_knockouts.txt; removed in r04)
        y = ops.conv3d(y, L["conv11"], skip=x0, backend=backend)
        if regress is not None:
            return ops.prob_regress(y, L["prob"], regress[0], regress[1], regress[2], out)
        ops.conv3d(y, L["prob"], out=out, backend=backend)
        return True


class CostAgg(nn.Module):
    """Plane-sweep similarity volume (mvsnet.py:102-153, "variance" mode).  Owns no parameters."""

    def __init__(self, mode="variance", in_channels=None):
        super().__init__()
        assert mode in ("variance", "adaptive"), "Don't support {}!".format(mode)
        if mode == "adaptive":
            raise NotImplementedError("agg_mode='adaptive' is unreachable from the reference's scripts "
                                      "(SURVEY.md section 2 row 8) and is not built")
        self.mode = mode

    @staticmethod
    def forward(ref_q4, src_q4, proj12, depth_dhw, group=None):
        """Features quad-planar [C/4,H,W,4]; returns [2,D,H,W].  With ``group`` the local source views are a shard
        and the partial volumes are summed over the process group (RCCL all-reduce).  One kernel, one launch
        configuration per shape (a function of C and D only): results are reproducible from run to run and rank to rank."""
        sim = ops.warp_corr(ref_q4, src_q4, proj12, depth_dhw, layout="q4")
        if group is not None:
            import torch.distributed as dist
            dist.all_reduce(sim, op=dist.ReduceOp.SUM, group=group)
        return sim


class DepthNet(nn.Module):
    """Dual-depth regression (mvsnet.py:11-100).  Owns no parameters; both entry points run K4."""

    def __init__(self, mode="regression"):
        super().__init__()

    @staticmethod
    def forward(cost_reg, depth_values, interval, want_prob=True, want_depth_values=True):
        """``depth_values``: [D,H,W] or ops.AffinePlanes (then the volume of the output dict is only formed on request)."""
        if cost_reg.dim() == 3:      # [4,H,W]: the expectations of the fused `prob` heads (CostRegNet.run(regress=...))
            dsp, prob = cost_reg, None
            hyps, conf = ops.depth_select(dsp, interval, 0)
        else:
            dsp, hyps, conf, prob = ops.depth_regress(cost_reg, depth_values, interval, 1.0, 0, want_prob)
        out = {"photometric_confidence": conf.unsqueeze(0), "depth_sub_plus": dsp.unsqueeze(0),
               "depth_values_c": hyps.unsqueeze(0), "interval": interval}
        if want_depth_values:
            vol = depth_values.volume() if isinstance(depth_values, ops.AffinePlanes) else depth_values
            out["depth_values"] = vol.unsqueeze(0)
        if prob is not None:
            # the key exists only when the volume does: the reference's eval driver maps tensor2numpy over the whole
            # dict (model.py:347, tools.py:108-115) and raises on a None leaf
            out["prob_volume"] = prob.unsqueeze(0)
        return out

    @staticmethod
    def refine(cost_reg, depth_values, interval, alpha=5):
        if cost_reg.dim() == 3:
            dsp = cost_reg
            depth, conf = ops.depth_select(dsp, interval, 1)
        else:
            dsp, depth, conf, _ = ops.depth_regress(cost_reg, depth_values, interval, float(alpha), 1, False)
        return {"depth": depth.unsqueeze(0), "photometric_confidence_refine": conf.unsqueeze(0),
                "depth_sub_plus_refine": dsp.unsqueeze(0)}


# ----------------------------------------------------------------------------------- the boundary
def _stack_outputs(outs):
    """Per-sample output dicts -> one dict batched along axis 0 (0-dim entries such as `interval`: the first sample's,
    it depends on the depth range only through values every sample of an eval batch shares)."""
    first = outs[0]
    res = {}
    for k, v in first.items():
        if isinstance(v, dict):
            res[k] = _stack_outputs([o[k] for o in outs])
        elif v.dim() == 0:
            res[k] = v
        else:
            res[k] = torch.cat([o[k] for o in outs], 0)
    return res


def shard_source_views(num_views: int, world_size: int, rank: int) -> List[int]:
    """Source-view indices (1-based into the V views) owned by ``rank``: {v : (v-1) % G == g}."""
    return [v for v in range(1, num_views) if (v - 1) % world_size == rank]


class MVSNet(nn.Module):
    """Drop-in for networks.mvsnet.MVSNet (mvsnet.py:156-260)."""

    def __init__(self, ndepths, depth_interval_ratio, cr_base_chs=None, fea_mode="fpn", agg_mode="variance",
                 depth_mode="regression", winner_take_all_to_generate_depth=True, inverse_depth=False,
                 verbose=True):
        super().__init__()
        if cr_base_chs is None:
            cr_base_chs = [8] * len(ndepths)
        self.ndepths = list(ndepths)
        self.depth_interval_ratio = list(depth_interval_ratio)
        self.fea_mode = fea_mode
        self.cr_base_chs = cr_base_chs
        self.num_stage = len(ndepths)
        self.inverse_depth = inverse_depth
        if verbose:  # the reference prints its configuration on construction (mvsnet.py:169-174)
            print("netphs:", ndepths)
            print("depth_intervals_ratio:", depth_interval_ratio)
            print("cr_base_chs:", cr_base_chs)
            print("fea_mode:", fea_mode)
            print("agg_mode:", agg_mode)
            print("depth_mode:", depth_mode)
        assert len(ndepths) == len(depth_interval_ratio)
        if fea_mode != "fpn":
            raise NotImplementedError("only fea_mode='fpn' is reachable from the reference's scripts")
        if any(c != 8 for c in cr_base_chs):
            raise NotImplementedError("kernels are compiled for cr_base_chs = 8 (mvsnet.py:160-161)")

        self.feature = FeatureNet(base_channels=8)
        self.cost_aggregation = CostAgg(agg_mode, self.feature.out_channels)
        self.cost_regularization = nn.ModuleList(
            [CostRegNet(2, self.cr_base_chs[i]) for i in range(self.num_stage)])
        self.cost_regularization_refine = nn.ModuleList(
            [CostRegNet(2, self.cr_base_chs[i], refine=True) for i in range(self.num_stage)])
        self.DepthNet = DepthNet(depth_mode)

        # knobs outside the reference's interface
        self.return_prob_volume = True      # eval never reads prob_volume; bench turns it off
        self.return_depth_values = True     # ditto for the [1,D,H,W] hypothesis volume of the output dict
        self.affine_hypotheses = True       # linear sampling: planes = base + d * interval formed inside K1 / K4 (N2)
        self.feature_dtype = "f32"          # "f16": FeatureNet's outputs stored as fp16, K1 reads them with fp32
                                            # accumulation (EXTENSION, BASELINE configs[4]; the reference is fp32 only)
        self.conv_backend = "auto"          # "auto" | "direct" | "mfma"
        self.two_streams = True             # run the small / huge regularisation branches on two HIP streams
        self.feature_async_topdown = True   # FeatureNet's level-2/3 outputs on a third stream, under stage 1 (+1.3 %, r02)
        self.feature_group_views = None     # views per FeatureNet call (None: as many as fit a 2 GB activation)
        self.view_group = None              # torch.distributed group for source-view sharding (set_view_shard)
        self.view_rank, self.view_world = 0, 1
        self.shard_rows = False             # latency mode v2: H-slab regularisation over the view group
        self.row_collective = "reduce_scatter"   # ... fed by reduce_scatter + halo exchange | "all_reduce" + slice
        self.use_graph = False              # replay the whole forward as one HIP graph (static shapes; see forward)
        self._graph = None                  # (key, graph, static inputs, static outputs)
        self._packed_key = None
        self._streams = {}                  # per instance: (device index, role) -> side stream
        self.comm_log = None                # a list: every collective of the view-shard modes appends (kind, bytes sent, bytes received)
        self.eval()

    # -- lifecycle ---------------------------------------------------------------------------------
    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("dmvsnet_amd.MVSNet is the inference hot path; training is out of scope")
        return super().train(False)

    def _invalidate(self):
        self._packed_key = None
        self._fp_cache = None
        self.feature._packed = None

    def load_state_dict(self, state_dict, strict=True, **kw):
        # model.py:65-70 drops attn_mask keys before a strict load
        sd = {k: v for k, v in state_dict.items() if "attn_mask" not in k}
        r = super().load_state_dict(sd, strict=strict, **kw)
        self._invalidate()
        return r

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self._invalidate()
        return r

    def stage_level(self, s: int) -> int:
        """Feature-pyramid level (0 = 1/4 resolution ... 2 = full) of stage ``s``.  The reference is hard-wired to
        <= 3 stages, where this is ``s`` (mvsnet.py:214-216; a 4th stage raises KeyError 'stage4' there).  EXTENSION
        (BASELINE configs[4], 4 stages): the three FPN levels are kept and the extra stages run at the coarsest one --
        levels 0, 0, 1, 2 -- with a same-resolution hypothesis transition between them."""
        return s if self.num_stage <= 3 else max(0, s - (self.num_stage - 3))

    def _side_stream(self, device, role):
        """Side streams belong to the instance (two models, or several maps in flight, must not share one)."""
        key = (device.index, role, torch.cuda.current_stream(device).cuda_stream)
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=device)
        return self._streams[key]

    def set_view_shard(self, group, rank: int, world: int, shard_rows: bool = False, row_collective: Optional[str] = None):
        """Shard the source views of every depth map over ``group`` (one process per GPU, RCCL sum).
        ``shard_rows``: additionally regularise only this rank's H-slab (+ halo) of the summed volume and
        all-gather the regression outputs (latency mode v2)."""
        self.view_group, self.view_rank, self.view_world = group, rank, world
        self.shard_rows = bool(shard_rows)
        if row_collective is not None:
            if row_collective not in ("reduce_scatter", "all_reduce"):
                raise DmvsError(f"row_collective must be 'reduce_scatter' or 'all_reduce', not {row_collective!r}")
            self.row_collective = row_collective
        if self.shard_rows and self.return_prob_volume:
            # the slab path regresses without the softmax volume: say so instead of dropping the key silently
            import warnings
            warnings.warn("shard_rows: prob_volume is not produced on the H-slab path (set return_prob_volume=False)")

    def _fingerprint(self, device):
        """Cheap identity of every weight the packed copies were made from: storage address + in-place version.
        Catches load_state_dict on a sub-module, in-place edits and .to() -- anything the top-level hooks miss.
        The tensor LIST is cached (invalidated by load_state_dict / _apply); a forward then only compares ~800
        (pointer, version) pairs it reads through the cached references."""
        if getattr(self, "_fp_cache", None) is None:
            self._fp_cache = list(self.parameters()) + list(self.buffers())
        return (device,) + tuple((t.data_ptr(), t._version) for t in self._fp_cache)

    @torch.no_grad()
    def prepare(self, device):
        """Fold BatchNorm and pack every weight into kernel layout (once per device and weight version)."""
        pdev = next(self.parameters()).device
        if pdev != device:
            raise DmvsError(f"module parameters are on {pdev} but the inputs on {device}: call .to(device) first "
                            "(kernels take raw device pointers)")
        key = self._fingerprint(device)
        if self._packed_key == key:
            return
        for i in range(self.num_stage):
            self.cost_regularization[i].pack(f"reg{i}")
            self.cost_regularization_refine[i].pack(f"ref{i}")
        self.feature.pack()
        self._packed_key = key

    # -- latency mode v2: H-slab regularisation ------------------------------------------------------
    # Receptive field of the regularisation U-Net along H, in full-resolution rows: prob 1, conv11/9/7 (gather form of
    # the transposed convs) 1 row at 1/2, 1/4, 1/8 resolution, conv6 1 at 1/8, conv5 + conv4 at 1/4 ..., conv1 +
    # conv0 at full resolution: 30 rows (+ rounding of the stride-2 grids); slabs are multiples of 8 rows so the
    # three stride-2 levels and K4's (row % 4, col % 2) patterns line up with the unsharded run.  32 = the radius
    # rounded up to 8; it suffices only while slabs stay 8-aligned (rows r0 - 30 .. r1 + 22 are needed).  With the
    # direct-form kernels (ops.use_wino = False) the owned rows equal the replicated run BIT FOR BIT -- asserted with
    # torch.equal by tests/test_dist_gpu.py's direct-form case, which is what protects this constant; the default Winograd
    # layers re-associate where a slab's 2x2 output tiling differs from the full volume's (asserted: rel < 1e-6).
    ROW_HALO = 32

    @staticmethod
    def row_slabs(h: int, world: int):
        """Rows [r0, r1) owned by each rank (multiples of 8; trailing ranks may own fewer or none) and the padded
        slab height every rank exchanges in the collectives."""
        per = -(-h // (8 * world)) * 8
        return [(min(g * per, h), min((g + 1) * per, h)) for g in range(world)], per

    @classmethod
    def row_extent(cls, h: int, r0: int, r1: int):
        """Rows a rank regularises: its slab + ROW_HALO each side (an empty slab: a dummy 8-row block, results unused)."""
        if r1 <= r0:
            return 0, min(8, h)
        return max(0, r0 - cls.ROW_HALO), min(h, r1 + cls.ROW_HALO)

    def _reduce_rows(self, part: torch.Tensor, h: int) -> torch.Tensor:
        """Partial similarity volume of the local source views [2,D,h,w] -> this rank's extended slab
        [2,D,e1-e0,w] of the volume summed over the view group.

        ``row_collective = "reduce_scatter"``: the partial volumes are reduce-scattered along H
        (every rank RECEIVES only its own `per` rows: S / G bytes instead of the whole S of an all-reduce), then the
        halo rows come from the ranks that own them by point-to-point send / recv (<= 2 * ROW_HALO rows per rank).
        ``"all_reduce"``: r02's form -- every rank receives the whole volume and slices (kept: same bits, and the
        fallback where a backend lacks reduce_scatter)."""
        import torch.distributed as dist
        G, rank = self.view_world, self.view_rank
        slabs, per = self.row_slabs(h, G)
        r0, r1 = slabs[rank]
        e0, e1 = self.row_extent(h, r0, r1)
        log = self.comm_log
        if self.row_collective == "all_reduce":
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.view_group)
            if log is not None:
                log.append(("all_reduce", 4 * part.numel(), 4 * part.numel()))
            return part[:, :, e0:e1].contiguous()
        P, w = part.shape[0] * part.shape[1], part.shape[-1]
        send = part.new_zeros((G * per, P, w))            # rows first: a rank's slab is one contiguous chunk
        send[:h] = part.permute(2, 0, 1, 3).reshape(h, P, w)
        own = part.new_empty((per, P, w))
        dist.reduce_scatter_tensor(own, send, op=dist.ReduceOp.SUM, group=self.view_group)
        if log is not None:
            log.append(("reduce_scatter", 4 * send.numel(), 4 * own.numel()))
        ext = part.new_zeros((e1 - e0, P, w))
        if r1 > r0:
            ext[r0 - e0:r1 - e0] = own[:r1 - r0]
        # gloo has no point-to-point path for device tensors (it is only used to exercise the multi-rank code on a
        # one-GPU box): stage its halo messages through the host.  RCCL sends / receives device memory directly.
        via_host = part.is_cuda and dist.get_backend(self.view_group) == "gloo"
        ops_, keep, landed = [], [], []
        # P2POp's peer is a GLOBAL rank; g is the rank inside the view group (several view groups per job: ADVICE r03)
        world_group = self.view_group is None or self.view_group is dist.group.WORLD
        for g in range(G):
            if g == rank:
                continue
            peer = g if world_group else dist.get_global_rank(self.view_group, g)
            g0, g1 = slabs[g]
            ge0, ge1 = self.row_extent(h, g0, g1)
            a, b = max(ge0, r0), min(ge1, r1)              # rows of mine inside g's extended slab
            if g1 > g0 and a < b:
                t = own[a - r0:b - r0].contiguous()
                t = t.cpu() if via_host else t
                keep.append(t)
                ops_.append(dist.P2POp(dist.isend, t, peer, group=self.view_group))
            a, b = max(e0, g0), min(e1, g1)                # rows of g inside my extended slab
            if r1 > r0 and a < b:
                dst = ext[a - e0:b - e0]
                buf = torch.empty(dst.shape, dtype=dst.dtype) if via_host else dst
                landed.append((dst, buf))
                ops_.append(dist.P2POp(dist.irecv, buf, peer, group=self.view_group))
        if ops_:
            for req in dist.batch_isend_irecv(ops_):
                req.wait()
        if log is not None:
            log.append(("halo_p2p", 4 * sum(t.numel() for t in keep), 4 * sum(d.numel() for d, _ in landed)))
        if via_host:
            for dst, buf in landed:
                dst.copy_(buf)
        return ext.reshape(e1 - e0, part.shape[0], part.shape[1], w).permute(1, 2, 0, 3).contiguous()

    def _gather_rows(self, planes: torch.Tensor, r0: int, r1: int, e0: int, h: int, per: int) -> torch.Tensor:
        """planes [P, he, w] computed on the extended slab starting at row e0 -> [P, h, w] on every rank
        (all-gather of the owned rows [r0, r1), padded to `per` rows)."""
        import torch.distributed as dist
        P, _, w = planes.shape
        send = torch.zeros((P, per, w), dtype=planes.dtype, device=planes.device)
        if r1 > r0:
            send[:, :r1 - r0] = planes[:, r0 - e0:r1 - e0]
        recv = torch.empty((self.view_world * P, per, w), dtype=planes.dtype, device=planes.device)   # rank-major
        dist.all_gather_into_tensor(recv, send, group=self.view_group)
        if self.comm_log is not None:
            self.comm_log.append(("all_gather", 4 * send.numel(), 4 * recv.numel()))
        return recv.view(self.view_world, P, per, w).permute(1, 0, 2, 3).reshape(P, self.view_world * per, w)[:, :h].contiguous()

    def _stage_rows(self, s, half, local, proj12, hyp, interval, C, reg_side):
        """One stage (main + refine pass) with the regularisation and regression restricted to this rank's H-slab
        (+ ROW_HALO rows each side) of the summed similarity volume; the regression outputs of the owned rows are
        all-gathered, so every rank ends with the full-size outputs of the unsharded run (the kernels see the same
        neighbourhoods: identical bits with the direct-form kernels, fp32 re-association level with the Winograd layers,
        whose 2x2 output tiling is anchored at the slab's first row)."""
        D, h, w = hyp.shape
        slabs, per = self.row_slabs(h, self.view_world)
        r0, r1 = slabs[self.view_rank]
        e0, e1 = self.row_extent(h, r0, r1)

        part = self.cost_aggregation.forward(half(0, 0), [half(v, 0) for v in local], proj12, hyp, None)
        sim = self._reduce_rows(part, h)
        hyp_e = ops.planes_rows(hyp, e0, e1)
        cost_reg = self.cost_regularization[s].run(sim, self.conv_backend, reg_side)
        dsp, hyps, conf, prob = ops.depth_regress(cost_reg, hyp_e, interval, 1.0, 0, False)
        g = self._gather_rows(torch.cat((dsp, hyps, conf[None]), 0), r0, r1, e0, h, per)
        out_main = {"photometric_confidence": g[8:9], "depth_sub_plus": g[None, 0:4], "depth_values_c": g[None, 4:8],
                    "interval": interval}
        if self.return_depth_values:
            out_main["depth_values"] = (hyp.volume() if isinstance(hyp, ops.AffinePlanes) else hyp).unsqueeze(0)

        hyp_c = out_main["depth_values_c"][0].contiguous()
        part_c = self.cost_aggregation.forward(half(0, C), [half(v, C) for v in local], proj12, hyp_c, None)
        sim_c = self._reduce_rows(part_c, h)
        cost_reg_c = self.cost_regularization_refine[s].run(sim_c, self.conv_backend, reg_side)
        dsp_r, depth, conf_r, _ = ops.depth_regress(cost_reg_c, hyp_c[:, e0:e1].contiguous(), interval, 5.0, 1, False)
        g = self._gather_rows(torch.cat((dsp_r, depth[None], conf_r[None]), 0), r0, r1, e0, h, per)
        out_ref = {"depth": g[4:5], "photometric_confidence_refine": g[5:6], "depth_sub_plus_refine": g[None, 0:4]}
        return out_main, out_ref

    # -- forward -----------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, imgs, proj_matrices, depth_values):
        """imgs [1,V,3,H,W]; proj_matrices {"stageK": [1,V,2,4,4]}; depth_values [1,n] (mvsnet.py:188).

        ``use_graph``: the ~170 kernel launches of a depth map are captured once per input shape into a HIP graph
        (torch.cuda.CUDAGraph: stream capture of the same launches, side streams included) and replayed; the host then
        enqueues one graph instead of ~170 kernels and ~150 allocations.  The returned tensors live in the graph's
        memory pool and are OVERWRITTEN by the next forward: copy what must outlive it (the eval driver converts the
        outputs to NumPy right after the call, model.py:347).  Not combined with view sharding (collectives)."""
        if imgs.dim() == 5 and imgs.shape[0] > 1:
            # the kernels work on one depth map (the eval loader's batch, model.py:330-336); a larger batch is its
            # samples one after the other, stacked back along the batch axis like the reference's outputs
            if self.use_graph:   # a replay overwrites the static outputs of the previous sample
                raise DmvsError("use_graph replays one depth map at a time; batch > 1 is not combined with it")
            outs = [self.forward(imgs[b:b + 1], {k: v[b:b + 1] for k, v in proj_matrices.items()}, depth_values[b:b + 1])
                    for b in range(imgs.shape[0])]
            return _stack_outputs(outs)
        if self.use_graph and self.view_group is None and imgs.is_cuda:
            return self._forward_graph(imgs, proj_matrices, depth_values)
        return self._forward(imgs, proj_matrices, depth_values)

    def _forward_graph(self, imgs, proj_matrices, depth_values):
        self.prepare(imgs.device)
        key = (self._packed_key, tuple(imgs.shape), tuple(depth_values.shape),
               tuple((k, tuple(v.shape)) for k, v in sorted(proj_matrices.items())), self.return_prob_volume,
               self.return_depth_values, self.affine_hypotheses,
               self.two_streams, self.conv_backend, self.feature_async_topdown, self.feature_group_views)
        if self._graph is None or self._graph[0] != key:
            self._graph = None
            s_imgs, s_dv = imgs.clone(), depth_values.clone()
            s_proj = {k: v.clone() for k, v in proj_matrices.items()}
            warm = torch.cuda.Stream(device=imgs.device)   # eager passes first: K1 autotune, LDS-size attributes
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):
                for _ in range(2):
                    self._forward(s_imgs, s_proj, s_dv)
            torch.cuda.current_stream().wait_stream(warm)
            torch.cuda.synchronize(imgs.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._forward(s_imgs, s_proj, s_dv)
            self._graph = (key, graph, (s_imgs, s_proj, s_dv), out)
        _, graph, (s_imgs, s_proj, s_dv), out = self._graph
        s_imgs.copy_(imgs)
        s_dv.copy_(depth_values)
        for k, v in proj_matrices.items():
            s_proj[k].copy_(v)
        graph.replay()
        return out

    @torch.no_grad()
    def _forward(self, imgs, proj_matrices, depth_values):
        if not imgs.is_cuda:
            raise DmvsError("dmvsnet_amd.MVSNet runs on a HIP device only (no CPU fallback); move the module and "
                            "its inputs to 'cuda' -- the CPU restatement is oracle/dmvs_oracle.py (tests only)")
        assert imgs.shape[0] == 1   # forward() splits larger batches
        self.prepare(imgs.device)
        V = imgs.size(1)
        H, W = imgs.shape[-2:]
        depth_values = depth_values.contiguous()
        local = list(range(1, V)) if self.view_group is None else shard_source_views(V, self.view_world, self.view_rank)
        rows = self.shard_rows and self.view_group is not None and self.view_world > 1

        # step 1: features of the reference view and of the local source views, all views in ONE batched call
        # (the reference loops over views, mvsnet.py:199-202; per-view results are identical)
        ops.mark("features")
        views = [0] + local
        batch = imgs[0] if len(views) == V else imgs[0, views]
        # view groups: a [32][g][H][W] activation must stay below the 2 GB range of a buffer descriptor
        gmax = self.feature_group_views or max(1, ((1 << 29) - 1) // (32 * H * W))
        groups = [list(range(i, min(i + gmax, len(views)))) for i in range(0, len(views), gmax)]
        side = self._side_stream(imgs.device, "fpn") if (self.feature_async_topdown and len(groups) == 1) else None
        self.feature._topdown_done = None
        # relative projections of every stage first: they depend on the cameras only, and a kernel launched between K4 of one
        # stage and K1 of the next would sit in the dependent chain (every kernel boundary is a cache write-back + a dispatch
        # gap of several microseconds on this multi-XCD GPU)
        # a loader that emits only the reference's three projection scales (general_eval.py:189-198) serves a deeper
        # pyramid BY LEVEL; one written for the extension carries an entry per stage ("stage1" .. "stageS")
        per_stage = self.num_stage <= 3 or "stage{}".format(self.num_stage) in proj_matrices
        pkeys = ["stage{}".format((s if per_stage else self.stage_level(s)) + 1) for s in range(self.num_stage)]
        proj_rel = {k: ops.relative_proj(proj_matrices[k][0].contiguous()) for k in dict.fromkeys(pkeys)}   # [V-1,12] each
        stacks = [self.feature.run(batch[g[0]:g[-1] + 1].contiguous(), side) for g in groups]   # each: 3 x [2, g, C/4, h, w, 4]
        if self.feature_dtype == "f16":
            if W % 8:
                raise DmvsError("feature_dtype='f16' needs an image width that is a multiple of 8 (pixel pairs at 1/4 scale)")
            if self.feature._topdown_done is not None:       # the casts read the side stream's outputs
                torch.cuda.current_stream().wait_event(self.feature._topdown_done)
                self.feature._topdown_done = None
            stacks = [tuple(o.half() for o in st) for st in stacks]   # (a cast kernel: layout plumbing)
        elif self.feature_dtype != "f32":
            raise DmvsError(f"feature_dtype must be 'f32' or 'f16', not {self.feature_dtype!r}")
        slot = {v: (gi, k) for gi, g in enumerate(groups) for k, i in enumerate(g) for v in [views[i]]}
        reg_side = self._side_stream(imgs.device, "reg") if self.two_streams else None

        outputs = {}
        last_depth = None
        last_level = 0
        for s in range(self.num_stage):
            key = "stage{}".format(s + 1)
            level = self.stage_level(s)          # FPN level of the stage: = s for the reference's <= 3 stages
            scale = 2 ** (3 - level - 1)         # mvsnet.py:214
            h, w = H // scale, W // scale
            D = self.ndepths[s]
            ops.mark(key)
            if level >= 1 and self.feature._topdown_done is not None:
                torch.cuda.current_stream().wait_event(self.feature._topdown_done)
                self.feature._topdown_done = None
            if s == 0:
                hyp, interval = ops.hypotheses_first(depth_values, D, h, w, self.inverse_depth, self.affine_hypotheses)
            else:
                hyp, interval = ops.hypotheses_next(last_depth, depth_values, self.depth_interval_ratio[s], D,
                                                    self.inverse_depth, self.affine_hypotheses,
                                                    up=2 if level > last_level else 1)
            last_level = level
            proj_all = proj_rel[pkeys[s]]
            proj12 = proj_all[[v - 1 for v in local]].contiguous() if len(local) != V - 1 else proj_all
            C = self.feature.out_channels[level]

            def half(v, c0, level=level):
                return stacks[slot[v][0]][level][1 if c0 else 0, slot[v][1]]   # [C/4, h, w, 4], contiguous

            if rows:
                out_main, out_ref = self._stage_rows(s, half, local, proj12, hyp, interval, C, reg_side)
            else:
                sim = self.cost_aggregation.forward(half(0, 0), [half(v, 0) for v in local], proj12, hyp, self.view_group)
                fuse = not self.return_prob_volume and ops.prob_fusable(D, w, self.conv_backend)
                cost_reg = self.cost_regularization[s].run(sim, self.conv_backend, reg_side, (hyp, interval, 1.0) if fuse else None)
                out_main = self.DepthNet.forward(cost_reg, hyp, interval, self.return_prob_volume, self.return_depth_values)

                hyp_c = out_main["depth_values_c"][0]
                sim_c = self.cost_aggregation.forward(half(0, C), [half(v, C) for v in local], proj12, hyp_c,
                                                      self.view_group)
                fuse_c = ops.prob_fusable(4, w, self.conv_backend)
                cost_reg_c = self.cost_regularization_refine[s].run(sim_c, self.conv_backend, reg_side, (hyp_c, interval, 5.0) if fuse_c else None)
                out_ref = self.DepthNet.refine(cost_reg_c, hyp_c, interval)

            outputs_stage = {**out_ref, **out_main}          # mvsnet.py:254
            last_depth = outputs_stage["depth"][0]
            outputs[key] = outputs_stage
            outputs.update(outputs_stage)
        ops.mark("end")
        return outputs
