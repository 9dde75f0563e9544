"""Thin Python wrappers over the C ABI (include/dmvs.h): tensors in, kernels enqueued on torch's
current HIP stream, tensors out.  PyTorch is used for device memory and streams only.

Every function requires CUDA(HIP) fp32 contiguous tensors and raises otherwise: there is no CPU path
in the product (the CPU restatement lives in oracle/ and is test infrastructure).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch

from . import _lib

CONV_S1, CONV_S2, DECONV_S2, CONV2D_K5S2, CONV2D_K1 = 0, 1, 2, 3, 4
RELU, SKIP_UP2, OUT_Q4, IN_VIEWS = 1, 2, 8, 16   # include/dmvs.h (bit value 4 is retired)


class KernelTimer:
    """HIP-event timing of individual kernel launches on torch's current stream (the stream every dmvs
    kernel is enqueued on).  Used by bench.py for the live roofline numbers; off (None) by default."""

    def __init__(self):
        self.records = []   # (family, start_event, end_event, flops, bytes, executed flops)
        self.labels = []    # per record: layer name (or the family)
        self.marks = []     # (label, event)
        self._pool = []

    def _event(self):
        return self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)

    def reserve(self, n):
        self._pool.extend(torch.cuda.Event(enable_timing=True) for _ in range(n))

    def begin(self):
        e = self._event()
        e.record()
        return e

    def end(self, family, start, flops, nbytes, exec_flops=None, label=None):
        """``flops`` = algorithmic (direct-form) FLOPs of the launch; ``exec_flops`` = what its MFMAs actually execute when
        that differs (Winograd layers: 16 products per 2x2 output patch instead of 36); ``label`` = the layer / kernel the launch
        belongs to (bench.py: `roofline.largest_kernel`)."""
        e = self._event()
        e.record()
        self.records.append((family, start, e, flops, nbytes, flops if exec_flops is None else exec_flops))
        self.labels.append(label or family)

    def mark(self, label):
        e = self._event()
        e.record()
        self.marks.append((label, e))

    def summary(self):
        """family -> dict(launches, ms, sum_ms, flops, bytes); call after torch.cuda.synchronize().
        ``ms`` is the family's BUSY time: the union of its launches' [start, end] intervals on the device timeline
        (kernels of the two regularisation branches run concurrently on two streams, so summing durations would
        count the overlap twice); ``sum_ms`` is the plain sum of the launch durations."""
        if not self.records:
            return {}
        base = self.records[0][1]
        per = {}
        for fam, s, e, fl, nb, xf in self.records:
            d = per.setdefault(fam, dict(launches=0, sum_ms=0.0, flops=0.0, bytes=0.0, exec_flops=0.0, iv=[]))
            d["exec_flops"] += xf
            t0, t1 = base.elapsed_time(s), base.elapsed_time(e)
            d["launches"] += 1
            d["sum_ms"] += t1 - t0
            d["flops"] += fl
            d["bytes"] += nb
            d["iv"].append((t0, t1))
        for d in per.values():
            busy, cur0, cur1 = 0.0, None, None
            for t0, t1 in sorted(d.pop("iv")):
                if cur1 is None or t0 > cur1:
                    if cur1 is not None:
                        busy += cur1 - cur0
                    cur0, cur1 = t0, t1
                else:
                    cur1 = max(cur1, t1)
            d["ms"] = busy + (cur1 - cur0 if cur1 is not None else 0.0)
        return per

    def by_label(self):
        """label -> dict(launches, sum_ms, flops, exec_flops, bytes): plain sums of the launch durations per layer name."""
        out = {}
        for (fam, s, e, fl, nb, xf), lab in zip(self.records, self.labels):
            d = out.setdefault(lab, dict(family=fam, launches=0, sum_ms=0.0, flops=0.0, exec_flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["sum_ms"] += s.elapsed_time(e)
            d["flops"] += fl
            d["exec_flops"] += xf
            d["bytes"] += nb
        return out

    def spans(self):
        """Elapsed ms between consecutive marks, summed per label of the span's START mark."""
        out = {}
        for (la, ea), (_, eb) in zip(self.marks[:-1], self.marks[1:]):
            out[la] = out.get(la, 0.0) + ea.elapsed_time(eb)
        return out


timer: Optional[KernelTimer] = None
# Optional launch log (profiling tooling): the family name of every K1..K4 launch, in host order, since it was set to
# a list.  scripts/pmc_summary.py joins it with rocprofv3's dispatch order to attribute PMC counters per family
# (kernel names alone cannot tell a FeatureNet launch of the MFMA conv kernel from a regularisation launch).
launch_log: Optional[list] = None


def _log(family: str) -> None:
    if launch_log is not None:
        launch_log.append(family)


def mark(label: str) -> None:
    if timer is not None:
        timer.mark(label)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.DmvsError("dmvsnet_amd kernels need tensors on a HIP device (no CPU fallback); got "
                                 f"{t.device}")
        if t.dtype != torch.float32:
            raise RuntimeError(f"expected float32 tensor, got {t.dtype}")  # same class the reference raises
        if not t.is_contiguous():
            raise _lib.DmvsError("non-contiguous tensor passed to a dmvs kernel")


# ------------------------------------------------------------------------------------------ layout
def nchw_to_hwc(src: torch.Tensor, c0: int, C: int) -> torch.Tensor:
    """src [1,Ct,H,W] or [Ct,H,W] -> [H,W,C] of channels c0..c0+C."""
    _req(src)
    Ct, H, W = src.shape[-3:]
    assert 0 <= c0 and c0 + C <= Ct
    dst = torch.empty((H, W, C), dtype=torch.float32, device=src.device)
    _lib.check(_lib.load().dmvs_nchw_to_hwc(_ptr(src), c0, C, H, W, _ptr(dst), _stream()), "dmvs_nchw_to_hwc")
    return dst


def planar_to_hwc(stack: torch.Tensor, view: int, c0: int, C: int) -> torch.Tensor:
    """stack [Ct,V,H,W] (views as depth slices) -> [H,W,C] of channels c0..c0+C of one view."""
    _req(stack)
    Ct, V, H, W = stack.shape
    assert 0 <= c0 and c0 + C <= Ct and 0 <= view < V
    dst = torch.empty((H, W, C), dtype=torch.float32, device=stack.device)
    src = ctypes.c_void_p(stack.data_ptr() + view * H * W * 4)
    _lib.check(_lib.load().dmvs_planar_to_hwc(src, V * H * W, c0, C, H, W, _ptr(dst), _stream()), "dmvs_planar_to_hwc")
    return dst


def relative_proj(proj_pairs: torch.Tensor) -> torch.Tensor:
    """proj_pairs [V,2,4,4] -> [V-1,12] (rot 9 + trans 3 of src @ inv(ref))."""
    _req(proj_pairs)
    V = proj_pairs.shape[0]
    out = torch.empty((V - 1, 12), dtype=torch.float32, device=proj_pairs.device)
    _lib.check(_lib.load().dmvs_relative_proj(_ptr(proj_pairs), V, _ptr(out), _stream()), "dmvs_relative_proj")
    return out


# ------------------------------------------------------------------------------------------ hypotheses
@dataclass
class AffinePlanes:
    """Linear-depth hypothesis planes in affine form: plane d = base + d * step, d < D.
    ``base`` [H,W]; ``step`` 0-dim device tensor (the stage's interval).  K1 / K4 take it instead of a [D,H,W] volume."""
    base: torch.Tensor
    step: torch.Tensor
    D: int

    @property
    def shape(self):
        return (self.D,) + tuple(self.base.shape)

    @property
    def device(self):
        return self.base.device

    def rows(self, r0: int, r1: int) -> "AffinePlanes":
        return AffinePlanes(self.base[r0:r1].contiguous(), self.step, self.D)

    def volume(self) -> torch.Tensor:
        """The [D,H,W] tensor the reference materialises (same rounding as the kernels: d * step, then + base)."""
        d = torch.arange(self.D, dtype=torch.float32, device=self.base.device).view(-1, 1, 1)
        return self.base[None] + d * self.step


def planes_rows(planes, r0: int, r1: int):
    return planes.rows(r0, r1) if isinstance(planes, AffinePlanes) else planes[:, r0:r1].contiguous()


def hypotheses_first(depth_values: torch.Tensor, D: int, H: int, W: int, inverse: bool, affine: bool = False):
    """-> (planes [D,H,W] or, with ``affine`` (linear sampling only), AffinePlanes; interval)."""
    _req(depth_values)
    n = depth_values.shape[-1]
    itv = torch.empty((), dtype=torch.float32, device=depth_values.device)
    if affine and not inverse:
        base = torch.empty((H, W), dtype=torch.float32, device=depth_values.device)
        _lib.check(_lib.load().dmvs_hypothesis_base_first(_ptr(depth_values), n, D, H, W, _ptr(base), _ptr(itv), _stream()),
                   "dmvs_hypothesis_base_first")
        return AffinePlanes(base, itv, D), itv
    out = torch.empty((D, H, W), dtype=torch.float32, device=depth_values.device)
    _lib.check(_lib.load().dmvs_hypotheses_first(_ptr(depth_values), n, D, H, W, int(inverse), _ptr(out), _ptr(itv),
                                                 _stream()), "dmvs_hypotheses_first")
    return out, itv


def hypotheses_next(last_depth: torch.Tensor, depth_values: torch.Tensor, ratio: float, D: int, inverse: bool,
                    affine: bool = False, up: int = 2):
    """last_depth [h,w] -> planes [D,up*h,up*w] (or AffinePlanes) + interval.  ``up`` = 2: the reference's x2 resize
    between stages; 1: a same-resolution transition (pyramids deeper than three stages, an extension)."""
    _req(last_depth, depth_values)
    h, w = last_depth.shape[-2:]
    n = depth_values.shape[-1]
    itv = torch.empty((), dtype=torch.float32, device=last_depth.device)
    base_only = affine and not inverse
    out = torch.empty(((up * h, up * w) if base_only else (D, up * h, up * w)), dtype=torch.float32, device=last_depth.device)
    _lib.check(_lib.load().dmvs_hypotheses_next_up(_ptr(last_depth), h, w, int(up), _ptr(depth_values), n, float(ratio), D,
                                                   int(inverse), int(base_only), _ptr(out), _ptr(itv), _stream()),
               "dmvs_hypotheses_next_up")
    return (AffinePlanes(out, itv, D) if base_only else out), itv


# ------------------------------------------------------------------------------------------ K1
def hwc_to_q4(f_hwc: torch.Tensor) -> torch.Tensor:
    """[H,W,C] pixel-major -> [C/4,H,W,4] quad-planar (the layout K1's product kernel samples; layout glue for tests and
    tools -- FeatureNet's output epilogue writes quad-planar directly)."""
    H, W, C = f_hwc.shape
    return f_hwc.view(H, W, C // 4, 4).permute(2, 0, 1, 3).contiguous()


def warp_corr(ref: torch.Tensor, src: Sequence[torch.Tensor], proj12: torch.Tensor, depth_dhw: torch.Tensor,
              out: Optional[torch.Tensor] = None, accumulate: bool = False, C: Optional[int] = None,
              pix_stride: Optional[int] = None, variant: int = 0, family: str = "warp_corr",
              layout: Optional[str] = None) -> torch.Tensor:
    """K1.  Features quad-planar [C/4,H,W,4] (``layout="q4"``: the product kernel) or pixel-major [H,W,pix_stride]
    (``"hwc"``: the generic kernel); by default told apart by the trailing dimension (4 = quad-planar; a 4-channel
    pixel-major map does not exist in this network).  proj12 [nsrc,12], depth [D,H,W] (or AffinePlanes) -> sim [2,D,H,W].
    ``variant``: launch-configuration knob of the q4 kernel (0 = default; see dmvs_warp_corr_q4), an explicit argument
    of the C entry point -- no process-wide state."""
    affine = isinstance(depth_dhw, AffinePlanes)
    f16 = ref.dtype == torch.float16      # fp16 FEATURES (extension, q4 layout only); everything else stays fp32
    if f16:
        for t in (ref, *src):
            if not (t.is_cuda and t.dtype == torch.float16 and t.is_contiguous()):
                raise _lib.DmvsError("fp16 features: every view must be a contiguous float16 HIP tensor")
    if affine:
        _req(proj12, depth_dhw.base, depth_dhw.step, *(() if f16 else (ref, *src)))
    else:
        _req(proj12, depth_dhw, *(() if f16 else (ref, *src)))
    D, H, W = depth_dhw.shape
    if layout is None:
        layout = "q4" if (ref.dim() == 4 and ref.shape[-1] == 4) else "hwc"
    if f16 and layout != "q4":
        raise _lib.DmvsError("fp16 features are a quad-planar (q4) extension; the pixel-major kernel is fp32 only")
    if layout == "q4":
        assert ref.dim() == 4 and tuple(ref.shape[1:]) == (H, W, 4), (tuple(ref.shape), (H, W))
        C = 4 * ref.shape[0]
    else:
        pix_stride = ref.shape[-1] if pix_stride is None else pix_stride
        C = pix_stride if C is None else C
    nsrc = len(src)
    if out is None:
        out = torch.empty((2, D, H, W), dtype=torch.float32, device=ref.device)
        assert not accumulate
    if nsrc == 0:  # a view shard with no local source view contributes zeros
        if not accumulate:
            out.zero_()
        return out
    assert proj12.shape[0] == nsrc
    arr = (ctypes.c_void_p * nsrc)(*[s.data_ptr() for s in src])
    lib = _lib.load()
    t0 = timer.begin() if timer is not None else None
    if layout == "q4":
        fn = lib.dmvs_warp_corr_q4_f16 if f16 else lib.dmvs_warp_corr_q4
        _lib.check(fn(_ptr(ref), arr, nsrc, _ptr(proj12), None if affine else _ptr(depth_dhw),
                                         _ptr(depth_dhw.base) if affine else None, _ptr(depth_dhw.step) if affine else None,
                      _ptr(out), C, D, H, W, int(accumulate), int(variant), _stream()), "dmvs_warp_corr_q4")
    elif affine:
        _lib.check(lib.dmvs_warp_corr_affine(_ptr(ref), arr, nsrc, pix_stride, _ptr(proj12), _ptr(depth_dhw.base),
                                             _ptr(depth_dhw.step), _ptr(out), C, D, H, W, int(accumulate), _stream()),
                   "dmvs_warp_corr_affine")
    else:
        _lib.check(lib.dmvs_warp_corr(_ptr(ref), arr, nsrc, pix_stride, _ptr(proj12), _ptr(depth_dhw),
                                      _ptr(out), C, D, H, W, int(accumulate), _stream()), "dmvs_warp_corr")
    _log(family)
    if t0 is not None:
        # algorithmic bytes: features once, similarity volume written once, hypotheses once -- the [D,H,W] volume, or
        # only the [H,W] base plane when the affine form is used (count what is actually read)
        hyp = H * W if affine else D * H * W
        timer.end(family, t0, nsrc * D * H * W * (10.0 * C + 25),
                  (2.0 if f16 else 4.0) * (nsrc + 1) * C * H * W + 4.0 * (2 * D * H * W + hyp))
    return out


# ------------------------------------------------------------------------------------------ K2 / K3
@dataclass
class ConvLayer:
    """One packed conv layer of the regularisation net (weights already on the device)."""
    name: str
    mode: int            # CONV_S1 / CONV_S2 / DECONV_S2
    kdepth: int          # 3, or 1 for the 2D bottleneck layers of the refine net
    cin: int
    cout: int
    w_direct: Optional[torch.Tensor]    # [taps][Cin][Cout] (None: K3-only layer)
    w_mfma: Optional[torch.Tensor]      # MFMA A-fragment order, or None if the shape is not supported by K3
    scale: Optional[torch.Tensor]       # BN folded: gamma / sqrt(var + eps)
    shift: Optional[torch.Tensor]       # beta - mean * scale
    relu: bool
    w_wino: Optional[torch.Tensor] = None   # Winograd F(2x2,3x3) weights (stride-1 3x3 layers K3w is compiled for)
    w_wino_fpn: Optional[torch.Tensor] = None   # out3 only: composite filters of the fused level-3 merge (pack_wino_fpn)
    ones: dict = field(default_factory=dict)    # out3 only: device -> [constant-one buffers, the last one the largest] (see _ones_hw)
    w_c8: Optional[torch.Tensor] = None     # FeatureNet conv0.0 / conv0.1 only: K3s weights (pack_c8; Cin 3 or 8 -> 8)
    w_coarse: Optional[torch.Tensor] = None  # conv4 / conv6 (3D and 2D forms): K3r register-stationary Winograd weights (pack_coarse)
    w_split: Optional[torch.Tensor] = None   # conv1 only: bf16 terms for the split PROBE (pack_split); never used by `auto`
    w_zmarch: Optional[torch.Tensor] = None  # conv2 (16 -> 16, 3x3x3): K3z z-marching register-stationary Winograd weights (pack_zmarch)

    def out_shape(self, D, H, W):
        if self.mode in (CONV_S1, CONV2D_K1):
            return D, H, W
        if self.mode == CONV2D_K5S2:
            return D, (H + 1) // 2, (W + 1) // 2
        if self.mode == CONV_S2:
            return ((D + 1) // 2 if self.kdepth == 3 else D), (H + 1) // 2, (W + 1) // 2
        return (2 * D if self.kdepth == 3 else D), 2 * H, 2 * W


def pack_direct(w: torch.Tensor, transposed: bool) -> torch.Tensor:
    """PyTorch conv weight -> [kd][kh][kw][Cin][Cout] (2D weights are treated as kd=1)."""
    if w.dim() == 4:
        w = w.unsqueeze(2)
    if transposed:  # [Cin][Cout][k..] -> taps, Cin, Cout
        return w.permute(2, 3, 4, 0, 1).contiguous()
    return w.permute(2, 3, 4, 1, 0).contiguous()


def pack_mfma(w: torch.Tensor, cin: int, cout: int, mode: int, kdepth: int) -> Optional[torch.Tensor]:
    """Host-side packing into K3's fragment order via the library; None if K3 does not cover the shape."""
    lib = _lib.load()
    n = lib.dmvs_conv3d_mfma_weight_floats(cin, cout, mode, kdepth)
    if n <= 0:
        return None
    wc = w.detach().to("cpu", torch.float32).contiguous()
    out = torch.empty(n, dtype=torch.float32)
    _lib.check(lib.dmvs_pack_conv_weights_mfma(ctypes.c_void_p(wc.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                               cin, cout, mode, kdepth), "dmvs_pack_conv_weights_mfma")
    return out


def pack_wino(w: torch.Tensor, cin: int, cout: int, kdepth: int) -> Optional[torch.Tensor]:
    """Host-side G g G^T transform + packing for K3w (csrc/conv3d_wino.hip); None if the layer shape is not compiled."""
    lib = _lib.load()
    n = lib.dmvs_conv3d_wino_weight_floats(cin, cout, kdepth)
    if n <= 0:
        return None
    wc = w.detach().to("cpu", torch.float32).contiguous()
    out = torch.empty(n, dtype=torch.float32)
    _lib.check(lib.dmvs_pack_conv_weights_wino(ctypes.c_void_p(wc.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                               cin, cout, kdepth), "dmvs_pack_conv_weights_wino")
    return out


def pack_coarse(w: torch.Tensor, cin: int, cout: int, kdepth: int) -> Optional[torch.Tensor]:
    """Host-side G g G^T transform + packing for K3r (csrc/conv3d_coarse.hip: the coarse-level conv4 / conv6 with the
    filters held in registers); None if the layer shape is not compiled."""
    lib = _lib.load()
    n = lib.dmvs_conv3d_coarse_weight_floats(cin, cout, kdepth)
    if n <= 0:
        return None
    wc = w.detach().to("cpu", torch.float32).contiguous()
    out = torch.empty(n, dtype=torch.float32)
    _lib.check(lib.dmvs_pack_conv_weights_coarse(ctypes.c_void_p(wc.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                 cin, cout, kdepth), "dmvs_pack_conv_weights_coarse")
    return out


def pack_zmarch(w: torch.Tensor, cin: int, cout: int, kdepth: int) -> Optional[torch.Tensor]:
    """Host-side G g G^T transform + packing for K3z (csrc/conv3d_zmarch.hip: conv2 marching along z with the filters held in
    registers); None if the layer shape is not compiled."""
    lib = _lib.load()
    n = lib.dmvs_conv3d_zmarch_weight_floats(cin, cout, kdepth)
    if n <= 0:
        return None
    wc = w.detach().to("cpu", torch.float32).contiguous()
    out = torch.empty(n, dtype=torch.float32)
    _lib.check(lib.dmvs_pack_conv_weights_zmarch(ctypes.c_void_p(wc.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                 cin, cout, kdepth), "dmvs_pack_conv_weights_zmarch")
    return out


def pack_split(w: torch.Tensor) -> Optional[torch.Tensor]:
    """conv1 weight [16, 8, 3, 3, 3] -> three bf16 terms in MFMA B-operand order for the bf16-split PROBE
    (csrc/conv3d_split.hip; never part of the product's dispatch); None for any other shape."""
    if tuple(w.shape) != (16, 8, 3, 3, 3):
        return None
    lib = _lib.load()
    wc = w.detach().to("cpu", torch.float32).contiguous()
    out = torch.empty(lib.dmvs_conv3d_split_weight_floats(8, 16), dtype=torch.float32)
    _lib.check(lib.dmvs_pack_conv_weights_split(ctypes.c_void_p(wc.data_ptr()), ctypes.c_void_p(out.data_ptr()), 8, 16),
               "dmvs_pack_conv_weights_split")
    return out


# PROBE switch, never on in the product: 3 or 6 = run every conv1 (8 -> 16, stride 2) of the regularisation nets
# through the bf16-split kernel with that many term products; bench.py reports such a run only as `value_split`
split_probe = 0


def conv3d_split(x: torch.Tensor, layer: "ConvLayer", terms: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [8,D,H,W] -> relu(bn(conv3d stride 2)) [16,(D+1)/2,(H+1)/2,(W+1)/2] through the bf16-split probe kernel."""
    _req(x, out)
    if layer.w_split is None or layer.mode != CONV_S2 or layer.kdepth != 3:
        raise _lib.DmvsError(f"layer {layer.name}: the bf16-split probe covers conv1 (8 -> 16, stride 2, 3x3x3) only")
    Cin, D, H, W = x.shape
    Do, Ho, Wo = layer.out_shape(D, H, W)
    if out is None:
        out = torch.empty((16, Do, Ho, Wo), dtype=torch.float32, device=x.device)
    t0 = timer.begin() if timer is not None else None
    _lib.check(_lib.load().dmvs_conv3d_split_probe(_ptr(x), _ptr(out), _ptr(layer.w_split), _ptr(layer.scale), _ptr(layer.shift),
                                                   D, H, W, int(terms), RELU if layer.relu else 0, _stream()), "dmvs_conv3d_split_probe")
    _log("conv3d_mfma")
    if t0 is not None:
        fl = 2.0 * 27 * 8 * 16 * Do * Ho * Wo
        timer.end("conv3d_mfma", t0, fl, 4.0 * (8 * D * H * W + 16 * Do * Ho * Wo), fl, label=layer.name)
    return out


def pack_c8(w: torch.Tensor) -> Optional[torch.Tensor]:
    """nn.Conv2d weight [8, Cin, 3, 3] -> K3s operand order (csrc/conv2d_c8.hip); None for any other shape."""
    if w.dim() != 4 or w.shape[0] != 8 or tuple(w.shape[2:]) != (3, 3):
        return None
    lib = _lib.load()
    n = lib.dmvs_conv2d_c8_weight_floats(int(w.shape[1]))
    if n <= 0:
        return None
    wc = w.detach().to("cpu", torch.float32).contiguous()
    out = torch.empty(n, dtype=torch.float32)
    _lib.check(lib.dmvs_pack_conv_weights_c8(ctypes.c_void_p(wc.data_ptr()), ctypes.c_void_p(out.data_ptr()), int(w.shape[1])),
               "dmvs_pack_conv_weights_c8")
    return out


# FeatureNet conv0.0 -> conv0.1 as ONE kernel (dmvs_featurenet_conv0); False = two K3s launches (A/B, parity tests)
use_c8_fused = True


def featurenet_conv0(imgs: torch.Tensor, l0: "ConvLayer", l1: "ConvLayer", family: Optional[str] = None) -> Optional[torch.Tensor]:
    """imgs [V,3,H,W] -> conv0.1(conv0.0(imgs)) [8,V,H,W] (module.py:283-286) in one row sweep, the intermediate in registers.
    None when the layers do not carry K3s weights + folded BN + ReLU, or beyond the kernel's offset range (the caller then
    runs the layers one by one)."""
    _req(imgs)
    if not (use_c8 and use_c8_fused) or l0.w_c8 is None or l1.w_c8 is None or not (l0.relu and l1.relu):
        return None
    ts = (l0.w_c8, l0.scale, l0.shift, l1.w_c8, l1.scale, l1.shift)
    if any(t is None for t in ts):
        return None
    for t in ts:
        if t.device != imgs.device:
            raise _lib.DmvsError(f"layer {l0.name}: weights on {t.device}, images on {imgs.device}")
    V, c3, H, W = imgs.shape
    assert c3 == 3, imgs.shape
    out = torch.empty((8, V, H, W), dtype=torch.float32, device=imgs.device)
    t0 = timer.begin() if timer is not None else None
    code = _lib.load().dmvs_featurenet_conv0(_ptr(imgs), _ptr(out), *(_ptr(t) for t in ts), V, H, W, _stream())
    if code == _lib.EUNSUPPORTED:
        if t0 is not None:
            timer._pool.append(t0)
        return None
    _lib.check(code, "featurenet_conv0")
    fam = family or "conv3d_mfma"
    _log(fam)
    if t0 is not None:
        timer.end(fam, t0, 2.0 * 9 * (3 + 8) * 8 * V * H * W, 4.0 * (3 + 8) * V * H * W, label="feature.conv0.fused")
    return out


def pack_wino_fpn(w3: torch.Tensor, w_lat: torch.Tensor, b_lat: torch.Tensor) -> Optional[torch.Tensor]:
    """Composite Winograd filters of FeatureNet's level-3 merge (inner2 folded into out3, module.py:333-336) for
    dmvs_conv3d_wino_fpn2; w3 [16,32,3,3], w_lat [32,8], b_lat [32].  None for any other shape."""
    if tuple(w3.shape) != (16, 32, 3, 3) or tuple(w_lat.shape) != (32, 8) or tuple(b_lat.shape) != (32,):
        return None
    lib = _lib.load()
    c = [t.detach().to("cpu", torch.float32).contiguous() for t in (w3, w_lat, b_lat)]
    out = torch.empty(lib.dmvs_conv3d_wino_fpn_weight_floats(), dtype=torch.float32)
    _lib.check(lib.dmvs_pack_conv_weights_wino_fpn(*(ctypes.c_void_p(t.data_ptr()) for t in c), ctypes.c_void_p(out.data_ptr())),
               "dmvs_pack_conv_weights_wino_fpn")
    return out


def _ones_hw(layer: "ConvLayer", H: int, W: int, device) -> torch.Tensor:
    """The constant-one image the folded bias term of out3 convolves (dmvs_conv3d_wino_fpn2); the kernel only needs H * W
    contiguous ones, so ONE buffer per device serves every image size by prefix.  It is owned by the LAYER and only ever
    replaced by a larger one, the old ones staying referenced: the kernel gets a raw pointer, and a captured HIP graph
    (MVSNet.use_graph) keeps replaying with it -- growth is monotone, so an eval over many image sizes holds at most
    ~2x the largest image, not one tensor per size."""
    key = str(device)
    cur = layer.ones.get(key)
    if cur is None or cur[-1].numel() < H * W:
        n = H * W if cur is None else max(H * W, 2 * cur[-1].numel())
        layer.ones.setdefault(key, []).append(torch.ones(n, dtype=torch.float32, device=device))
        cur = layer.ones[key]
    return cur[-1]


# K3s (the 4x4x1-MFMA row sweep of FeatureNet's 8-channel full-resolution layers) wherever a layer carries w_c8; False
# leaves them to the direct-form K3 kernel (A/B, parity tests)
use_c8 = True
# K3w (Winograd form of the stride-1 3x3 layers) is used wherever a layer carries w_wino and the call has no residual /
# quad-planar output; False forces the direct-form K3 kernel everywhere (A/B, parity tests).
use_wino = True
# ... for `auto` only on volumes of at least this many workgroups.  0 = always: the 1/8-scale layers of a few dozen workgroups
# are 10-20 % slower in K3w than in the direct form (together ~0.02 ms per depth map), but a kernel choice that depends on
# the volume size would make the view-group / row-slab / view-shard modes differ from the plain forward in the last bits
WINO_MIN_BLOCKS = 0
# K3r (register-stationary Winograd, persistent) wherever a layer carries w_coarse (conv4 / conv6 of the regularisation nets) and
# the call has no residual; False leaves them to K3w / K3 (A/B, parity tests).  Like K3w the choice does not depend on the volume.
use_coarse = True
# K3z (z-marching register-stationary Winograd) wherever a layer carries w_zmarch (conv2 of the regularisation nets), no residual,
# planar output, W % 4 == 0; False leaves it to K3w / K3 (A/B, parity tests).  The choice does not depend on the volume.
use_zmarch = True
# ... for `auto` only on volumes at least this deep: a z segment of zs planes costs zs + 2 pipeline stages (two halo planes), so at
# D = 4 / 2 (stage 3, the refine passes) half of the patch transforms are halo work and K3w's two-plane tiles are as fast
# (profiles/r06_h_conv2_layers.txt: 0.157 vs 0.153 ms at D = 4; 0.084 vs 0.099 at D = 32, 0.144 vs 0.170 at D = 16).  A rule on D
# only: view groups, row slabs and view shards (which cut H, never D) pick the same kernel as the plain forward.
ZMARCH_MIN_DEPTH = 8
# ... and for FeatureNet's stride-1 3x3 layers of a shape K3r compiles (conv2.1 / conv2.2: 32 -> 32 on the [C][V][H][W] stack), read at
# pack time (MVSNet.prepare).  Off: measured SLOWER than K3w there (0.097 vs 0.082 ms per layer at config 2,
# profiles/r06_a_layers_quick_experiments.txt -- 9250 units of 48 MFMAs per wave between barriers, one whole-CU workgroup)
use_coarse_feature = False


def conv3d(x: torch.Tensor, layer: ConvLayer, skip: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None, backend: str = "auto", skip_up2: bool = False,
           family: Optional[str] = None, out_q4: bool = False, in_views: bool = False) -> torch.Tensor:
    """x [Cin,D,H,W] -> [Cout,Do,Ho,Wo];  y = relu(conv(x)*scale+shift) (+ skip).
    ``out_q4``: the result is written as two quad-planar halves, returned as [2,Do,Cout/8,Ho,Wo,4] (K3 only).
    ``in_views``: x is the loader's image stack [V,3,H,W], read in place by a 4-channel layer (DMVS_IN_VIEWS)."""
    _req(x, skip, out)
    if in_views:
        D, c3, H, W = x.shape
        if c3 != 3 or layer.cin != 4 or layer.w_mfma is None or skip is not None or backend == "direct":
            raise _lib.DmvsError(f"layer {layer.name}: in_views is the K3 form of FeatureNet's first layer ([V,3,H,W] input)")
        Cin = 4
    else:
        Cin, D, H, W = x.shape
    assert Cin == layer.cin, (layer.name, Cin, layer.cin)
    Do, Ho, Wo = layer.out_shape(D, H, W)
    oshape = (2, Do, layer.cout // 8, Ho, Wo, 4) if out_q4 else (layer.cout, Do, Ho, Wo)
    if out is None:
        out = torch.empty(oshape, dtype=torch.float32, device=x.device)
    else:
        assert tuple(out.shape) == oshape
    if out_q4 and (skip is not None or layer.w_mfma is None or backend == "direct"):
        raise _lib.DmvsError(f"layer {layer.name}: quad-planar output is a K3 epilogue without residual")
    if skip is not None:
        want = (layer.cout, Do, Ho // 2, Wo // 2) if skip_up2 else tuple(out.shape)
        assert tuple(skip.shape) == want, (tuple(skip.shape), want)
    if backend in ("split3", "split6") or (split_probe and backend == "auto" and layer.w_split is not None and skip is None
                                           and not out_q4 and not in_views):
        return conv3d_split(x, layer, 3 if backend == "split3" else 6 if backend == "split6" else split_probe, out=out)
    use_mfma = layer.w_mfma is not None and (backend in ("auto", "mfma") or layer.w_direct is None)
    if backend == "mfma" and layer.w_mfma is None:
        raise _lib.DmvsError(f"layer {layer.name}: shape not covered by the MFMA kernel")
    lib = _lib.load()
    if backend == "c8" and layer.w_c8 is None:
        raise _lib.DmvsError(f"layer {layer.name}: not one of the 8-channel layers of the row-sweep kernel")
    if layer.w_c8 is not None and skip is None and not out_q4 and (backend == "c8" or (backend == "auto" and use_c8)):
        for t in (layer.w_c8, layer.scale, layer.shift):
            if t is not None and t.device != x.device:
                raise _lib.DmvsError(f"layer {layer.name}: weights on {t.device}, activations on {x.device}")
        cin = 3 if in_views else layer.cin
        t0 = timer.begin() if timer is not None else None
        code = lib.dmvs_conv2d_c8(_ptr(x), _ptr(out), _ptr(layer.w_c8), _ptr(layer.scale), _ptr(layer.shift), cin, D, H, W,
                                  (RELU if layer.relu else 0) | (IN_VIEWS if in_views else 0), _stream())
        if code == 0:
            fam = family or "conv3d_mfma"
            _log(fam)
            if t0 is not None:
                timer.end(fam, t0, 2.0 * 9 * cin * 8 * D * H * W, 4.0 * (cin + 8) * D * H * W, label=layer.name)
            return out
        if code != _lib.EUNSUPPORTED or backend == "c8":
            _lib.check(code, f"conv3d[{layer.name}, c8]")
        if t0 is not None:
            timer._pool.append(t0)   # beyond the kernel's offset range: the K3 kernel below runs instead
    if backend == "zmarch" and (layer.w_zmarch is None or skip is not None or out_q4 or in_views):
        raise _lib.DmvsError(f"layer {layer.name}: shape / residual / layout not covered by the z-marching kernel")
    if layer.w_zmarch is not None and skip is None and not out_q4 and not in_views and (
            backend == "zmarch" or (backend == "auto" and use_zmarch and use_wino and D >= ZMARCH_MIN_DEPTH)):
        for t in (layer.w_zmarch, layer.scale, layer.shift):
            if t is not None and t.device != x.device:
                raise _lib.DmvsError(f"layer {layer.name}: weights on {t.device}, activations on {x.device}")
        t0 = timer.begin() if timer is not None else None
        code = lib.dmvs_conv3d_zmarch(_ptr(x), _ptr(out), _ptr(layer.w_zmarch), _ptr(layer.scale), _ptr(layer.shift),
                                      layer.cin, layer.cout, D, H, W, layer.kdepth, RELU if layer.relu else 0, _stream())
        if code == 0:
            fam = family or "conv3d_mfma"
            _log(fam)
            if t0 is not None:
                fl = 2.0 * 9 * layer.kdepth * layer.cin * layer.cout * D * H * W
                timer.end(fam, t0, fl, 4.0 * (layer.cin + layer.cout) * D * H * W, fl / 2.25, label=layer.name)
            return out
        if code != _lib.EUNSUPPORTED or backend == "zmarch":
            _lib.check(code, f"conv3d[{layer.name}, zmarch]")
        if t0 is not None:
            timer._pool.append(t0)   # W % 4 != 0 / unaligned / too large: K3w or K3 below
    if backend == "coarse" and (layer.w_coarse is None or skip is not None or out_q4 or in_views):
        raise _lib.DmvsError(f"layer {layer.name}: shape / residual / layout not covered by the register-stationary kernel")
    if layer.w_coarse is not None and skip is None and not out_q4 and not in_views and (
            backend == "coarse" or (backend == "auto" and use_coarse and use_wino)):
        for t in (layer.w_coarse, layer.scale, layer.shift):
            if t is not None and t.device != x.device:
                raise _lib.DmvsError(f"layer {layer.name}: weights on {t.device}, activations on {x.device}")
        t0 = timer.begin() if timer is not None else None
        code = lib.dmvs_conv3d_coarse(_ptr(x), _ptr(out), _ptr(layer.w_coarse), _ptr(layer.scale), _ptr(layer.shift),
                                      layer.cin, layer.cout, D, H, W, layer.kdepth, RELU if layer.relu else 0, _stream())
        if code == 0:
            fam = family or "conv3d_mfma"
            _log(fam)
            if t0 is not None:
                fl = 2.0 * 9 * layer.kdepth * layer.cin * layer.cout * D * H * W
                timer.end(fam, t0, fl, 4.0 * (layer.cin + layer.cout) * D * H * W, fl / 2.25, label=layer.name)
            return out
        if code != _lib.EUNSUPPORTED or backend == "coarse":
            _lib.check(code, f"conv3d[{layer.name}, coarse]")
        if t0 is not None:
            timer._pool.append(t0)
    if backend == "wino" and (layer.w_wino is None or in_views):
        raise _lib.DmvsError(f"layer {layer.name}: shape / input layout not covered by the Winograd kernel")
    # (in_views: K3w reads planar [C][D][H][W] only -- the image stack would be misread)
    if layer.w_wino is not None and skip is None and not in_views and (backend == "wino" or (
            backend == "auto" and use_wino
            and lib.dmvs_conv3d_wino_plan(layer.cin, layer.cout, D, H, W, layer.kdepth) >= WINO_MIN_BLOCKS)):
        if layer.w_wino.device != x.device:
            raise _lib.DmvsError(f"layer {layer.name}: weights on {layer.w_wino.device}, activations on {x.device}")
        t0 = timer.begin() if timer is not None else None
        code = lib.dmvs_conv3d_wino(_ptr(x), _ptr(out), _ptr(layer.w_wino), _ptr(layer.scale), _ptr(layer.shift),
                                    layer.cin, layer.cout, D, H, W, layer.kdepth,
                                    (RELU if layer.relu else 0) | (OUT_Q4 if out_q4 else 0), _stream())
        if code == 0:
            fam = family or "conv3d_mfma"
            _log(fam)
            if t0 is not None:   # FLOPs counted in the direct form (what the layer computes), as for every K3 launch
                fl = 2.0 * 9 * layer.kdepth * layer.cin * layer.cout * D * H * W
                # executed: 16 of 36 products; conv0 (Cin = 2) pads its 6 (channel, depth tap) pairs to two k-groups of 4
                timer.end(fam, t0, fl, 4.0 * (layer.cin + layer.cout) * D * H * W, fl / 2.25 * (8.0 / 6.0 if layer.cin == 2 else 1.0),
                          label=layer.name)
            return out
        if code != _lib.EUNSUPPORTED or backend == "wino":
            _lib.check(code, f"conv3d[{layer.name}, wino]")
        if t0 is not None:
            timer._pool.append(t0)   # shape / alignment not covered: the direct-form kernel below runs instead
    fn = lib.dmvs_conv3d_mfma if use_mfma else lib.dmvs_conv3d_direct
    w = layer.w_mfma if use_mfma else layer.w_direct
    for t in (w, layer.scale, layer.shift):   # raw pointers go to the kernel: a weight left on the CPU / another GPU faults
        if t is not None and t.device != x.device:
            raise _lib.DmvsError(f"layer {layer.name}: weights on {t.device}, activations on {x.device}")
    t0 = timer.begin() if timer is not None else None
    code = fn(_ptr(x), _ptr(out), _ptr(w), _ptr(layer.scale), _ptr(layer.shift), _ptr(skip), layer.cin, layer.cout,
              D, H, W, layer.mode, layer.kdepth,
              (RELU if layer.relu else 0) | (SKIP_UP2 if skip_up2 else 0) | (OUT_Q4 if out_q4 else 0)
              | (IN_VIEWS if in_views else 0), _stream())
    _lib.check(code, f"conv3d[{layer.name}, {'mfma' if use_mfma else 'direct'}]")
    fam = family or ("conv3d_mfma" if use_mfma else ("prob_head" if layer.cout == 2 else "conv3d_direct"))
    _log(fam)
    if t0 is not None:
        taps = 25 if layer.mode == CONV2D_K5S2 else (1 if layer.mode == CONV2D_K1 else 9 * layer.kdepth)
        vox = D * H * W if layer.mode == DECONV_S2 else Do * Ho * Wo   # deconv: MACs counted on the input grid
        nbytes = 4.0 * (layer.cin * D * H * W + layer.cout * Do * Ho * Wo * (2 if skip is not None else 1))
        timer.end(fam, t0, 2.0 * taps * layer.cin * layer.cout * vox, nbytes, label=layer.name)
    return out


def conv3d_fpn(lat: torch.Tensor, td: torch.Tensor, layer: ConvLayer, out_q4: bool = False,
               family: Optional[str] = None) -> Optional[torch.Tensor]:
    """out = conv3x3(b_lat + w_lat . lat + up2(td)) as ONE Winograd convolution (FeatureNet's inner2 + upsample-add + out3,
    the lateral conv folded into ``layer.w_wino_fpn`` on the host: pack_wino_fpn).  lat [8,V,H,W], td [32,V,H/2,W/2].
    Returns None when the shape is not covered (the caller then runs the layers separately)."""
    _req(lat, td)
    Cl, V, H, W = lat.shape
    Cin = td.shape[0]
    assert tuple(td.shape) == (Cin, V, H // 2, W // 2) and layer.cin == Cin
    if not use_wino or layer.w_wino_fpn is None or (Cl, Cin, layer.cout) != (8, 32, 16):
        return None
    for t in (layer.w_wino_fpn, layer.scale, layer.shift):
        if t is not None and t.device != lat.device:
            raise _lib.DmvsError(f"layer {layer.name}: weights on {t.device}, activations on {lat.device}")
    oshape = (2, V, layer.cout // 8, H, W, 4) if out_q4 else (layer.cout, V, H, W)
    out = torch.empty(oshape, dtype=torch.float32, device=lat.device)
    t0 = timer.begin() if timer is not None else None
    code = _lib.load().dmvs_conv3d_wino_fpn2(_ptr(lat), _ptr(td), _ptr(_ones_hw(layer, H, W, lat.device)), _ptr(out),
                                             _ptr(layer.w_wino_fpn), _ptr(layer.scale), _ptr(layer.shift), V, H, W,
                                             (RELU if layer.relu else 0) | (OUT_Q4 if out_q4 else 0), _stream())
    if code == _lib.EUNSUPPORTED:
        if t0 is not None:
            timer._pool.append(t0)
        return None
    _lib.check(code, f"conv3d_fpn[{layer.name}]")
    _log(family or "conv3d_mfma")
    if t0 is not None:
        vox = V * H * W
        fl = 2.0 * vox * (9 * Cin * layer.cout + Cl * Cin)
        # executed: (3 x 16 + 8 x 9) MFMAs of 2048 FLOP per 64 pixels
        timer.end(family or "conv3d_mfma", t0, fl, 4.0 * (Cl * vox + Cin * vox / 4 + layer.cout * vox), vox * 120 * 2048 / 64.0,
                  label="feature.out3.fpn")
    return out


# ------------------------------------------------------------------------------------------ K4
# `prob` -> K4 for the passes one workgroup can hold (D = 4 / 8; r06): the `prob` head of a branch regresses its own
# two channels (prob_regress) and K4 shrinks to the selection on [4,H,W] (depth_select).  bench.py --no-prob-fused = the two kernels.
use_prob_fused = True
PROB_FUSED_DEPTHS = (4, 8)


def prob_fusable(D: int, W: int, backend: str) -> bool:
    return use_prob_fused and backend in ("auto", "mfma") and D in PROB_FUSED_DEPTHS and W % 4 == 0


def prob_regress(x: torch.Tensor, layer: ConvLayer, depth_dhw, interval: torch.Tensor, alpha: float, out: torch.Tensor) -> bool:
    """x [Cin,D,H,W] (conv11's output) -> out [2,H,W]: softmax over D of alpha * prob(x) and the depth expectation, the branch's two
    channels of depth_sub_plus (module.py:379,397 + mvsnet.py:19-20,68-69).  ``depth_dhw`` as for depth_regress.  Returns False when
    the shape is not covered (the caller then runs conv3d + depth_regress)."""
    affine = isinstance(depth_dhw, AffinePlanes)
    _req(x, depth_dhw.base if affine else depth_dhw, interval, out)
    if affine and depth_dhw.step.data_ptr() != interval.data_ptr():
        raise _lib.DmvsError("prob_regress: AffinePlanes.step is not the `interval` tensor -- the planes regressed on would differ "
                             "from the planes K1 correlated")
    Cin, D, H, W = x.shape
    assert layer.cout == 2 and Cin == layer.cin and tuple(out.shape) == (2, H, W) and tuple(depth_dhw.shape) == (D, H, W)
    if layer.w_direct.device != x.device:
        raise _lib.DmvsError(f"layer {layer.name}: weights on {layer.w_direct.device}, activations on {x.device}")
    t0 = timer.begin() if timer is not None else None
    code = _lib.load().dmvs_prob_regress(_ptr(x), _ptr(layer.w_direct), Cin, D, H, W, None if affine else _ptr(depth_dhw),
                                         _ptr(depth_dhw.base) if affine else None, _ptr(interval) if affine else None,
                                         float(alpha), _ptr(out), _stream())
    if code == _lib.EUNSUPPORTED:
        if t0 is not None:
            timer._pool.append(t0)
        return False
    _lib.check(code, f"prob_regress[{layer.name}]")
    _log("prob_head")
    if t0 is not None:
        hyp = H * W if affine else D * H * W
        timer.end("prob_head", t0, 2.0 * 27 * Cin * 2 * D * H * W, 4.0 * (Cin * D * H * W + hyp + 2 * H * W), label=layer.name)
    return True


def depth_select(dsp: torch.Tensor, interval: torch.Tensor, mode: int):
    """dsp [4,H,W] (two prob_regress calls) -> (sel ([4,H,W] | [H,W]), conf [H,W]): K4's part behind the expectations."""
    _req(dsp, interval)
    _, H, W = dsp.shape
    sel = torch.empty((4, H, W) if mode == 0 else (H, W), dtype=torch.float32, device=dsp.device)
    conf = torch.empty((H, W), dtype=torch.float32, device=dsp.device)
    t0 = timer.begin() if timer is not None else None
    _lib.check(_lib.load().dmvs_depth_select(_ptr(dsp), _ptr(interval), mode, H, W, _ptr(sel), _ptr(conf), _stream()), "dmvs_depth_select")
    _log("depth_regress")
    if t0 is not None:
        timer.end("depth_regress", t0, 0.0, 4.0 * (4 + (5 if mode == 0 else 2)) * H * W)
    return sel, conf


def depth_regress(logits: torch.Tensor, depth_dhw: torch.Tensor, interval: torch.Tensor, alpha: float, mode: int,
                  want_prob: bool):
    """logits [4,D,H,W], depth [D,H,W] (or AffinePlanes) -> (dsp [4,H,W], sel ([4,H,W] | [H,W]), conf [H,W], prob | None)."""
    affine = isinstance(depth_dhw, AffinePlanes)
    _req(logits, depth_dhw.base if affine else depth_dhw, interval)
    if affine and depth_dhw.step.data_ptr() != interval.data_ptr():
        # the kernel forms the planes as base + d * interval[0]: K1 correlated on base + d * step, they must be one scalar
        raise _lib.DmvsError("depth_regress: AffinePlanes.step is not the `interval` tensor handed to K4 -- the planes "
                             "regressed on would differ from the planes K1 correlated")
    _, D, H, W = logits.shape
    assert tuple(depth_dhw.shape) == (D, H, W)
    dev = logits.device
    dsp = torch.empty((4, H, W), dtype=torch.float32, device=dev)
    sel = torch.empty((4, H, W) if mode == 0 else (H, W), dtype=torch.float32, device=dev)
    conf = torch.empty((H, W), dtype=torch.float32, device=dev)
    prob = torch.empty_like(logits) if want_prob else None
    t0 = timer.begin() if timer is not None else None
    fn = _lib.load().dmvs_depth_regress_affine if affine else _lib.load().dmvs_depth_regress
    _lib.check(fn(_ptr(logits), _ptr(depth_dhw.base if affine else depth_dhw), _ptr(interval), float(alpha), mode, D, H, W,
                  _ptr(dsp), _ptr(sel), _ptr(conf), _ptr(prob), _stream()), "dmvs_depth_regress")
    _log("depth_regress")
    if t0 is not None:
        # algorithmic bytes: the 4 logit planes per hypothesis once + the hypotheses ([D,H,W], or only the [H,W] base plane of the
        # affine form -- VERDICT r05: count what is read) + the outputs (dsp 4, sel 4 | 1, conf 1 planes) (+ the softmax volume)
        hyp = H * W if affine else D * H * W
        timer.end("depth_regress", t0, 0.0, 4.0 * (4 * D * H * W + hyp + (9 if mode == 0 else 6) * H * W + (4 * D * H * W if want_prob else 0)))
    return dsp, sel, conf, prob
