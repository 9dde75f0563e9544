"""GPU (-m gpu): parity of the HIP path, called through the C ABI, against the oracle and the golden
vectors generated from the reference.  Tolerances: the north-star bound is 1e-3 relative L1 on the final
depth; per-op bounds here are much tighter (fp32 re-association level)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from dmvsnet_amd import MVSNet, ops, synth  # noqa: E402
from dmvsnet_amd._lib import DmvsError  # noqa: E402
from oracle import dmvs_oracle as O  # noqa: E402

DEV = "cuda:0"
T = torch.from_numpy


def cu(a):
    a = T(a) if isinstance(a, np.ndarray) else a
    return a.to(DEV).contiguous()


def assert_close(got, want, atol, rtol=0.0, what=""):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    want = want.detach().cpu().numpy() if torch.is_tensor(want) else want
    np.testing.assert_allclose(got, want, atol=atol, rtol=rtol, err_msg=what)


def rnd(*shape, seed=0, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return T(g.standard_normal(shape, dtype=np.float32) * np.float32(scale))


# ------------------------------------------------------------------------------------------ small kernels
def test_nchw_to_hwc():
    for C, H, W in ((8, 5, 37), (16, 33, 20), (32, 17, 300)):
        x = rnd(1, 2 * C, H, W, seed=C)
        for c0 in (0, C):
            got = ops.nchw_to_hwc(cu(x), c0, C)
            assert torch.equal(got.cpu(), x[0, c0:c0 + C].permute(1, 2, 0).contiguous())


def test_relative_proj():
    cams = synth.synth_cameras(1184, 1600, 5)
    for s in (1, 2, 3):
        P = cams[f"stage{s}"]
        got = ops.relative_proj(cu(P[0])).cpu()
        for v in range(1, 5):
            rot, tr = O.relative_projection(P[:, v].double(), P[:, 0].double())
            want = torch.cat((rot[0].reshape(-1), tr[0].reshape(-1))).float()
            # elements formed by cancellation carry ~1e-6 relative noise in either implementation
            assert_close(got[v - 1], want, atol=1e-5, rtol=1e-5, what=f"stage{s} view{v}")
            # what matters: projected pixel positions (far corner, near and far plane) agree to 1e-3 px
            sc = 2 ** (3 - s)
            x = torch.tensor([1600 / sc - 1, 1184 / sc - 1, 1.0])
            for d in (425.0, 935.0):
                pa = (got[v - 1][:9].view(3, 3).double() @ x.double()) * d + got[v - 1][9:].double()
                pb = (rot[0] @ x.double()) * d + tr[0]
                assert ((pa[:2] / pa[2]) - (pb[:2] / pb[2])).abs().max() < 1e-3


@pytest.mark.parametrize("inv", [0, 1])
def test_hypotheses(golden, inv):
    g = golden("op_hypotheses.npz")
    dv = synth.synth_depth_values()
    s, i = ops.hypotheses_first(cu(dv), 8, 6, 8, bool(inv))
    assert_close(s, g[f"first_inv{inv}"][0], atol=2e-4)
    assert_close(i, g[f"first_inv{inv}_itv"], atol=1e-5)
    ratio = 2.0  # the fixture used pix = 2 * depth_interval
    s, i = ops.hypotheses_next(cu(g["last"][0]), cu(dv), ratio, 8, bool(inv))
    assert_close(s, g[f"later_inv{inv}_up"][0], atol=3e-4)
    assert_close(i, g[f"later_inv{inv}_itv"], atol=1e-5)


# ------------------------------------------------------------------------------------------ K1
class _K1:
    """One K1 kernel configuration: feature layout + launch variant."""

    def __init__(self, layout, variant):
        self.layout, self.variant = layout, variant

    def feat(self, f):  # [1,C,H,W] (CPU) -> device tensor in the kernel's layout
        hwc = cu(f[0].permute(1, 2, 0).contiguous())
        return ops.hwc_to_q4(hwc) if self.layout == "q4" else hwc

    def hwc(self, f_hwc):  # [H,W,C] on the device -> the kernel's layout
        return ops.hwc_to_q4(f_hwc) if self.layout == "q4" else f_hwc

    def __call__(self, ref, src, p12, depth, **kw):
        return ops.warp_corr(ref, src, p12, depth, variant=self.variant, layout=self.layout, **kw)


@pytest.fixture(params=[("q4", 0), ("q4", 8), ("q4", 2), ("q4", 3), ("hwc", 0)],
                ids=["q4", "q4_dc4", "q4_win53", "q4_win80", "hwc_generic"])
def k1(request):
    """Every K1 parity test runs against the product kernel (quad-planar features, LDS windows staged in channel
    slabs) in its launch configurations -- 8 / 4 planes per workgroup, 40 / 53 / 80 KB windows -- and against the
    generic pixel-major kernel behind dmvs_warp_corr (taps through the vector L1)."""
    return _K1(*request.param)


def _hwc(f):  # [1,C,H,W] -> device [H,W,C]
    return cu(f[0].permute(1, 2, 0).contiguous())


def _q4_halves_to_planar(o):
    """[2, V, Cq, H, W, 4] (the DMVS_OUT_Q4 epilogue: two quad-planar halves per view) -> [2 * Cq * 4, V, H, W]."""
    halves = [h.permute(1, 4, 0, 2, 3).reshape(-1, h.shape[0], h.shape[2], h.shape[3]) for h in o]
    return torch.cat(halves, 0)


def test_warp_corr_golden(golden, k1):
    g = golden("op_costagg.npz")
    feats = [k1.feat(T(g[f"feat{v}"])) for v in range(3)]
    p12 = ops.relative_proj(cu(g["proj"][0]))
    sim = k1(feats[0], feats[1:], p12, cu(g["depth"][0]))
    assert_close(sim, g["sim"][0], atol=1e-5)
    # accumulate=1: two single-view launches add up to the same volume (view-shard contract)
    part = k1(feats[0], feats[1:2], p12[:1].contiguous(), cu(g["depth"][0]))
    k1(feats[0], feats[2:], p12[1:].contiguous(), cu(g["depth"][0]), out=part, accumulate=True)
    assert_close(part, g["sim"][0], atol=1e-5)


def test_homo_warping_out_of_bounds(golden, k1):
    """z<0 and out-of-image planes: group correlation of the golden warped volume with a one-hot reference."""
    g = golden("op_homo_warping.npz")
    src = T(g["src"])
    C, H, W = src.shape[1:]
    proj = (T(g["src_proj"]) @ torch.inverse(T(g["ref_proj"])))[0]
    p12 = torch.cat((proj[:3, :3].reshape(-1), proj[:3, 3])).view(1, 12)
    ref = torch.ones(1, C, H, W)
    sim = k1(k1.feat(ref), [k1.feat(src)], cu(p12), cu(g["depth"][0]))
    want = T(g["warped"])[0].view(C // 2, 2, -1, H, W).mean(0)
    assert_close(sim, want, atol=1e-5)


def _smooth(x, k=5):
    return F.avg_pool2d(x, k, 1, k // 2)


@pytest.mark.parametrize("smooth", [True, False])
@pytest.mark.parametrize("C,D,H,W,V", [(32, 5, 19, 70, 3), (16, 9, 40, 100, 2), (8, 4, 64, 130, 4), (16, 11, 30, 67, 3)])
def test_warp_corr_vs_oracle(C, D, H, W, V, smooth, k1):
    """Ragged sizes (W not a multiple of the pixel tile, D not a multiple of the depth chunk).
    Tap positions agree with ATen's to ~1e-4 px (fp32 coordinate rounding at |coord| ~ 100); the value error is
    that times the feature gradient, so white-noise features (gradient ~1 per px) get the looser bound."""
    feats = [rnd(1, C, H, W, seed=10 + v) for v in range(V)]
    if smooth:
        feats = [_smooth(f) * 3 for f in feats]
    cams = synth.synth_cameras(H * 4, W * 4, V)["stage1"]
    depth = 450.0 + 60.0 * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1) + rnd(1, D, H, W, seed=3, scale=4.0)
    want = O.warp_corr(feats, cams, depth)
    sim = k1(k1.feat(feats[0]), [k1.feat(f) for f in feats[1:]], ops.relative_proj(cu(cams[0])), cu(depth[0]))
    assert_close(sim, want[0], atol=3e-5 if smooth else 5e-4)
    assert (sim.cpu() - want[0]).abs().mean() < (3e-6 if smooth else 3e-5)
    assert want.abs().mean() > 1e-3


# ------------------------------------------------------------------------------------------ K2 / K3
def _layer(w, mode, kd, bn=True, seed=0):
    tr = mode == ops.DECONV_S2
    cin, cout = (w.shape[0], w.shape[1]) if tr else (w.shape[1], w.shape[0])
    g = np.random.Generator(np.random.PCG64(seed))
    scale = T((0.5 + g.random(cout)).astype(np.float32)) if bn else None
    shift = T((0.2 * g.standard_normal(cout)).astype(np.float32)) if bn else None
    wm = ops.pack_mfma(w, cin, cout, mode, kd)
    return ops.ConvLayer("t", mode, kd, cin, cout, cu(ops.pack_direct(w, tr)), None if wm is None else cu(wm),
                         None if scale is None else cu(scale), None if shift is None else cu(shift), bn), scale, shift


def _conv_ref(x, w, mode, kd, scale, shift, skip):
    x5 = x[None]
    if kd == 1:
        w5 = w.unsqueeze(2) if w.dim() == 4 else w
    else:
        w5 = w
    if mode == ops.DECONV_S2:
        if kd == 3:
            y = F.conv_transpose3d(x5, w5, None, 2, 1, 1)
        else:
            y = F.conv_transpose3d(x5, w5, None, (1, 2, 2), (0, 1, 1), (0, 1, 1))
    else:
        s = 1 if mode == ops.CONV_S1 else 2
        y = F.conv3d(x5, w5, None, (s if kd == 3 else 1, s, s), (kd // 2, 1, 1))
    y = y[0]
    if scale is not None:
        y = torch.relu(y * scale.view(-1, 1, 1, 1) + shift.view(-1, 1, 1, 1))
    if skip is not None:
        y = y + skip
    return y


CONV_CASES = [
    # (cin, cout, mode, kd, D, H, W)
    (2, 16, ops.CONV_S1, 3, 8, 16, 40), (8, 16, ops.CONV_S2, 3, 8, 16, 40), (16, 16, ops.CONV_S1, 3, 4, 9, 21),
    (16, 32, ops.CONV_S2, 3, 4, 8, 24), (32, 32, ops.CONV_S1, 3, 2, 8, 16), (32, 64, ops.CONV_S2, 3, 2, 8, 16),
    (64, 64, ops.CONV_S1, 3, 1, 5, 7), (64, 32, ops.DECONV_S2, 3, 1, 5, 7), (32, 16, ops.DECONV_S2, 3, 2, 6, 9),
    (16, 8, ops.DECONV_S2, 3, 4, 8, 20), (8, 2, ops.CONV_S1, 3, 8, 16, 40), (8, 16, ops.CONV_S2, 3, 4, 16, 24),
    (16, 32, ops.CONV_S2, 3, 2, 8, 12),   # refine net: D 4 -> 2 -> 1
    (32, 64, ops.CONV_S2, 1, 1, 8, 12), (64, 64, ops.CONV_S1, 1, 1, 4, 6), (64, 32, ops.DECONV_S2, 1, 1, 4, 6),
    (8, 16, ops.CONV_S2, 3, 5, 37, 51),   # odd sizes round up
]


@pytest.mark.parametrize("backend", ["direct", "mfma"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv3d(case, backend):
    cin, cout, mode, kd, D, H, W = case
    tr = mode == ops.DECONV_S2
    shape = ((cin, cout) if tr else (cout, cin)) + ((kd, 3, 3) if kd == 3 else (3, 3))
    w = rnd(*shape, seed=cin * 100 + cout, scale=1.0 / np.sqrt(cin * 9 * kd))
    bn = cout != 2
    layer, scale, shift = _layer(w, mode, kd, bn=bn, seed=cin + cout)
    if backend == "mfma" and layer.w_mfma is None:
        pytest.skip("shape not covered by the MFMA kernel")
    x = rnd(cin, D, H, W, seed=1)
    Do, Ho, Wo = layer.out_shape(D, H, W)
    for use_skip in (False, True):
        skip = rnd(cout, Do, Ho, Wo, seed=2) if use_skip else None
        want = _conv_ref(x, w, mode, kd, scale, shift, skip)
        got = ops.conv3d(cu(x), layer, skip=None if skip is None else cu(skip), backend=backend)
        assert tuple(got.shape) == tuple(want.shape)
        assert_close(got, want, atol=2e-5, what=f"{case} skip={use_skip}")


WINO_CASES = [
    # (cin, cout, kd, D, H, W): every layer shape K3w is compiled for; ragged H / D (partial tile rows and planes),
    # W not a multiple of the 32-column tile, several channel chunks, one- and two-stage LDS pipelines
    (16, 16, 3, 5, 13, 44), (16, 16, 3, 1, 21, 40), (32, 32, 3, 3, 11, 36), (64, 64, 3, 2, 7, 20), (64, 64, 1, 2, 9, 24),
    (16, 16, 1, 3, 19, 72), (32, 32, 1, 2, 10, 100), (16, 16, 3, 8, 74, 100), (32, 16, 1, 2, 21, 48),
    (2, 16, 3, 5, 19, 44), (2, 16, 3, 4, 74, 100), (2, 16, 3, 1, 9, 36),
]


@pytest.mark.parametrize("D,H,W", [(2, 6, 40), (1, 9, 33), (3, 20, 64), (16, 74, 100)])
def test_deconv_residual_prefetch_is_bit_identical(D, H, W):
    """conv11 (transposed 16 -> 8 + residual) prefetches its residual under the MFMAs (counted vmcnt pipeline); the
    arithmetic is the same, so the result must equal the epilogue-load form BIT FOR BIT (dmvs_tune A/B), and ATen at 2e-5."""
    from dmvsnet_amd import _lib
    lib = _lib.load()
    w = rnd(16, 8, 3, 3, 3, seed=3, scale=0.1)
    layer, scale, shift = _layer(w, ops.DECONV_S2, 3)
    x, skip = rnd(16, D, H, W, seed=4), rnd(8, 2 * D, 2 * H, 2 * W, seed=5)
    try:
        _lib.check(lib.dmvs_tune(b"k3_deconv_prefetch", 0), "tune")
        base = ops.conv3d(cu(x), layer, skip=cu(skip), backend="mfma").clone()
        _lib.check(lib.dmvs_tune(b"k3_deconv_prefetch", 1), "tune")
        pref = ops.conv3d(cu(x), layer, skip=cu(skip), backend="mfma")
    finally:
        lib.dmvs_tune(b"k3_deconv_prefetch", 1)
    assert torch.equal(base, pref)
    assert_close(pref, _conv_ref(x, w, ops.DECONV_S2, 3, scale, shift, skip), atol=2e-5)


@pytest.mark.parametrize("case", WINO_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv3d_wino(case):
    """K3w (Winograd F(2x2,3x3) on the fp32 MFMA) against ATen's direct fp32 convolution -- same tolerance as the
    direct-form kernels' test -- and against K3 itself (re-association level)."""
    cin, cout, kd, D, H, W = case
    w = rnd(*((cout, cin) + ((3, 3, 3) if kd == 3 else (3, 3))), seed=cin * 100 + cout + kd, scale=1.0 / np.sqrt(cin * 9 * kd))
    layer, scale, shift = _layer(w, ops.CONV_S1, kd, bn=True, seed=cin + cout)
    ww = ops.pack_wino(w, cin, cout, kd)
    assert ww is not None
    layer.w_wino = cu(ww)
    x = rnd(cin, D, H, W, seed=1)
    want = _conv_ref(x, w, ops.CONV_S1, kd, scale, shift, None)
    got = ops.conv3d(cu(x), layer, backend="wino")
    assert_close(got, want, atol=2e-5, what=f"{case}")
    direct = ops.conv3d(cu(x), layer, backend="mfma")
    assert (got - direct).abs().max().item() < 1e-5
    if cin > 2:
        q4 = ops.conv3d(cu(x), layer, backend="wino", out_q4=True)   # quad-planar halves: the swapped-operand epilogue
        assert_close(_q4_halves_to_planar(q4), want, atol=2e-5, what=f"{case} q4")
    assert want.abs().mean() > 0.05


@pytest.mark.parametrize("cfg", [(16, 16, 3), (32, 32, 3), (64, 64, 3), (64, 64, 1), (16, 16, 1), (32, 32, 1), (32, 16, 1), (2, 16, 3)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv3d_wino_random_shapes(cfg):
    """Seeded random volumes (W % 4 == 0, everything else ragged: one-row / one-plane volumes, widths below one tile,
    several persistent tiles per workgroup on the big ones) -- K3w against the direct-form K3 kernel on the same input."""
    cin, cout, kd = cfg
    g = np.random.Generator(np.random.PCG64(cin * 7 + cout + kd))
    w = rnd(*((cout, cin) + ((3, 3, 3) if kd == 3 else (3, 3))), seed=cin + cout + kd, scale=1.0 / np.sqrt(cin * 9 * kd))
    layer, scale, shift = _layer(w, ops.CONV_S1, kd, bn=True, seed=3)
    layer.w_wino = cu(ops.pack_wino(w, cin, cout, kd))
    shapes = [(1, 1, 4), (1, 2, 8), (3, 1, 36), (2, 3, 4)] + \
        [(int(g.integers(1, 7)), int(g.integers(1, 70)), 4 * int(g.integers(1, 40))) for _ in range(6)] + [(3, 150, 260)]
    for D, H, W in shapes:
        x = cu(rnd(cin, D, H, W, seed=D * 1000 + H * 10 + W))
        a = ops.conv3d(x, layer, backend="wino")
        b = ops.conv3d(x, layer, backend="mfma")
        assert torch.isfinite(a).all()
        assert (a - b).abs().max().item() < 2e-5, (cfg, D, H, W)   # 64 x 27-term sums: a few 1e-6 of re-association


ZMARCH_CASES = [
    # (D, H, W): conv2 (16 -> 16, 3x3x3) -- the config-2 shapes at reduced size + the edges: one plane, fewer planes than a segment,
    # D not a multiple of the segment, volumes smaller than one 8 x 8 group, ragged right / bottom groups, more columns than
    # persistent workgroup slots (several columns per workgroup: the ring wraps, the finish crosses column boundaries)
    (16, 74, 100), (4, 148, 200), (32, 37, 52), (2, 74, 100), (1, 21, 40), (5, 13, 44), (3, 9, 8), (1, 1, 4), (7, 30, 36),
    (9, 64, 136), (2, 296, 400), (1, 8, 8), (17, 10, 12),
]


@pytest.mark.parametrize("case", ZMARCH_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv3d_zmarch(case):
    """K3z (csrc/conv3d_zmarch.hip: conv2 in Winograd F(2x2,3x3) form, filters in registers, marching along z) against ATen's direct
    fp32 convolution at the direct kernels' tolerance, against K3 itself, bit-identical run to run, every output written."""
    D, H, W = case
    w = rnd(16, 16, 3, 3, 3, seed=161 + D, scale=1.0 / np.sqrt(16 * 27))
    layer, scale, shift = _layer(w, ops.CONV_S1, 3, bn=True, seed=32)
    wz = ops.pack_zmarch(w, 16, 16, 3)
    assert wz is not None
    layer.w_zmarch = cu(wz)
    x = rnd(16, D, H, W, seed=1)
    want = _conv_ref(x, w, ops.CONV_S1, 3, scale, shift, None)
    out = torch.full((16, D, H, W), float("nan"), device=DEV)   # every output must be written
    got = ops.conv3d(cu(x), layer, backend="zmarch", out=out)
    assert_close(got, want, atol=2e-5, what=f"{case}")
    direct = ops.conv3d(cu(x), layer, backend="mfma")
    assert (got - direct).abs().max().item() < 2e-5   # 16 x 27-term sums: re-association level
    assert torch.equal(got, ops.conv3d(cu(x), layer, backend="zmarch"))   # run to run: same bits
    if H * W > 64:
        assert want.abs().mean() > 0.05


def test_conv3d_zmarch_segment_and_grid_do_not_change_the_bits():
    """A column's z segment length and the persistent grid only change WHO computes an output plane and when: per output the
    products are accumulated in the same order (depth tap 0, 1, 2 x channel groups), so every choice gives the same bits -- what
    lets the launcher pick the segment per volume (view groups / row slabs / view shards then equal the plain forward)."""
    from dmvsnet_amd import _lib
    lib = _lib.load()
    w = rnd(16, 16, 3, 3, 3, seed=77, scale=0.05)
    layer, scale, shift = _layer(w, ops.CONV_S1, 3, bn=True, seed=3)
    layer.w_zmarch = cu(ops.pack_zmarch(w, 16, 16, 3))
    for D, H, W in ((12, 40, 52), (5, 70, 96)):
        x = cu(rnd(16, D, H, W, seed=D))
        base = ops.conv3d(x, layer, backend="zmarch").clone()
        try:
            for zs in (1, 2, 3, 4, 8, 16):
                _lib.check(lib.dmvs_tune(b"k3z_zs", zs), "tune")
                assert torch.equal(base, ops.conv3d(x, layer, backend="zmarch")), (D, zs)
            _lib.check(lib.dmvs_tune(b"k3z_zs", 0), "tune")
            for grid in (8, 64, 2048):
                _lib.check(lib.dmvs_tune(b"k3z_grid", grid), "tune")
                assert torch.equal(base, ops.conv3d(x, layer, backend="zmarch")), (D, grid)
        finally:
            lib.dmvs_tune(b"k3z_zs", 0)
            lib.dmvs_tune(b"k3z_grid", 0)
    assert lib.dmvs_tune(b"k3z_grid", 12) != 0 and lib.dmvs_tune(b"k3z_zs", 65) != 0


def test_conv3d_zmarch_random_shapes_and_dispatch():
    """Seeded random volumes (W % 4 == 0, everything else ragged) against the direct-form K3 kernel; `auto` takes K3z for a layer
    that carries its weights (no residual, planar output), K3w / K3 when W % 4 != 0, with a residual or with ops.use_zmarch off;
    an explicit `zmarch` never falls back silently; raw sums without BatchNorm / ReLU come through."""
    g = np.random.Generator(np.random.PCG64(99))
    w = rnd(16, 16, 3, 3, 3, seed=5, scale=0.05)
    layer, scale, shift = _layer(w, ops.CONV_S1, 3, bn=True, seed=4)
    layer.w_zmarch = cu(ops.pack_zmarch(w, 16, 16, 3))
    layer.w_wino = cu(ops.pack_wino(w, 16, 16, 3))
    shapes = [(1, 1, 4), (1, 2, 8), (3, 1, 36), (2, 3, 4)] + \
        [(int(g.integers(1, 20)), int(g.integers(1, 90)), 4 * int(g.integers(1, 40))) for _ in range(8)] + [(3, 150, 260)]
    for D, H, W in shapes:
        x = cu(rnd(16, D, H, W, seed=D * 1000 + H * 10 + W))
        a = ops.conv3d(x, layer, out=torch.full((16, D, H, W), float("nan"), device=DEV))   # auto -> K3z from D = 8 up, K3w below
        assert torch.equal(a, ops.conv3d(x, layer, backend="zmarch" if D >= ops.ZMARCH_MIN_DEPTH else "wino")), (D, H, W)
        a = ops.conv3d(x, layer, backend="zmarch")
        b = ops.conv3d(x, layer, backend="mfma")
        assert torch.isfinite(a).all() and (a - b).abs().max().item() < 2e-5, (D, H, W)
    x = rnd(16, 3, 10, 18, seed=8)   # W % 4 != 0: not covered -> auto falls back (K3w needs W % 4 == 0 too: K3), explicit raises
    assert_close(ops.conv3d(cu(x), layer), _conv_ref(x, w, ops.CONV_S1, 3, scale, shift, None), atol=2e-5)
    with pytest.raises(DmvsError):
        ops.conv3d(cu(x), layer, backend="zmarch")
    x = rnd(16, 3, 10, 20, seed=9)
    skip = rnd(16, 3, 10, 20, seed=10)
    assert_close(ops.conv3d(cu(x), layer, skip=cu(skip)), _conv_ref(x, w, ops.CONV_S1, 3, scale, shift, skip), atol=2e-5)
    with pytest.raises(DmvsError):
        ops.conv3d(cu(x), layer, skip=cu(skip), backend="zmarch")
    keep, ops.use_zmarch = ops.use_zmarch, False
    try:
        assert torch.equal(ops.conv3d(cu(x), layer), ops.conv3d(cu(x), layer, backend="wino"))
    finally:
        ops.use_zmarch = keep
    plain, _, _ = _layer(w, ops.CONV_S1, 3, bn=False)
    plain.w_zmarch = layer.w_zmarch
    assert_close(ops.conv3d(cu(x), plain, backend="zmarch"), _conv_ref(x, w, ops.CONV_S1, 3, None, None, None), atol=2e-5)
    assert ops.pack_zmarch(rnd(32, 32, 3, 3, 3), 32, 32, 3) is None and ops.pack_zmarch(rnd(16, 16, 3, 3), 16, 16, 1) is None


@pytest.mark.parametrize("D,H,W", [(8, 16, 40), (5, 37, 51), (4, 9, 33), (16, 74, 100), (1, 1, 1), (3, 8, 64)])
def test_conv3d_split_probe(D, H, W):
    """The bf16-split PROBE (csrc/conv3d_split.hip; not in the product path): conv1 (8 -> 16, stride 2) with the
    fp32 operands split into three exact bf16 terms.  The SIX-term form (hh + hm + mh + hl + mm + lh, fp32 accumulation) is held to
    the fp32 kernels' own bound -- 2e-5 of the output scale against ATen's fp32 convolution -- and must be as close to a float64
    convolution as the fp32 MFMA kernel is (within 2x); the THREE-term form stops at 2^-16 per product and is only bounded loosely."""
    w = rnd(16, 8, 3, 3, 3, seed=11 + D, scale=1.0 / np.sqrt(8 * 27))
    layer, scale, shift = _layer(w, ops.CONV_S2, 3, bn=True, seed=7)
    layer.w_split = cu(ops.pack_split(w))
    x = rnd(8, D, H, W, seed=2)
    want = _conv_ref(x, w, ops.CONV_S2, 3, scale, shift, None)
    y64 = F.conv3d(x[None].double(), w.double(), None, 2, 1)[0]
    truth = torch.relu(y64 * scale.double().view(-1, 1, 1, 1) + shift.double().view(-1, 1, 1, 1))
    out = torch.full(tuple(want.shape), float("nan"), device=DEV)
    got6 = ops.conv3d(cu(x), layer, backend="split6", out=out)
    assert_close(got6, want, atol=2e-5, what=f"six-term {D, H, W}")
    got3 = ops.conv3d(cu(x), layer, backend="split3")
    assert_close(got3, want, atol=2e-3, what=f"three-term {D, H, W}")
    fp32 = ops.conv3d(cu(x), layer, backend="mfma")
    e6 = (got6.double().cpu() - truth).abs().max().item()
    e32 = (fp32.double().cpu() - truth).abs().max().item()
    e3 = (got3.double().cpu() - truth).abs().max().item()
    assert e6 <= 2.0 * e32 + 1e-7, (e6, e32)          # emulated fp32: as close to the exact result as the fp32 kernel
    assert e3 >= e6                                     # and the three-term form is measurably coarser (or equal on tiny volumes)
    assert torch.equal(got6, ops.conv3d(cu(x), layer, backend="split6"))
    # never part of `auto`
    assert torch.equal(ops.conv3d(cu(x), layer), fp32)


COARSE_CASES = [
    # (cin, cout, kd, D, H, W): the config-2 coarse shapes at reduced size + the edge cases: W % 4 != 0 (dword tile loads, scalar
    # stores: the 37 x 50 volumes of stage 1), volumes smaller than one 8 x 8 group, one group, ragged right / bottom groups,
    # more units than workgroup slots (several units per workgroup: the ring wraps, the merged finish runs), depth 1 / 2 / 8
    (64, 64, 3, 8, 37, 50), (64, 64, 3, 4, 74, 100), (64, 64, 3, 1, 9, 12), (64, 64, 3, 2, 5, 7), (64, 64, 3, 3, 64, 136),
    (32, 32, 3, 16, 74, 100), (32, 32, 3, 2, 30, 52), (32, 32, 3, 1, 8, 8), (32, 32, 3, 5, 17, 23), (32, 32, 3, 8, 148, 200),
    (64, 64, 1, 1, 37, 50), (64, 64, 1, 1, 74, 100), (64, 64, 1, 3, 20, 36), (64, 64, 1, 1, 148, 200), (64, 64, 1, 1, 3, 5),
    (32, 32, 1, 1, 74, 100), (32, 32, 1, 1, 37, 50), (32, 32, 1, 2, 296, 400), (32, 32, 1, 1, 1, 1),
]


@pytest.mark.parametrize("case", COARSE_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv3d_coarse(case):
    """K3r (csrc/conv3d_coarse.hip: conv4 / conv6 and their 2D forms, Winograd F(2x2,3x3) with register-stationary filters,
    persistent workgroups) against ATen's direct fp32 convolution at the direct kernels' tolerance, and against K3 itself."""
    cin, cout, kd, D, H, W = case
    w = rnd(*((cout, cin) + ((3, 3, 3) if kd == 3 else (3, 3))), seed=cin * 100 + cout + kd, scale=1.0 / np.sqrt(cin * 9 * kd))
    layer, scale, shift = _layer(w, ops.CONV_S1, kd, bn=True, seed=cin + cout)
    wr = ops.pack_coarse(w, cin, cout, kd)
    assert wr is not None
    layer.w_coarse = cu(wr)
    x = rnd(cin, D, H, W, seed=1)
    want = _conv_ref(x, w, ops.CONV_S1, kd, scale, shift, None)
    out = torch.full((cout, D, H, W), float("nan"), device=DEV)   # every output must be written
    got = ops.conv3d(cu(x), layer, backend="coarse", out=out)
    assert_close(got, want, atol=2e-5, what=f"{case}")
    direct = ops.conv3d(cu(x), layer, backend="mfma")
    assert (got - direct).abs().max().item() < 2e-5   # 64 x 27-term sums: a few 1e-6 of re-association (as K3w vs K3)
    assert torch.equal(got, ops.conv3d(cu(x), layer, backend="coarse"))   # run to run: same bits
    if H * W > 64:
        assert want.abs().mean() > 0.05


@pytest.mark.parametrize("cfg", [(32, 32, 3), (64, 64, 3), (32, 32, 1), (64, 64, 1)], ids=lambda c: "x".join(map(str, c)))
def test_conv3d_coarse_random_shapes(cfg):
    """Seeded random volumes -- any W (both tile loaders), ragged everything, from single voxels to several units per persistent
    workgroup -- K3r against the direct-form K3 kernel on the same input (a supplement: each is separately checked against ATen)."""
    cin, cout, kd = cfg
    g = np.random.Generator(np.random.PCG64(cin * 3 + cout + kd))
    w = rnd(*((cout, cin) + ((3, 3, 3) if kd == 3 else (3, 3))), seed=cin + cout + kd, scale=1.0 / np.sqrt(cin * 9 * kd))
    layer, scale, shift = _layer(w, ops.CONV_S1, kd, bn=True, seed=4)
    layer.w_coarse = cu(ops.pack_coarse(w, cin, cout, kd))
    shapes = [(1, 1, 1), (1, 2, 9), (3, 1, 37), (2, 9, 8), (1, 8, 9)] + \
        [(int(g.integers(1, 6)), int(g.integers(1, 80)), int(g.integers(1, 130))) for _ in range(7)] + [(3, 151, 263)]
    for D, H, W in shapes:
        x = cu(rnd(cin, D, H, W, seed=D * 1000 + H * 10 + W))
        a = ops.conv3d(x, layer, backend="coarse", out=torch.full((cout, D, H, W), float("nan"), device=DEV))
        b = ops.conv3d(x, layer, backend="mfma")
        assert torch.isfinite(a).all(), (cfg, D, H, W)
        assert (a - b).abs().max().item() < 2e-5, (cfg, D, H, W)


@pytest.mark.parametrize("case", [(32, 32, 3, 8, 148, 200), (64, 64, 3, 4, 74, 100), (64, 64, 1, 1, 148, 200), (32, 32, 1, 2, 296, 400)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv3d_coarse_counted_wait_is_bit_identical(case):
    """K3r's ring wait is a COUNTED `s_waitcnt vmcnt(NS)` (the newest NS loads may stay in flight); the conservative `vmcnt(0)`
    form (dmvs_tune("k3r_counted_wait", 0)) reads the same tiles, so the two must agree BIT FOR BIT -- the dynamic half of the gate
    whose static half is tests/test_static_isa.py.  Shapes with many units per persistent workgroup: the ring wraps."""
    from dmvsnet_amd import _lib
    lib = _lib.load()
    cin, cout, kd, D, H, W = case
    w = rnd(*((cout, cin) + ((3, 3, 3) if kd == 3 else (3, 3))), seed=17 + cin + kd, scale=1.0 / np.sqrt(cin * 9 * kd))
    layer, scale, shift = _layer(w, ops.CONV_S1, kd, bn=True, seed=9)
    layer.w_coarse = cu(ops.pack_coarse(w, cin, cout, kd))
    x = cu(rnd(cin, D, H, W, seed=2))
    try:
        _lib.check(lib.dmvs_tune(b"k3r_counted_wait", 0), "tune")
        base = ops.conv3d(x, layer, backend="coarse").clone()
        _lib.check(lib.dmvs_tune(b"k3r_counted_wait", 1), "tune")
        for _ in range(3):
            assert torch.equal(base, ops.conv3d(x, layer, backend="coarse"))
    finally:
        lib.dmvs_tune(b"k3r_counted_wait", 1)


@pytest.mark.parametrize("grid", [32, 1024])
def test_conv3d_coarse_grid_knob(grid):
    """dmvs_tune("k3r_grid"): 32 persistent workgroups (conv6: one slot per XCD and cout group, ~100 units each) and 1024 (more than
    one workgroup per CU asked for) against ATen and against the default grid, bit for bit."""
    from dmvsnet_amd import _lib
    lib = _lib.load()
    for cin, kd, D, H, W in ((64, 3, 3, 40, 72), (32, 3, 4, 50, 60), (64, 1, 2, 33, 52)):
        w = rnd(*((cin, cin) + ((3, 3, 3) if kd == 3 else (3, 3))), seed=23 + cin + kd, scale=1.0 / np.sqrt(cin * 9 * kd))
        layer, scale, shift = _layer(w, ops.CONV_S1, kd, bn=True, seed=5)
        layer.w_coarse = cu(ops.pack_coarse(w, cin, cin, kd))
        x = rnd(cin, D, H, W, seed=6)
        base = ops.conv3d(cu(x), layer, backend="coarse").clone()
        try:
            _lib.check(lib.dmvs_tune(b"k3r_grid", grid), "tune")
            got = ops.conv3d(cu(x), layer, backend="coarse", out=torch.full((cin, D, H, W), float("nan"), device=DEV))
        finally:
            lib.dmvs_tune(b"k3r_grid", 256)
        assert torch.equal(got, base)
        assert_close(got, _conv_ref(x, w, ops.CONV_S1, kd, scale, shift, None), atol=2e-5)
    assert lib.dmvs_tune(b"k3r_grid", 48) != 0 and lib.dmvs_tune(b"k3r_grid", 2048) != 0   # not a multiple of 32 / out of range


def test_conv3d_coarse_dispatch():
    """`auto` takes K3r for the layers that carry its weights (no residual, planar output), K3w / K3 otherwise; without BatchNorm
    and ReLU the raw sums come through; shapes it is not compiled for have no K3r weights."""
    w = rnd(32, 32, 3, 3, 3, seed=5, scale=0.06)
    layer, scale, shift = _layer(w, ops.CONV_S1, 3, bn=True, seed=2)
    layer.w_wino = cu(ops.pack_wino(w, 32, 32, 3))
    layer.w_coarse = cu(ops.pack_coarse(w, 32, 32, 3))
    x = rnd(32, 4, 20, 28, seed=3)
    a = ops.conv3d(cu(x), layer)
    assert torch.equal(a, ops.conv3d(cu(x), layer, backend="coarse"))
    keep, ops.use_coarse = ops.use_coarse, False
    try:
        assert torch.equal(ops.conv3d(cu(x), layer), ops.conv3d(cu(x), layer, backend="wino"))
    finally:
        ops.use_coarse = keep
    skip = rnd(32, 4, 20, 28, seed=4)
    assert_close(ops.conv3d(cu(x), layer, skip=cu(skip)), _conv_ref(x, w, ops.CONV_S1, 3, scale, shift, skip), atol=2e-5)
    plain, _, _ = _layer(w, ops.CONV_S1, 3, bn=False)
    plain.w_coarse = layer.w_coarse
    assert_close(ops.conv3d(cu(x), plain, backend="coarse"), _conv_ref(x, w, ops.CONV_S1, 3, None, None, None), atol=2e-5)
    with pytest.raises(DmvsError):   # an explicit `coarse` never falls back silently
        ops.conv3d(cu(x), layer, skip=cu(skip), backend="coarse")
    assert ops.pack_coarse(rnd(16, 16, 3, 3, 3), 16, 16, 3) is None
    with pytest.raises(DmvsError):
        ops.conv3d(cu(rnd(16, 2, 8, 8)), _layer(rnd(16, 16, 3, 3, 3, scale=0.1), ops.CONV_S1, 3)[0], backend="coarse")


@pytest.mark.parametrize("V,H,W", [(1, 8, 32), (3, 20, 68), (5, 33, 70), (2, 64, 128)])
def test_first_feature_layer_reads_the_image_stack_in_place(V, H, W):
    """DMVS_IN_VIEWS: FeatureNet's conv0.0 (RGB + one zero-weight channel) on the loader's [V,3,H,W] stack must equal the same
    layer on the planar [4,V,H,W] copy with a zero channel BIT FOR BIT (the 4th channel reads the next view's red plane, or
    past the buffer for the last view: zero weights, zero contribution); both tile loaders (W % 4 == 0 or not)."""
    w = rnd(8, 3, 3, 3, seed=31, scale=0.2)
    w4 = torch.cat((w, torch.zeros_like(w[:, :1])), 1).contiguous()
    layer, _, _ = _layer(w4, ops.CONV_S1, 1)
    imgs = torch.rand(V, 3, H, W, generator=torch.Generator().manual_seed(V)) * 1000.0   # large values: nothing may leak
    planar = torch.cat((imgs.permute(1, 0, 2, 3), torch.zeros(1, V, H, W)), 0).contiguous()
    want = ops.conv3d(cu(planar), layer, backend="mfma")
    got = ops.conv3d(cu(imgs), layer, backend="mfma", in_views=True)
    assert torch.equal(got, want)
    assert_close(got, _conv_ref(planar, w4, ops.CONV_S1, 1, layer.scale.cpu(), layer.shift.cpu(), None), atol=5e-2, rtol=2e-5)


C8_SHAPES = [(1, 1, 5), (1, 8, 32), (2, 13, 61), (3, 20, 62), (2, 37, 63), (1, 9, 124), (5, 33, 125), (2, 70, 200)]


@pytest.mark.parametrize("V,H,W", C8_SHAPES)
@pytest.mark.parametrize("cin", [3, 8])
def test_conv2d_c8_rowsweep(cin, V, H, W):
    """K3s (csrc/conv2d_c8.hip): FeatureNet's conv0.0 / conv0.1 (module.py:283-286) as a register-only row sweep on the
    4x4x1 MFMA vs ATen on the CPU (fp32), and vs the direct-form K3 kernel.  Shapes around the 62-pixel strip width (61, 62,
    63, 124, 125: the neighbour-lane taps across strip edges), rows that are no multiple of the x4-unrolled sweep, one-row and
    one-column images, several views (no view may see its neighbour's rows through the zero padding)."""
    w = rnd(8, cin, 3, 3, seed=40 + cin, scale=0.25)
    wk = torch.cat((w, torch.zeros_like(w[:, :1])), 1).contiguous() if cin == 3 else w
    layer, scale, shift = _layer(wk, ops.CONV_S1, 1, bn=True, seed=cin)
    layer.w_c8 = cu(ops.pack_c8(w))
    x = rnd(cin, V, H, W, seed=7 * V + H, scale=3.0)
    want = _conv_ref(x, w, ops.CONV_S1, 1, scale, shift, None)
    if cin == 8:
        got = ops.conv3d(cu(x), layer, backend="c8")
        k3 = ops.conv3d(cu(x), layer, backend="mfma")
    else:   # planar 3-channel input through the raw entry point; the product reads the loader's image stack in place
        import ctypes
        from dmvsnet_amd import _lib
        P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        xc, got = cu(x), torch.full((8, V, H, W), float("nan"), device=DEV)
        _lib.check(_lib.load().dmvs_conv2d_c8(P(xc), P(got), P(layer.w_c8), P(layer.scale), P(layer.shift), 3, V, H, W,
                                              ops.RELU, None), "dmvs_conv2d_c8")
        stack = cu(x.permute(1, 0, 2, 3).contiguous())   # [V,3,H,W]
        assert torch.equal(ops.conv3d(stack, layer, backend="c8", in_views=True), got)
        k3 = ops.conv3d(stack, layer, backend="mfma", in_views=True)
    assert_close(got, want, atol=2e-5, rtol=2e-5)
    assert_close(got, k3.cpu(), atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("V,H,W", [(1, 1, 3), (1, 7, 59), (2, 13, 60), (3, 22, 61), (2, 35, 121), (1, 9, 180), (2, 70, 200)])
def test_featurenet_conv0_fused(V, H, W):
    """dmvs_featurenet_conv0: conv0.0 -> conv0.1 (module.py:283-286) in one sweep, the intermediate in registers, must equal
    the two K3s launches BIT FOR BIT (same operations in the same order; the intermediate's zero padding is a mask, not
    conv0.0 evaluated outside the image) and ATen within 2e-5.  Widths around the 60-pixel strip."""
    w0, w1 = rnd(8, 3, 3, 3, seed=50, scale=0.3), rnd(8, 8, 3, 3, seed=51, scale=0.2)
    l0, s0, b0 = _layer(torch.cat((w0, torch.zeros_like(w0[:, :1])), 1).contiguous(), ops.CONV_S1, 1, bn=True, seed=3)
    l1, s1, b1 = _layer(w1, ops.CONV_S1, 1, bn=True, seed=4)
    l0.w_c8, l1.w_c8 = cu(ops.pack_c8(w0)), cu(ops.pack_c8(w1))
    imgs = rnd(V, 3, H, W, seed=V + H + W, scale=2.0)
    got = ops.featurenet_conv0(cu(imgs), l0, l1)
    assert got is not None
    two = ops.conv3d(ops.conv3d(cu(imgs), l0, backend="c8", in_views=True), l1, backend="c8")
    assert torch.equal(got, two)
    x = imgs.permute(1, 0, 2, 3).contiguous()
    want = _conv_ref(_conv_ref(x, w0, ops.CONV_S1, 1, s0, b0, None), w1, ops.CONV_S1, 1, s1, b1, None)
    assert_close(got, want, atol=3e-5, rtol=3e-5)
    ops.use_c8_fused = False
    try:
        assert ops.featurenet_conv0(cu(imgs), l0, l1) is None
    finally:
        ops.use_c8_fused = True


def test_conv2d_c8_no_bn_and_dispatch():
    """No BN / no ReLU (identity epilogue), the `auto` dispatch (w_c8 present -> K3s; ops.use_c8 = False or a residual -> K3),
    argument checks of the entry point."""
    import ctypes
    from dmvsnet_amd import _lib
    w = rnd(8, 8, 3, 3, seed=3, scale=0.2)
    layer, _, _ = _layer(w, ops.CONV_S1, 1, bn=False)
    layer.w_c8 = cu(ops.pack_c8(w))
    x = rnd(8, 2, 19, 77, seed=5)
    want = _conv_ref(x, w, ops.CONV_S1, 1, None, None, None)
    ops.launch_log = log = []
    try:
        got = ops.conv3d(cu(x), layer)
        ops.use_c8 = False
        got_k3 = ops.conv3d(cu(x), layer)
    finally:
        ops.use_c8 = True
        ops.launch_log = None
    assert_close(got, want, atol=2e-5, rtol=2e-5)
    assert_close(got_k3, want, atol=2e-5, rtol=2e-5)
    assert len(log) == 2
    lib = _lib.load()
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    xc, out = cu(x), torch.empty((8, 2, 19, 77), device=DEV)
    assert lib.dmvs_conv2d_c8(P(xc), P(out), P(layer.w_c8), None, None, 4, 2, 19, 77, 0, None) == _lib.EUNSUPPORTED      # Cin
    assert lib.dmvs_conv2d_c8(P(xc), P(out), P(layer.w_c8), None, None, 8, 2, 19, 77, ops.IN_VIEWS, None) == _lib.EUNSUPPORTED
    assert lib.dmvs_conv2d_c8(P(xc), P(out), P(layer.w_c8), None, None, 8, 2, 19, 77, ops.OUT_Q4, None) == _lib.EUNSUPPORTED
    assert lib.dmvs_conv2d_c8(None, P(out), P(layer.w_c8), None, None, 8, 2, 19, 77, 0, None) == _lib.EINVAL
    assert ops.pack_c8(rnd(16, 8, 3, 3, seed=1)) is None and ops.pack_c8(rnd(8, 4, 3, 3, seed=1)) is None
    # beyond 2^28 output elements the byte offsets would reach the out-of-range markers: refused before any launch (the host
    # then runs the K3 kernel); the pointers are never dereferenced
    assert lib.dmvs_conv2d_c8(P(xc), P(out), P(layer.w_c8), None, None, 8, 64, 2048, 2048, 0, None) == _lib.EUNSUPPORTED
    assert lib.dmvs_featurenet_conv0(P(xc), P(out), P(layer.w_c8), P(xc), P(xc), P(layer.w_c8), P(xc), P(xc), 64, 2048, 2048,
                                     None) == _lib.EUNSUPPORTED
    assert lib.dmvs_featurenet_conv0(P(xc), P(out), P(layer.w_c8), None, None, P(layer.w_c8), P(xc), P(xc), 2, 19, 77, None) == _lib.EINVAL


def test_retired_flag_bit_is_rejected():
    """ABI 110: flag value 4 meant DMVS_OUT_HWC2 (pixel-major halves) in version 100; DMVS_OUT_Q4 is 8 now and a
    caller that still passes 4 gets DMVS_EUNSUPPORTED from every conv entry point instead of another output layout."""
    import ctypes
    from dmvsnet_amd import _lib
    lib = _lib.load()
    w = rnd(16, 16, 3, 3, seed=1, scale=0.1)
    layer, _, _ = _layer(w, ops.CONV_S1, 1)
    x, out = cu(rnd(16, 1, 8, 32, seed=2)), torch.empty((16, 1, 8, 32), device=DEV)
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    for flags in (4, 4 | ops.RELU):
        assert lib.dmvs_conv3d_mfma(P(x), P(out), P(layer.w_mfma), P(layer.scale), P(layer.shift), None, 16, 16, 1, 8, 32,
                                    ops.CONV_S1, 1, flags, None) == _lib.EUNSUPPORTED
        assert lib.dmvs_conv3d_direct(P(x), P(out), P(layer.w_direct), P(layer.scale), P(layer.shift), None, 16, 16, 1, 8, 32,
                                      ops.CONV_S1, 1, flags, None) == _lib.EUNSUPPORTED
    ww = cu(ops.pack_wino(w, 16, 16, 1))
    assert lib.dmvs_conv3d_wino(P(x), P(out), P(ww), P(layer.scale), P(layer.shift), 16, 16, 1, 8, 32, 1, 4, None) == _lib.EUNSUPPORTED
    assert ops.OUT_Q4 == 8


def test_conv3d_wino_falls_back():
    """W % 4 != 0 (no 16-byte tile loader), a residual or a quad-planar output: `auto` runs the direct-form kernel, an
    explicit `wino` raises; a layer shape K3w is not compiled for has no Winograd weights."""
    w = rnd(16, 16, 3, 3, 3, seed=3, scale=0.1)
    layer, scale, shift = _layer(w, ops.CONV_S1, 3, bn=True, seed=1)
    layer.w_wino = cu(ops.pack_wino(w, 16, 16, 3))
    x = rnd(16, 3, 9, 22, seed=1)
    want = _conv_ref(x, w, ops.CONV_S1, 3, scale, shift, None)
    assert_close(ops.conv3d(cu(x), layer), want, atol=2e-5)
    with pytest.raises(DmvsError):
        ops.conv3d(cu(x), layer, backend="wino")
    ops.WINO_MIN_BLOCKS, keep = 96, ops.WINO_MIN_BLOCKS   # the volume-size policy knob (off by default)
    try:
        small = ops.conv3d(cu(rnd(16, 3, 9, 24, seed=1)), layer)   # a handful of workgroups: `auto` keeps the direct form
        assert torch.equal(small, ops.conv3d(cu(rnd(16, 3, 9, 24, seed=1)), layer, backend="mfma"))
    finally:
        ops.WINO_MIN_BLOCKS = keep
    x = rnd(16, 3, 9, 24, seed=1)
    skip = rnd(16, 3, 9, 24, seed=2)
    assert_close(ops.conv3d(cu(x), layer, skip=cu(skip)), _conv_ref(x, w, ops.CONV_S1, 3, scale, shift, skip), atol=2e-5)
    assert ops.pack_wino(rnd(16, 8, 3, 3, 3), 8, 16, 3) is None


FEAT_CASES = [
    # (cin, cout, mode, V, H, W): the FeatureNet-only modes on a [C][V][H][W] stack; widths with W % 4 != 0 take the
    # dword tile loader and the scalar-store epilogue, the others the 16-byte paths
    (8, 16, ops.CONV2D_K5S2, 2, 18, 26), (8, 16, ops.CONV2D_K5S2, 2, 16, 40), (16, 32, ops.CONV2D_K5S2, 1, 11, 37),
    (32, 64, ops.CONV2D_K1, 2, 9, 20), (16, 32, ops.CONV2D_K1, 2, 10, 18), (8, 32, ops.CONV2D_K1, 1, 12, 36),
    (4, 8, ops.CONV_S1, 2, 9, 22), (8, 8, ops.CONV_S1, 2, 16, 40), (32, 16, ops.CONV_S1, 2, 8, 24), (32, 32, ops.CONV_S1, 1, 7, 13),
]


@pytest.mark.parametrize("case", FEAT_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv2d_feature_modes(case):
    cin, cout, mode, V, H, W = case
    k = 5 if mode == ops.CONV2D_K5S2 else 1 if mode == ops.CONV2D_K1 else 3
    stride = 2 if mode == ops.CONV2D_K5S2 else 1
    w = rnd(cout, cin, k, k, seed=cin * 7 + cout, scale=1.0 / np.sqrt(cin * k * k))
    g = np.random.Generator(np.random.PCG64(cin + cout))
    scale, shift = T((0.5 + g.random(cout)).astype(np.float32)), T((0.2 * g.standard_normal(cout)).astype(np.float32))
    wm = ops.pack_mfma(w, cin, cout, mode, 1)
    assert wm is not None
    layer = ops.ConvLayer("t", mode, 1, cin, cout, None, cu(wm), cu(scale), cu(shift), True)
    x = rnd(cin, V, H, W, seed=5)
    want = torch.relu(F.conv2d(x.permute(1, 0, 2, 3), w, None, stride, k // 2) * scale.view(1, -1, 1, 1)
                      + shift.view(1, -1, 1, 1)).permute(1, 0, 2, 3)            # [cout, V, Ho, Wo]
    got = ops.conv3d(cu(x), layer)
    assert_close(got, want, atol=2e-5, what=f"{case}")
    if cout % 8 == 0:   # quad-planar halves (DMVS_OUT_Q4)
        hw = ops.conv3d(cu(x), layer, out_q4=True)
        assert_close(_q4_halves_to_planar(hw), want, atol=2e-5, what=f"{case} q4")
    if mode == ops.CONV2D_K1 and want.shape[-1] % 2 == 0 and want.shape[-2] % 2 == 0:   # fused nearest x2 upsample-add
        sk = rnd(cout, V, want.shape[-2] // 2, want.shape[-1] // 2, seed=6)
        up = F.interpolate(sk.permute(1, 0, 2, 3), scale_factor=2, mode="nearest").permute(1, 0, 2, 3)
        got2 = ops.conv3d(cu(x), layer, skip=cu(sk), skip_up2=True)
        assert_close(got2, want + up, atol=2e-5, what=f"{case} skip_up2")


@pytest.mark.parametrize("V,H,W", [(2, 16, 40), (1, 34, 72), (3, 8, 32), (1, 2, 8), (2, 22, 104)])
def test_conv3d_fpn(V, H, W):
    """inner2 (1x1 + bias) + nearest x2 upsample-add + out3 (3x3) as ONE Winograd convolution (dmvs_conv3d_wino_fpn2) vs
    the three torch ops; ragged tile counts (H not a multiple of the tile, W not a multiple of 32)."""
    Cl, Cin, Cout = 8, 32, 16
    w_lat, b_lat = rnd(Cin, Cl, seed=1, scale=0.3), rnd(Cin, seed=2, scale=0.2)
    w3 = rnd(Cout, Cin, 3, 3, seed=3, scale=1.0 / np.sqrt(Cin * 9))
    lat, td = rnd(Cl, V, H, W, seed=4), rnd(Cin, V, H // 2, W // 2, seed=5)
    intra = F.conv2d(lat.permute(1, 0, 2, 3), w_lat[:, :, None, None], b_lat) + \
        F.interpolate(td.permute(1, 0, 2, 3), scale_factor=2, mode="nearest")
    want = F.conv2d(intra, w3, None, 1, 1).permute(1, 0, 2, 3)                      # [Cout, V, H, W]
    layer = ops.ConvLayer("t", ops.CONV_S1, 1, Cin, Cout, None, cu(ops.pack_mfma(w3, Cin, Cout, ops.CONV_S1, 1)), None, None, False)
    layer.w_wino_fpn = cu(ops.pack_wino_fpn(w3, w_lat, b_lat))   # the lateral conv + bias folded into the Winograd filters
    got = ops.conv3d_fpn(cu(lat), cu(td), layer)
    assert got is not None
    assert_close(got, want, atol=3e-5)
    hw = ops.conv3d_fpn(cu(lat), cu(td), layer, out_q4=True)
    assert_close(_q4_halves_to_planar(hw), want, atol=3e-5, what="q4")
    # a width the fused kernel does not cover is reported, not mis-computed (FeatureNet.run then runs inner2 and out3)
    assert ops.conv3d_fpn(cu(rnd(Cl, 1, 8, 36, seed=6)), cu(rnd(Cin, 1, 4, 18, seed=7)), layer) is None


def _net(ndepths, ratios, seed, inverse=False):
    net = MVSNet(ndepths, ratios, inverse_depth=inverse, verbose=False)
    sd = synth.synth_state_dict(net.state_dict(), seed)
    net.load_state_dict(sd)
    return net.to(DEV), sd


@pytest.mark.parametrize("backend", ["direct", "auto"])
def test_costreg_golden(golden, backend):
    g = golden("op_costreg.npz")
    net, _ = _net([8], [4], int(g["seed"]))
    net.prepare(torch.device(DEV))
    y = net.cost_regularization[0].run(cu(g["x"][0]), backend)
    assert_close(y, g["y_full"][0], atol=1e-4)
    yr = net.cost_regularization_refine[0].run(cu(g["xr"][0]), backend)
    assert_close(yr, g["yr_full"][0], atol=1e-4)


def test_featurenet_k3_golden(golden):
    """FeatureNet on the MFMA conv kernels (5x5 stride-2, 1x1 + fused upsample-add, 3x3) vs the reference."""
    g = golden("op_featurenet.npz")
    net, _ = _net([8], [4], int(g["seed"]))
    net.prepare(torch.device(DEV))
    img = cu(g["img"])                                   # [1,3,32,32]
    outs = net.feature.run(torch.cat((img, img * 0.5), 0))  # two "views": batching must not mix slices
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    fo = [O.feature_net(sd, x) for x in (T(g["img"]), T(g["img"]) * 0.5)]   # the oracle (ATen on the CPU), per view
    want = [torch.cat([torch.cat((f[f"stage{k}"], f[f"stage{k}_c"]), 1) for f in fo], 0) for k in (1, 2, 3)]
    for s, (o, w) in enumerate(zip(outs, want)):
        # o [2, V, C/4, h, w, 4]: the stageK / stageK_c halves, quad-planar (the DMVS_OUT_Q4 epilogue)
        assert tuple(o.shape[:2]) == (2, 2) and o.shape[-1] == 4 and o.is_contiguous()
        q2p = lambda t: t.permute(0, 3, 1, 2).reshape(-1, t.shape[1], t.shape[2])   # [C/4,h,w,4] -> [C,h,w]
        assert_close(q2p(o[0, 0]), g[f"stage{s + 1}"][0], atol=2e-5)
        assert_close(q2p(o[1, 0]), g[f"stage{s + 1}_c"][0], atol=2e-5)
        both = _q4_halves_to_planar(o).permute(1, 0, 2, 3)   # [V, 2C, h, w]
        assert_close(both, w, atol=2e-5)


def test_planar_to_hwc():
    """Layout glue for callers that keep planar feature stacks: [C][V][H][W] slice -> [H][W][C]."""
    x = torch.randn(12, 3, 9, 20, device=DEV)
    got = ops.planar_to_hwc(x, 1, 4, 8)
    assert_close(got, x[4:12, 1].permute(1, 2, 0), atol=0)


# ------------------------------------------------------------------------------------------ K4
def test_depth_regress_golden(golden):
    g = golden("op_depthnet.npz")
    itv = cu(g["interval"])
    dsp, hyps, conf, prob = ops.depth_regress(cu(g["logits"][0]), cu(g["depth_values"][0]), itv, 1.0, 0, True)
    assert_close(dsp, g["dsp"][0], atol=2e-4)
    assert_close(hyps, g["hyps"][0], atol=2e-3)
    assert_close(conf, g["conf"][0], atol=1e-5)
    assert_close(prob, g["prob"][0], atol=1e-6)
    dsp2, depth, conf2, _ = ops.depth_regress(cu(g["logits_c"][0]), cu(g["hyps"][0]), itv, 5.0, 1, False)
    assert_close(depth, g["depth"][0], atol=2e-4)
    assert_close(dsp2, g["dsp_refine"][0], atol=2e-4)
    assert_close(conf2, g["conf_refine"][0], atol=1e-5)


# ------------------------------------------------------------------------------------------ end to end
@pytest.mark.parametrize("name", ["e2e_c1.npz", "e2e_small3.npz", "e2e_small3_inv.npz"])
def test_end_to_end_golden(golden, name):
    """The boundary call against the reference's own outputs: depth rel-L1 <= 1e-3 (north star)."""
    g = golden(name)
    cfg = [int(v) for v in g["cfg"]]
    H, W, V, seed, inv = cfg[:5]
    ns = (len(cfg) - 5) // 2
    ndepths, ratios = cfg[5:5 + ns], cfg[5 + ns:]
    net, _ = _net(ndepths, ratios, seed, bool(inv))
    imgs, proj, dv = synth.synth_inputs(H, W, V, seed)
    out = net(cu(imgs), {k: cu(v) for k, v in proj.items()}, cu(dv))
    want_keys = {"depth", "depth_sub_plus", "depth_sub_plus_refine", "depth_values", "depth_values_c", "interval",
                 "photometric_confidence", "photometric_confidence_refine", "prob_volume"}
    assert want_keys | {f"stage{s + 1}" for s in range(ns)} == set(out.keys())
    net.return_prob_volume = False      # the knob INTEGRATION.md recommends for eval: the key disappears, nothing is None
    out2 = net(cu(imgs), {k: cu(v) for k, v in proj.items()}, cu(dv))
    assert "prob_volume" not in out2 and "prob_volume" not in out2["stage1"]
    assert all(v is not None for v in out2.values())
    assert torch.equal(out2["depth"], out["depth"])
    for s in range(ns):
        st = out[f"stage{s + 1}"]
        ref = g[f"stage{s + 1}.depth"]
        got = st["depth"].cpu().numpy()
        assert got.shape == ref.shape
        rel = np.abs(got - ref).mean() / np.abs(ref).mean()
        assert rel < 1e-3, (name, s, rel)
        assert rel < 2e-5, (name, s, rel)  # what fp32 re-association actually costs
        # measured r04: max abs 1.4e-6 ... 3.1e-5 over the three goldens (the confidence is a steep function of the spread of
        # the four regressed depths, so its error is ~10x the depths'); r01-r03 had 5e-3 here
        assert_close(st["photometric_confidence"], g[f"stage{s + 1}.photometric_confidence"], atol=2e-4)
        assert_close(st["photometric_confidence_refine"], g[f"stage{s + 1}.photometric_confidence_refine"], atol=2e-4)
        assert_close(st["interval"], g[f"stage{s + 1}.interval"], atol=1e-5)
        assert_close(st["depth_sub_plus"], g[f"stage{s + 1}.depth_sub_plus"], atol=0, rtol=1e-4)
    sc = 2 ** (3 - ns)  # a 1-stage net stops at quarter resolution (mvsnet.py:214)
    assert out["depth"].shape == (1, H // sc, W // sc) and out["photometric_confidence"].shape == (1, H // sc, W // sc)


def test_feature_view_groups_and_single_stream():
    """Host-side knobs must not change results: FeatureNet in view groups (large configs), one vs two streams, the
    top-down path on its own stream, the fused vs unfused level-3 merge."""
    net, _ = _net([16, 8, 8], [3, 2, 1], 1)
    imgs, proj, dv = synth.synth_inputs(64, 96, 3, 1)
    args = (cu(imgs), {k: cu(v) for k, v in proj.items()}, cu(dv))
    base = net(*args)["depth"].clone()
    net.feature_group_views = 2
    assert torch.equal(net(*args)["depth"], base)
    net.feature_group_views = None
    net.two_streams = False
    assert torch.equal(net(*args)["depth"], base)
    net.two_streams = True
    assert net.feature_async_topdown                      # default: FeatureNet's top-down path on a third stream
    net.feature_async_topdown = False
    assert torch.equal(net(*args)["depth"], base)
    # inner2 / upsample-add / out3 as three kernels: the same sums; bit-identical while out3 runs the same kernel form in
    # both (direct-form K3), re-association level when the fused one is the Winograd kernel and the small unfused
    # volume stays with K3 (ops.WINO_MIN_BLOCKS)
    net.feature.fuse_topdown = False
    assert ((net(*args)["depth"] - base).abs() / base.abs()).max().item() < 2e-5   # depths ~600 mm: a few fp32 ulps
    ops.use_wino = False
    try:
        unfused = net(*args)["depth"].clone()
        net.feature.fuse_topdown = True
        assert torch.equal(net(*args)["depth"], unfused)
    finally:
        ops.use_wino = True
        net.feature.fuse_topdown = True


def test_full_size_properties(k1):
    """BASELINE config-2 stage-1 shape (C=32, D=64, 296x400): properties that need no oracle run --
    linearity of K1 in the source features and additivity over view shards."""
    C, D, H, W, V = 32, 64, 296, 400, 5
    g = torch.Generator(device="cpu").manual_seed(0)
    feats = [torch.randn(H, W, C, generator=g) for _ in range(V)]
    cams = synth.synth_cameras(H * 4, W * 4, V)["stage1"]
    p12 = ops.relative_proj(cu(cams[0]))
    hyp, _ = ops.hypotheses_first(cu(synth.synth_depth_values()), D, H, W, False)
    ref = k1.hwc(cu(feats[0]))
    src = [k1.hwc(cu(f)) for f in feats[1:]]
    full = k1(ref, src, p12, hyp)
    parts = k1(ref, src[:2], p12[:2].contiguous(), hyp)
    k1(ref, src[2:], p12[2:].contiguous(), hyp, out=parts, accumulate=True)
    assert_close(parts, full, atol=1e-5)
    scaled = k1(ref, [2.0 * s for s in src], p12, hyp)
    assert_close(scaled, 2.0 * full, atol=1e-5)
    assert torch.isfinite(full).all() and full.abs().mean() > 1e-3
    # and a direct comparison with the oracle on the same shape (two views; a few seconds of CPU)
    nchw = [f.permute(2, 0, 1)[None].contiguous() for f in feats[:3]]
    want = O.warp_corr(nchw, cams[:, :3], hyp.cpu()[None])
    got = k1(ref, src[:2], p12[:2].contiguous(), hyp)
    assert_close(got, want[0], atol=1e-3)  # white-noise features: tap-position rounding x unit gradient
    assert (got.cpu() - want[0]).abs().mean() < 1e-5   # (q4: FMA projection + direct pixel coordinate, see warp_corr.hip)


# ------------------------------------------------------------------------------------------ bench-path instantiations
# The launcher picks the big workgroup tiles only when they yield >= 768 workgroups (conv3d_mfma.hip kMinBlocks); the
# small cases above never get there.  These cases are sized to cross that line for every tile family the full-size
# configs run (dmvs_conv3d_mfma_plan proves which tile is launched), and are compared with ATen on the CPU.
BIG_CASES = [
    # (cin, cout, mode, kd, D, H, W, expected TZ, TY)
    (2, 16, ops.CONV_S1, 3, 16, 128, 192, 2, 8), (16, 16, ops.CONV_S1, 3, 16, 128, 192, 2, 8),
    (16, 16, ops.CONV_S1, 3, 16, 128, 190, 2, 8),    # W % 4 != 0: dword tile loader, scalar stores
    (32, 32, ops.CONV_S1, 3, 16, 128, 192, 2, 8), (64, 64, ops.CONV_S1, 3, 12, 128, 256, 2, 8),
    (8, 16, ops.CONV_S2, 3, 16, 256, 384, 2, 4), (16, 32, ops.CONV_S2, 3, 16, 256, 384, 2, 4),
    (32, 64, ops.CONV_S2, 3, 16, 256, 384, 2, 4),
    (16, 32, ops.CONV_S2, 3, 2, 768, 1024, 1, 8),    # 3D stride-2 layer whose output has depth 1 (refine conv3)
    (32, 64, ops.CONV_S2, 1, 3, 512, 1024, 1, 8), (64, 64, ops.CONV_S1, 1, 3, 256, 512, 1, 16),
    (16, 16, ops.CONV_S1, 1, 3, 256, 512, 1, 16), (32, 32, ops.CONV_S1, 1, 3, 256, 512, 1, 16),
    (64, 32, ops.DECONV_S2, 3, 4, 37, 50, 2, 1), (32, 16, ops.DECONV_S2, 3, 8, 74, 100, 2, 2),   # conv7: two 16-channel blocks
    (16, 8, ops.DECONV_S2, 3, 16, 74, 100, 2, 2), (64, 32, ops.DECONV_S2, 1, 1, 74, 100, 1, 2),  # split over the wave pairs
    (64, 32, ops.DECONV_S2, 3, 1, 148, 200, 1, 2),   # conv7 of stage 3: a 3D transposed conv on a depth-1 input
    (16, 8, ops.DECONV_S2, 3, 1, 148, 200, 1, 4),    # depth-1 input of a 3D transposed conv (refine conv11 at D 1 -> 2)
    # two-block (Cout = 64) layers between the thresholds: the plain small tiles (>= 1024 workgroups) ...
    (64, 64, ops.CONV_S1, 3, 8, 128, 128, 2, 2), (32, 64, ops.CONV_S2, 3, 16, 256, 256, 2, 2),
    (64, 64, ops.CONV_S1, 1, 4, 128, 256, 1, 4),
    # ... and the M-block-split tiles the 1/8-scale bottleneck of every config-2 pass runs (plan bit 17)
    (64, 64, ops.CONV_S1, 3, 4, 74, 100, 2, 1), (32, 64, ops.CONV_S2, 3, 8, 148, 200, 2, 1),
    (64, 64, ops.CONV_S1, 3, 1, 148, 200, 1, 2), (32, 64, ops.CONV_S2, 1, 1, 74, 100, 1, 2),
    (64, 64, ops.CONV_S1, 1, 1, 37, 50, 1, 2),
]
MBS_TILES = {(2, 1), (1, 2)}   # only the split variant uses these tile shapes


@pytest.mark.parametrize("case", BIG_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv3d_big_tiles(case):
    from dmvsnet_amd import _lib
    cin, cout, mode, kd, D, H, W, tz, ty = case
    plan = _lib.load().dmvs_conv3d_mfma_plan(cin, cout, D, H, W, mode, kd)
    assert plan > 0 and ((plan >> 8) & 255, plan & 255) == (tz, ty), (case, hex(plan))
    assert bool(plan & (1 << 17)) == ((tz, ty) in MBS_TILES and not mode == ops.DECONV_S2), (case, hex(plan))
    tr = mode == ops.DECONV_S2
    shape = ((cin, cout) if tr else (cout, cin)) + ((kd, 3, 3) if kd == 3 else (3, 3))
    w = rnd(*shape, seed=cin * 100 + cout + 1, scale=1.0 / np.sqrt(cin * 9 * kd))
    layer, scale, shift = _layer(w, mode, kd, bn=True, seed=cin + cout + 1)
    assert layer.w_mfma is not None
    x = rnd(cin, D, H, W, seed=7)
    Do, Ho, Wo = layer.out_shape(D, H, W)
    use_skip = tr or (cin == cout)
    skip = rnd(cout, Do, Ho, Wo, seed=8) if use_skip else None
    want = _conv_ref(x, w, mode, kd, scale, shift, skip)
    got = ops.conv3d(cu(x), layer, skip=None if skip is None else cu(skip), backend="mfma")
    assert_close(got, want, atol=2e-5, what=str(case))


BIG_FEAT_CASES = [
    # FeatureNet modes at sizes that take the 16-row (stride 1) / 8-row (stride 2) per-slice tiles
    (4, 8, ops.CONV_S1, 3, 256, 512, 16), (8, 8, ops.CONV_S1, 3, 256, 512, 16), (32, 16, ops.CONV_S1, 3, 256, 512, 16),
    (8, 16, ops.CONV2D_K5S2, 3, 512, 1024, 8), (16, 32, ops.CONV2D_K5S2, 3, 512, 1024, 8),
    (32, 64, ops.CONV2D_K1, 3, 256, 512, 16), (16, 32, ops.CONV2D_K1, 3, 256, 512, 16), (8, 32, ops.CONV2D_K1, 3, 256, 512, 16),
]


@pytest.mark.parametrize("case", BIG_FEAT_CASES, ids=lambda c: "x".join(map(str, c)))
def test_feature_big_tiles(case):
    from dmvsnet_amd import _lib
    cin, cout, mode, V, H, W, ty = case
    plan = _lib.load().dmvs_conv3d_mfma_plan(cin, cout, V, H, W, mode, 1)
    assert plan > 0 and ((plan >> 8) & 255, plan & 255) == (1, ty), (case, hex(plan))
    k = 5 if mode == ops.CONV2D_K5S2 else 1 if mode == ops.CONV2D_K1 else 3
    stride = 2 if mode == ops.CONV2D_K5S2 else 1
    w = rnd(cout, cin, k, k, seed=cin * 7 + cout + 1, scale=1.0 / np.sqrt(cin * k * k))
    g = np.random.Generator(np.random.PCG64(cin + cout + 1))
    scale, shift = T((0.5 + g.random(cout)).astype(np.float32)), T((0.2 * g.standard_normal(cout)).astype(np.float32))
    layer = ops.ConvLayer("t", mode, 1, cin, cout, None, cu(ops.pack_mfma(w, cin, cout, mode, 1)), cu(scale), cu(shift), True)
    x = rnd(cin, V, H, W, seed=5)
    want = torch.relu(F.conv2d(x.permute(1, 0, 2, 3), w, None, stride, k // 2) * scale.view(1, -1, 1, 1)
                      + shift.view(1, -1, 1, 1)).permute(1, 0, 2, 3)
    assert_close(ops.conv3d(cu(x), layer), want, atol=2e-5, what=str(case))
    if cout % 8 == 0:   # the quad-planar epilogue the warp kernel's inputs come from
        hw = ops.conv3d(cu(x), layer, out_q4=True)
        assert_close(_q4_halves_to_planar(hw), want, atol=2e-5, what=f"{case} q4")
    if mode == ops.CONV2D_K1:
        sk = rnd(cout, V, want.shape[-2] // 2, want.shape[-1] // 2, seed=6)
        up = F.interpolate(sk.permute(1, 0, 2, 3), scale_factor=2, mode="nearest").permute(1, 0, 2, 3)
        assert_close(ops.conv3d(cu(x), layer, skip=cu(sk), skip_up2=True), want + up, atol=2e-5, what=f"{case} skip_up2")


def test_conv3d_fpn_big_tile():
    """The fused level-3 merge with >= 768 16-row tiles (the variant config 2 runs)."""
    from dmvsnet_amd import _lib
    V, H, W = 3, 256, 512
    assert _lib.load().dmvs_conv3d_mfma_plan(32, 16, V, H, W, ops.CONV_S1, 1) & 255 == 16
    Cl, Cin, Cout = 8, 32, 16
    w_lat, b_lat = rnd(Cin, Cl, seed=1, scale=0.3), rnd(Cin, seed=2, scale=0.2)
    w3 = rnd(Cout, Cin, 3, 3, seed=3, scale=1.0 / np.sqrt(Cin * 9))
    lat, td = rnd(Cl, V, H, W, seed=4), rnd(Cin, V, H // 2, W // 2, seed=5)
    intra = F.conv2d(lat.permute(1, 0, 2, 3), w_lat[:, :, None, None], b_lat) + \
        F.interpolate(td.permute(1, 0, 2, 3), scale_factor=2, mode="nearest")
    want = F.conv2d(intra, w3, None, 1, 1).permute(1, 0, 2, 3)
    layer = ops.ConvLayer("t", ops.CONV_S1, 1, Cin, Cout, None, cu(ops.pack_mfma(w3, Cin, Cout, ops.CONV_S1, 1)), None, None, False)
    layer.w_wino_fpn = cu(ops.pack_wino_fpn(w3, w_lat, b_lat))
    hw = ops.conv3d_fpn(cu(lat), cu(td), layer, out_q4=True)
    assert hw is not None
    assert_close(_q4_halves_to_planar(hw), want, atol=3e-5)


@pytest.mark.parametrize("D", [4, 8, 16, 32, 64])
def test_depth_regress_no_prob_variants(D):
    """K4 without the softmax volume (what eval and the bench run): the register-resident D = 4 / 8 instantiations and
    the generic three-sweep one, vs the oracle and vs the volume-writing instantiation (same bits)."""
    H, W = 12, 300   # W > 256: two workgroups per row, the second ragged
    logits = rnd(1, 4, D, H, W, seed=D, scale=2.0)
    depth = (500.0 + 3.0 * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1) + rnd(1, D, H, W, seed=D + 1, scale=0.5))
    itv = torch.tensor(2.65)
    for mode, alpha, ref in ((0, 1.0, O.depth_regress_main(logits, depth, itv)),
                             (1, 5.0, O.depth_regress_refine(logits, depth, itv, 5.0))):
        a = ops.depth_regress(cu(logits[0]), cu(depth[0]), cu(itv), alpha, mode, False)
        b = ops.depth_regress(cu(logits[0]), cu(depth[0]), cu(itv), alpha, mode, True)
        for x, y in zip(a[:3], b[:3]):
            assert torch.equal(x, y)
        assert a[3] is None and b[3] is not None
        # depths are ~600 mm (fp32 ulp 6e-5): a D-term expectation in a different summation order differs by a few
        # ulp; the six-stack extrapolation (3m - 2M ...) amplifies that by up to 5
        if mode == 0:
            assert_close(a[0], ref["depth_sub_plus"][0], atol=1e-3)
            assert_close(a[1], ref["depth_values_c"][0], atol=5e-3)
            assert_close(a[2], ref["photometric_confidence"][0], atol=1e-4)
            assert_close(b[3], ref["prob_volume"][0], atol=1e-6)
        else:
            assert_close(a[0], ref["depth_sub_plus_refine"][0], atol=1e-3)
            assert_close(a[1], ref["depth"][0], atol=1e-3)
            assert_close(a[2], ref["photometric_confidence_refine"][0], atol=1e-4)


@pytest.mark.parametrize("affine", [False, True])
@pytest.mark.parametrize("D,H,W", [(4, 40, 68), (8, 33, 100), (4, 16, 32), (8, 100, 260)])
def test_prob_regress_fused_is_bit_identical(D, H, W, affine):
    """`prob` -> K4 (dmvs_prob_regress + dmvs_depth_select): the fused heads against the two kernels they replace
    (dmvs_conv3d_direct -> logits -> dmvs_depth_regress) BIT FOR BIT -- expectations, selection and confidence, main (alpha 1, four
    refine hypotheses) and refine mode (alpha 5, one depth) -- on shapes with ragged tile rows / columns (H % 16, (W - 1) % 32), one
    and several tiles, both hypothesis forms; and against the oracle's softmax / expectation at the K4 test's bounds."""
    g = np.random.Generator(np.random.PCG64(D * 1000 + W))
    xs = [cu(rnd(8, D, H, W, seed=50 + i, scale=1.0)) for i in range(2)]
    layers = []
    for i in range(2):
        w = T((0.15 * g.standard_normal((2, 8, 3, 3, 3))).astype(np.float32))
        layers.append(ops.ConvLayer(f"t{i}.prob", ops.CONV_S1, 3, 8, 2, cu(ops.pack_direct(w, False)), None, None, None, False))
    itv = cu(torch.tensor(2.65))
    if affine:
        base = cu((500.0 + rnd(H, W, seed=7, scale=20.0)).contiguous())
        hyp = ops.AffinePlanes(base, itv, D)
        vol = hyp.volume()
    else:
        vol = cu((500.0 + 3.0 * torch.arange(D, dtype=torch.float32).view(D, 1, 1) + rnd(D, H, W, seed=D + 1, scale=0.5)).contiguous())
        hyp = vol
    logits = torch.empty((4, D, H, W), dtype=torch.float32, device="cuda")
    for i in range(2):
        ops.conv3d(xs[i], layers[i], out=logits[2 * i:2 * i + 2], backend="direct")
    for mode, alpha in ((0, 1.0), (1, 5.0)):
        dsp_ref, sel_ref, conf_ref, _ = ops.depth_regress(logits, hyp, itv, alpha, mode, False)
        dsp = torch.full((4, H, W), float("nan"), dtype=torch.float32, device="cuda")
        for i in range(2):
            assert ops.prob_regress(xs[i], layers[i], hyp, itv, alpha, dsp[2 * i:2 * i + 2])
        sel, conf = ops.depth_select(dsp, itv, mode)
        assert torch.equal(dsp, dsp_ref) and torch.equal(sel, sel_ref) and torch.equal(conf, conf_ref), (mode, D, H, W)
        ref = (O.depth_regress_main if mode == 0 else O.depth_regress_refine)(logits.cpu()[None], vol.cpu()[None], itv.cpu(), *(() if mode == 0 else (alpha,)))
        assert_close(dsp, ref["depth_sub_plus" if mode == 0 else "depth_sub_plus_refine"][0], atol=1e-3)
    # shapes the fused form declines: the caller keeps the two kernels
    assert not ops.prob_regress(cu(rnd(8, 16, 16, 32, seed=1)), layers[0], cu(rnd(16, 16, 32, seed=2)), itv, 1.0, torch.empty((2, 16, 32), device="cuda"))
    assert not ops.prob_regress(cu(rnd(8, 4, 16, 30, seed=1)), layers[0], cu(rnd(4, 16, 30, seed=2)), itv, 1.0, torch.empty((2, 16, 30), device="cuda"))


def test_fused_heads_end_to_end_equal_two_kernels():
    """The whole forward with the fused heads (the default wherever D is 4 or 8 and prob_volume is not asked for) against the same
    forward with ops.use_prob_fused = False: every output BIT FOR BIT (linear and inverse-depth sampling)."""
    for inverse in (False, True):
        net, _ = _net([16, 8, 8], [3, 2, 1], 3, inverse)
        net.return_prob_volume = False
        imgs, proj, dv = synth.synth_inputs(128, 160, 4, 3)
        args = (cu(imgs), {k: cu(v) for k, v in proj.items()}, cu(dv))
        try:
            ops.launch_log = []
            a = {k: v.clone() for k, v in net(*args).items() if torch.is_tensor(v)}
            n_fused = len(ops.launch_log)
            ops.use_prob_fused = False
            ops.launch_log = []
            b = net(*args)
            n_two = len(ops.launch_log)
        finally:
            ops.use_prob_fused = True
            ops.launch_log = None
        assert n_fused == n_two     # (5 passes of this config are fused: prob x 2 + K4 -> head x 2 + select: the same launch count)
        for k, v in a.items():
            assert torch.equal(v, b[k]), (inverse, k)


@pytest.mark.parametrize("C", [8, 16, 32])
def test_warp_corr_scattered_hypotheses(C, k1):
    """Neighbouring pixels with very different hypotheses (the refine passes' checkerboard of small / huge
    estimates): the window of a tile spans the whole depth scatter, so the q4 kernel stages it in channel slabs (or, where
    even one quad plane does not fit, takes its global-tap path); the generic kernel additionally gets a padded pixel
    stride (features handed over as a channel slice of a wider tensor)."""
    D, H, W, V = 4, 48, 160, 3
    feats = [_smooth(rnd(1, 2 * C, H, W, seed=70 + v)) * 3 for v in range(V)]
    cams = synth.synth_cameras(H * 4, W * 4, V)["stage1"]
    depth = 450.0 + 400.0 * torch.rand(1, D, H, W, generator=torch.Generator().manual_seed(1))
    want = O.warp_corr([f[:, :C].contiguous() for f in feats], cams, depth)
    p12 = ops.relative_proj(cu(cams[0]))
    if k1.layout == "q4":
        fs = [k1.feat(f[:, :C].contiguous()) for f in feats]
        sim = k1(fs[0], fs[1:], p12, cu(depth[0]))
    else:
        wide = [cu(f[0].permute(1, 2, 0).contiguous()) for f in feats]            # [H, W, 2C]: pix_stride = 2C
        sim = k1(wide[0], wide[1:], p12, cu(depth[0]), C=C, pix_stride=2 * C)
    assert_close(sim, want[0], atol=1e-4)
    assert (sim.cpu() - want[0]).abs().mean() < 1e-5


@pytest.mark.parametrize("C", [8, 32])
def test_warp_corr_behind_camera_and_far_outside(C, k1):
    """Boxes the q4 kernel's corner bound cannot cover: a source camera turned so that part of the depth range lies
    BEHIND it (denominator changes sign inside a tile -> exact global-tap path) and hypotheses that project far outside
    the source image (clamped coordinates, zero border)."""
    D, H, W, V = 6, 40, 96, 3
    feats = [_smooth(rnd(1, C, H, W, seed=90 + v)) * 3 for v in range(V)]
    cams = synth.synth_cameras(H * 4, W * 4, V)["stage1"].clone()
    # view 1: rotate by ~100 degrees about y and move it into the scene; view 2: a large sideways shift
    th = 1.75
    R = torch.tensor([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], dtype=torch.float32)
    cams[0, 1, 0, :3, :3] = R
    cams[0, 1, 0, :3, 3] = torch.tensor([40.0, 3.0, 500.0])
    cams[0, 2, 0, :3, 3] = torch.tensor([-900.0, 150.0, 20.0])
    depth = 300.0 + 150.0 * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1) + rnd(1, D, H, W, seed=5, scale=30.0)
    want = O.warp_corr(feats, cams, depth)
    sim = k1(k1.feat(feats[0]), [k1.feat(f) for f in feats[1:]], ops.relative_proj(cu(cams[0])), cu(depth[0]))
    assert torch.isfinite(sim).all()
    # the behind-camera taps mirror through the image centre with a huge magnification: compare where the oracle is smooth
    diff = (sim.cpu() - want[0]).abs()
    assert diff.mean() < 2e-5 and (diff > 1e-3).float().mean() < 1e-3


def _q4_window_pieces(sx, ox, sy, oy, H, W):
    """Window sizes (16-byte pieces per quad plane) the q4 kernel stages for every 32 x 8 tile under the projection
    ix = sx * x + ox, iy = sy * y + oy (depth-independent): the corner bound of warp_corr.hip restated on the host."""
    out = set()
    f = np.float32
    for ty in range(0, H, 8):
        for tx in range(0, W, 32):
            xs = [f(sx) * f(x) + f(ox) for x in (tx, min(tx + 31, W - 1))]
            ys = [f(sy) * f(y) + f(oy) for y in (ty, min(ty + 7, H - 1))]
            cx = np.clip(xs, -1.0, W)
            cy = np.clip(ys, -1.0, H)
            x0, x1 = max(int(np.floor(cx.min() - 1 / 64)), -1), min(int(np.floor(cx.max() + 1 / 64)) + 1, W + 1)
            y0, y1 = max(int(np.floor(cy.min() - 1 / 64)), -1), min(int(np.floor(cy.max() + 1 / 64)) + 1, H + 1)
            out.add((x1 - x0 + 1) * (y1 - y0 + 1))
    return out


@pytest.mark.parametrize("C,variant,targets", [(32, 0, (318, 319, 637, 638)), (16, 0, (637, 638)),
                                               (32, 2, (425, 426)), (32, 3, (637, 638, 639, 1278)),
                                               (16, 3, (1278,))],   # (317 and 1277 are prime: no such window)
                         ids=["c32_win40", "c16_win40", "c32_win53", "c32_win80", "c16_win80"])
def test_warp_corr_window_sizes_across_the_plane_pitch(C, variant, targets):
    """a slab mode's quad-plane pitch is WINQ / planes rounded DOWN to 4 pieces; windows of pitch + 1 ..
    WINQ / planes pieces used to pass the mode test and then overlapped the next plane in LDS (taps at the window's first /
    last pieces read another channel quad).  Affine projections are searched on the host until tiles with exactly those
    window sizes exist, then the q4 kernel is compared with the generic (global-tap) kernel, MAX-abs."""
    D, H, W = 4, 64, 160
    g = np.random.Generator(np.random.PCG64(11))
    want, picks = set(targets), []
    for _ in range(20000):
        if not want:
            break
        sx, sy = g.uniform(0.6, 2.6), g.uniform(0.6, 4.0)
        ox, oy = g.uniform(0.0, 3.0), g.uniform(0.0, 3.0)
        hit = _q4_window_pieces(sx, ox, sy, oy, H, W) & want
        if hit:
            want -= hit
            picks.append((sx, ox, sy, oy))
    assert not want, f"no projection found for window sizes {sorted(want)}"
    feats = [_smooth(rnd(1, C, H, W, seed=120 + v)) * 3 for v in range(2)]
    depth = cu(500.0 + 10.0 * torch.arange(D, dtype=torch.float32).view(D, 1, 1).expand(D, H, W).contiguous())
    hwc = [_hwc(f) for f in feats]
    q4 = [ops.hwc_to_q4(f) for f in hwc]
    for sx, ox, sy, oy in picks:
        # p(d) = rot (x, y, 1) d + trans with trans = 0: ix = sx x + ox for every depth
        p12 = cu(torch.tensor([[sx, 0, ox, 0, sy, oy, 0, 0, 1, 0, 0, 0]], dtype=torch.float32))
        want_sim = ops.warp_corr(hwc[0], hwc[1:], p12, depth, layout="hwc")
        sim = ops.warp_corr(q4[0], q4[1:], p12, depth, layout="q4", variant=variant)
        assert_close(sim, want_sim, atol=2e-5, what=f"scale ({sx:.3f}, {sy:.3f}) offset ({ox:.2f}, {oy:.2f})")


@pytest.mark.parametrize("C,D,H,W", [(32, 8, 40, 96), (8, 4, 96, 200)])
def test_warp_corr_ten_source_views(C, D, H, W, k1):
    """BASELINE configs[2] / [3] have 11 views: nsrc = 10 in one K1 launch (two 8-view corner tables), vs the oracle."""
    V = 11
    feats = [_smooth(rnd(1, C, H, W, seed=40 + v)) * 3 for v in range(V)]
    cams = synth.synth_cameras(H * 4, W * 4, V)["stage1"]
    depth = 450.0 + 60.0 * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1) + rnd(1, D, H, W, seed=3, scale=4.0)
    want = O.warp_corr(feats, cams, depth)
    sim = k1(k1.feat(feats[0]), [k1.feat(f) for f in feats[1:]], ops.relative_proj(cu(cams[0])), cu(depth[0]))
    assert_close(sim, want[0], atol=1e-4)
    assert (sim.cpu() - want[0]).abs().mean() < 1e-5


def _e2e_vs_oracle(name, H=None, W=None, inverse=False):
    cfg = dict(synth.CONFIGS[name])
    if H is not None:
        cfg.update(H=H, W=W)
    torch.set_num_threads(min(__import__("os").cpu_count() or 1, 32))
    net = MVSNet(cfg["ndepths"], cfg["ratios"], inverse_depth=inverse, verbose=False)
    sd = synth.synth_state_dict(net.state_dict(), 0)
    net.load_state_dict(sd)
    net = net.to(DEV)
    net.return_prob_volume = False                      # the bench / eval configuration
    imgs, proj, dv = synth.synth_inputs(cfg["H"], cfg["W"], cfg["V"], 0)
    out = net(cu(imgs), {k: cu(v) for k, v in proj.items()}, cu(dv))
    torch.cuda.synchronize()
    ref = O.mvsnet_forward(sd, cfg["ndepths"], cfg["ratios"], imgs, proj, dv, inverse_depth=inverse)
    rels = []
    for s in range(len(cfg["ndepths"])):
        d, r = out[f"stage{s + 1}"]["depth"].cpu(), ref[f"stage{s + 1}"]["depth"]
        assert d.shape == r.shape
        rels.append(float((d - r).abs().mean() / r.abs().mean()))
        c, rc = out[f"stage{s + 1}"]["photometric_confidence"].cpu(), ref[f"stage{s + 1}"]["photometric_confidence"]
        assert float((c - rc).abs().mean()) < 1e-4
        # not only means: the worst pixel, and the share of pixels where the ORDER of a (small | huge) pair
        # of regressed depths -- what the checkerboard selection of mvsnet.py:25-56, 80-91 keys on -- differs (measured at
        # c2: 0 / 0 / 0.18 % -- stage-3 pairs that agree to the last bits, where min / max return the same depth either way)
        assert float((d - r).abs().max()) < 0.05, ("max abs depth error (mm)", s, float((d - r).abs().max()))
        flips = torch.zeros(d.shape[-2:], dtype=torch.bool)
        for key in ("depth_sub_plus", "depth_sub_plus_refine"):
            a, b = out[f"stage{s + 1}"][key][0].cpu(), ref[f"stage{s + 1}"][key][0]
            assert float((a - b).abs().max()) < 0.05, (key, s)
            for ch in (0, 2):
                flips |= (a[ch] < a[ch + 1]) != (b[ch] < b[ch + 1])
        assert float(flips.float().mean()) < 5e-3, ("flipped selections", s, float(flips.float().mean()))
    return rels


@pytest.mark.timeout(900)
def test_full_size_c2_end_to_end_vs_oracle():
    """BASELINE configs[1] at FULL size, exactly as bench.py runs it (return_prob_volume=False: the big-tile conv
    instantiations, K4 <false,0> at D = 64 / 32 and <false,8>, <false,4>), vs the oracle: depth rel-L1 per stage
    <= 1e-5 (north-star bound 1e-3).  ~30 s of CPU for the oracle on the GPU box."""
    rels = _e2e_vs_oracle("c2")
    assert all(r < 1e-5 for r in rels), rels


@pytest.mark.timeout(1200)
def test_full_size_c3_end_to_end_vs_oracle():
    """BASELINE configs[2] at FULL size on one GPU (DTU 1600x1184, 11 views, 64/32/8): the shape the 4-GPU config
    names, nsrc = 10 through every full-size kernel instantiation, vs the oracle (about a minute of CPU)."""
    rels = _e2e_vs_oracle("c3")
    assert all(r < 1e-5 for r in rels), rels


@pytest.mark.timeout(1200)
def test_full_size_c4_end_to_end_vs_oracle():
    """BASELINE configs[3] at FULL size on one GPU (Tanks&Temples shape 1920x1024, 11 views, 64/32/8): nsrc = 10
    through every full-size kernel instantiation, vs the oracle (about a minute of CPU on the GPU box)."""
    rels = _e2e_vs_oracle("c4")
    assert all(r < 1e-5 for r in rels), rels


@pytest.mark.timeout(900)
def test_full_size_dtu_recipe_end_to_end_vs_oracle():
    """The reference's OWN DTU eval recipe (scripts/dtu_test.sh:10-29) at full size: 864 x 1152, 5 views, 48 / 32 / 8
    planes, ratios 4 / 2 / 1, --inverse_depth: D = 48 (K4's generic instantiation, K1 with 12 plane chunks), the
    materialised inverse-depth hypothesis volumes through K1 / K4, vs the oracle."""
    rels = _e2e_vs_oracle("dtu", inverse=True)
    assert synth.CONFIGS["dtu"]["inverse"] and all(r < 1e-5 for r in rels), rels


@pytest.mark.timeout(1200)
def test_full_size_tnt_recipe_end_to_end_vs_oracle():
    """The reference's Tanks&Temples recipe (scripts/tank_test.sh:10-23, filter/tank_test_config.py:10-11): a 1080 x 2048
    frame becomes 1056 x 2048 under the loader's base-32 rule (general_eval.py:97-110 -- asserted against eval_io's
    ResizePolicy here), 11 views, 64 / 32 / 8, ratios 3 / 2 / 1, linear sampling; vs the oracle."""
    from dmvsnet_amd.eval_io import ResizePolicy
    cfg = synth.CONFIGS["tnt"]
    assert ResizePolicy(1080, 2048).target(1080, 2048) == (cfg["H"], cfg["W"])
    rels = _e2e_vs_oracle("tnt")
    assert all(r < 1e-5 for r in rels), rels


@pytest.mark.timeout(1800)
def test_full_size_c5_extension_vs_generalised_oracle():
    """BASELINE configs[4]'s shape at FULL size on one GPU (2048x1536, 7 views, 4 stages 96/64/32/8) -- the declared
    extension (MVSNet.stage_level) against the oracle generalised the same way, fp32 features; then the same forward with
    fp16 feature storage against the product's own fp32 result (the parity target of that half of the extension)."""
    rels = _e2e_vs_oracle("c5")
    assert len(rels) == 4 and all(r < 1e-5 for r in rels), rels
    cfg = synth.CONFIGS["c5"]
    net = MVSNet(cfg["ndepths"], cfg["ratios"], verbose=False)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
    net = net.to(DEV)
    net.return_prob_volume = False
    imgs, proj, dv = synth.synth_inputs(cfg["H"], cfg["W"], cfg["V"], 0)
    args = (cu(imgs), {k: cu(v) for k, v in proj.items()}, cu(dv))
    ref = net(*args)["depth"].clone()
    net.feature_dtype = "f16"
    out = net(*args)["depth"]
    assert float((out - ref).abs().mean() / ref.abs().mean()) < 1e-3


@pytest.mark.timeout(900)
def test_eleven_views_end_to_end_vs_oracle():
    """BASELINE configs[2] / [3] shape (11 views, 64/32/8) at a quarter of the linear size, inverse-depth sampling (the
    DTU recipe) -- nsrc = 10 through the whole network."""
    rels = _e2e_vs_oracle("c3", H=288, W=416, inverse=True)
    assert all(r < 2e-5 for r in rels), rels


@pytest.mark.parametrize("C,D,H,W,V", [(32, 8, 24, 72, 3), (16, 11, 40, 100, 3), (8, 4, 64, 130, 5)])
@pytest.mark.parametrize("variant", [0, 16, 3])
def test_warp_corr_fp16_features(C, D, H, W, V, variant):
    """Extension (BASELINE configs[4] "fp16 features"): K1 on fp16 quad-planar features with fp32 products and sums, against
    the fp32 kernel fed with the SAME (fp16-rounded) feature values -- the only difference left is the summation order
    -- and against the oracle on those values; windows staged in pixel pairs, scattered hypotheses (slab modes)."""
    feats = [(_smooth(rnd(1, C, H, W, seed=20 + v)) * 3).half().float() for v in range(V)]   # exactly representable in fp16
    cams = synth.synth_cameras(H * 4, W * 4, V)["stage1"]
    depth = 450.0 + 60.0 * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1) + rnd(1, D, H, W, seed=3, scale=25.0)
    p12 = ops.relative_proj(cu(cams[0]))
    k = _K1("q4", variant)
    f32 = [k.feat(f) for f in feats]
    f16 = [t.half() for t in f32]
    want = k(f32[0], f32[1:], p12, cu(depth[0]))
    got = ops.warp_corr(f16[0], f16[1:], p12, cu(depth[0]), variant=variant)
    assert_close(got, want, atol=1e-5, what="fp16-feature kernel vs fp32 kernel on the same values")
    assert_close(got, O.warp_corr(feats, cams, depth)[0], atol=5e-5)
    # z < 0 / far outside (global-tap path) and affine planes
    planes = ops.AffinePlanes(cu(depth[0, 0].contiguous()), cu(torch.tensor(7.5)), D)
    assert_close(ops.warp_corr(f16[0], f16[1:], p12, planes, variant=variant), k(f32[0], f32[1:], p12, planes), atol=1e-5)


def test_fp16_feature_end_to_end_within_contract():
    """feature_dtype = 'f16' on the whole network (4-stage pyramid of BASELINE configs[4], small size): depth within
    the north-star bound 1e-3 rel-L1 of the product's own fp32 path (the parity target of this extension)."""
    ndepths, ratios = [24, 16, 8, 8], [4, 3, 2, 1]
    net = MVSNet(ndepths, ratios, verbose=False)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 4))
    net = net.to(DEV)
    net.return_prob_volume = False
    imgs, proj, dv = synth.synth_inputs(96, 160, 5, 4)
    args = (cu(imgs), {k: cu(v) for k, v in proj.items()}, cu(dv))
    ref = net(*args)["depth"].clone()
    net.feature_dtype = "f16"
    out = net(*args)["depth"]
    rel = float((out - ref).abs().mean() / ref.abs().mean())
    assert 0 < rel < 1e-3, rel


def test_four_stage_pyramid_extension_vs_generalised_oracle():
    """BASELINE configs[4] names a 4-stage pyramid, which the reference cannot express (KeyError 'stage4', SURVEY.md
    8c).  The product's declared extension (MVSNet.stage_level: the extra stage runs at the coarsest FPN level, with a
    same-resolution hypothesis transition) against the oracle generalised the same way -- for <= 3 stages both ARE the
    reference's semantics (the golden tests above), so this checks the plumbing of the extra stage: K1 at D = 24 on a
    second coarse pass, the up = 1 hypothesis kernel, per-level feature / projection selection."""
    ndepths, ratios = [24, 16, 8, 8], [4, 3, 2, 1]
    H, W, V = 96, 160, 4
    net = MVSNet(ndepths, ratios, verbose=False)
    assert [net.stage_level(s) for s in range(4)] == [0, 0, 1, 2]
    assert [MVSNet([8, 8, 8], [3, 2, 1], verbose=False).stage_level(s) for s in range(3)] == [0, 1, 2]
    sd = synth.synth_state_dict(net.state_dict(), 2)
    net.load_state_dict(sd)
    net = net.to(DEV)
    net.return_prob_volume = False
    imgs, proj, dv = synth.synth_inputs(H, W, V, 2)      # the reference's three projection scales only
    out = net(cu(imgs), {k: cu(v) for k, v in proj.items()}, cu(dv))
    ref = O.mvsnet_forward(sd, ndepths, ratios, imgs, proj, dv)
    for s in range(4):
        d, r = out[f"stage{s + 1}"]["depth"].cpu(), ref[f"stage{s + 1}"]["depth"]
        assert d.shape == r.shape == (1, H // (4, 4, 2, 1)[s], W // (4, 4, 2, 1)[s])
        assert float((d - r).abs().mean() / r.abs().mean()) < 2e-5, s
    # an explicit "stage4" entry (a loader written for the extension) is used when present: same result
    proj4 = {"stage1": proj["stage1"], "stage2": proj["stage1"], "stage3": proj["stage2"], "stage4": proj["stage3"]}
    out4 = net(cu(imgs), {k: cu(v) for k, v in proj4.items()}, cu(dv))
    assert torch.equal(out4["depth"], out["depth"])
    # inverse-depth sampling through the same-resolution transition (volume form of the hypothesis kernel)
    neti = MVSNet(ndepths, ratios, inverse_depth=True, verbose=False)
    neti.load_state_dict(sd)
    neti = neti.to(DEV)
    neti.return_prob_volume = False
    outi = neti(cu(imgs), {k: cu(v) for k, v in proj.items()}, cu(dv))
    refi = O.mvsnet_forward(sd, ndepths, ratios, imgs, proj, dv, inverse_depth=True)
    assert float((outi["depth"].cpu() - refi["depth"]).abs().mean() / refi["depth"].abs().mean()) < 2e-5


def test_graph_replay_matches_eager():
    """MVSNet.use_graph: the forward captured into one HIP graph (side streams included) and replayed gives the same
    bits as the eager launches, also after the inputs change, and re-captures when the shape does."""
    net, _ = _net([16, 8, 8], [3, 2, 1], 2)
    net.return_prob_volume = False
    outs = {}
    for seed, (H, W) in ((1, (64, 96)), (2, (64, 96)), (3, (96, 128))):
        imgs, proj, dv = synth.synth_inputs(H, W, 3, seed)
        args = (cu(imgs), {k: cu(v) for k, v in proj.items()}, cu(dv))
        net.use_graph = False
        eager = {k: v.clone() for k, v in net(*args).items() if torch.is_tensor(v)}
        net.use_graph = True
        got = net(*args)
        torch.cuda.synchronize()
        for k, v in eager.items():
            assert torch.equal(got[k], v), (seed, k)
        assert set(got["stage2"].keys()) == {k for k in got.keys() if not k.startswith("stage")}
        outs[seed] = got["depth"]
    assert outs[1] is outs[2]          # same shape: same static output (overwritten by the replay)
    assert outs[3] is not outs[2]      # new shape: new capture


# ------------------------------------------------------------------------------------------ N2: affine hypotheses
def test_affine_hypotheses_vs_golden_and_volume_path(golden):
    """Linear sampling in affine form (plane d = base + d * interval): the base planes against plane 0 of the
    reference's volumes, the re-materialised volume against the whole of them, and K1 / K4 fed with the affine form
    against the same kernels fed with the volume."""
    g = golden("op_hypotheses.npz")
    dv = synth.synth_depth_values()
    a, i = ops.hypotheses_first(cu(dv), 8, 6, 8, False, affine=True)
    assert isinstance(a, ops.AffinePlanes) and a.shape == (8, 6, 8)
    assert_close(a.base, g["first_inv0"][0][0], atol=2e-4)
    assert_close(a.volume(), g["first_inv0"][0], atol=3e-4)
    assert_close(i, g["first_inv0_itv"], atol=1e-5)
    a, i = ops.hypotheses_next(cu(g["last"][0]), cu(dv), 2.0, 8, False, affine=True)
    assert_close(a.base, g["later_inv0_up"][0][0], atol=3e-4)
    assert_close(a.volume(), g["later_inv0_up"][0], atol=4e-4)
    assert_close(i, g["later_inv0_itv"], atol=1e-5)
    # inverse-depth sampling is not affine: the volume comes back
    v, _ = ops.hypotheses_next(cu(g["last"][0]), cu(dv), 2.0, 8, True, affine=True)
    assert torch.is_tensor(v)
    # K1 / K4 on a realistic stage: affine vs volume
    C, D, H, W, V = 16, 8, 40, 96, 3
    feats = [_smooth(rnd(1, C, H, W, seed=90 + v)) * 3 for v in range(V)]
    cams = synth.synth_cameras(H * 2, W * 2, V)["stage2"]
    last = (600.0 + 40.0 * torch.rand(H // 2, W // 2, generator=torch.Generator().manual_seed(3)))
    planes, itv = ops.hypotheses_next(cu(last), cu(dv), 2.0, D, False, affine=True)
    vol, itv2 = ops.hypotheses_next(cu(last), cu(dv), 2.0, D, False)
    assert_close(planes.volume(), vol, atol=2e-4)
    p12 = ops.relative_proj(cu(cams[0]))
    for k in (_K1("q4", 0), _K1("q4", 8), _K1("q4", 3), _K1("hwc", 0)):
        fs = [k.feat(f) for f in feats]
        s_a = k(fs[0], fs[1:], p12, planes)
        s_v = k(fs[0], fs[1:], p12, vol)
        assert_close(s_a, s_v, atol=2e-5)
    want = O.warp_corr(feats, cams, vol.cpu()[None])
    assert_close(s_a, want[0], atol=5e-5)
    logits = cu(rnd(4, D, H, W, seed=5, scale=2.0))
    for mode, alpha in ((0, 1.0), (1, 5.0)):
        ra = ops.depth_regress(logits, planes, itv, alpha, mode, mode == 0)
        rv = ops.depth_regress(logits, vol, itv2, alpha, mode, mode == 0)
        assert_close(ra[0], rv[0], atol=1e-3)
        assert_close(ra[1], rv[1], atol=5e-3)
        assert_close(ra[2], rv[2], atol=1e-4)


def test_affine_hypotheses_end_to_end_knob():
    """The whole network with planes formed inside K1 / K4 (default) against the materialised-volume path, and the
    output dict without the volumes (what bench / eval run)."""
    net, _ = _net([16, 8, 8], [3, 2, 1], 4)
    imgs, proj, dv = synth.synth_inputs(64, 96, 3, 4)
    args = (cu(imgs), {k: cu(v) for k, v in proj.items()}, cu(dv))
    a = net(*args)
    assert a["depth_values"].shape == (1, 8, 64, 96) and a["stage1"]["depth_values"].shape == (1, 16, 16, 24)
    net.affine_hypotheses = False
    b = net(*args)
    for s in ("stage1", "stage2", "stage3"):
        rel = ((a[s]["depth"] - b[s]["depth"]).abs().mean() / b[s]["depth"].abs().mean()).item()
        assert rel < 2e-6, (s, rel)
        assert_close(a[s]["depth_values"], b[s]["depth_values"], atol=5e-3)   # later stages: the volumes follow last_depth
    net.affine_hypotheses = True
    net.return_depth_values = False
    net.return_prob_volume = False
    c = net(*args)
    assert "depth_values" not in c and "depth_values" not in c["stage2"] and torch.equal(c["depth"], a["depth"])


def test_batch_of_two_equals_two_calls():
    """mvsnet.py:188-260 accepts any batch size; here a batch is its samples one after the other."""
    net, _ = _net([16, 8, 8], [3, 2, 1], 6)
    net.return_prob_volume = False
    a = synth.synth_inputs(64, 96, 3, 6)
    b = synth.synth_inputs(64, 96, 3, 7)
    imgs = cu(torch.cat((a[0], b[0]), 0))
    proj = {k: cu(torch.cat((a[1][k], b[1][k]), 0)) for k in a[1]}
    dv = cu(torch.cat((a[2], b[2]), 0))
    both = net(imgs, proj, dv)
    assert both["depth"].shape == (2, 64, 96) and both["stage1"]["depth_values"].shape == (2, 16, 16, 24)
    for i, s in enumerate((a, b)):
        one = net(cu(s[0]), {k: cu(v) for k, v in s[1].items()}, cu(s[2]))
        assert torch.equal(both["depth"][i], one["depth"][0])
        assert torch.equal(both["stage2"]["photometric_confidence"][i], one["stage2"]["photometric_confidence"][0])
