"""Host side of K3w (csrc/conv3d_wino.hip) on the CPU: the packed Winograd filters (dmvs_pack_conv_weights_wino,
_wino_fpn) decoded back from the kernel's consumption order and used in a NumPy restatement of the kernel's arithmetic
(input transform B^T d B, per-position products, output transform A^T M A) must reproduce the direct convolution
(module.py:120-157 Conv2d/Conv3d semantics; the level-3 merge of module.py:333-336).  No GPU, no kernel launch."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dmvsnet_amd import _lib

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)


def _pack(w, cin, cout, kd):
    lib = _lib.load()
    n = lib.dmvs_conv3d_wino_weight_floats(cin, cout, kd)
    assert n > 0
    out = np.empty(n, dtype=np.float32)
    wc = np.ascontiguousarray(w, dtype=np.float32)
    assert lib.dmvs_pack_conv_weights_wino(ctypes.c_void_p(wc.ctypes.data), ctypes.c_void_p(out.ctypes.data), cin, cout, kd) == 0
    return out


def _unpack(p, cin, cout, kd, gpc):
    """-> U[xi][kz][ci][co] from the order chunk, kz, k-group, 16-channel block, quarter, lane, xi % 4."""
    mb_n = cout // 16
    U = np.zeros((16, kd, cin, cout), dtype=np.float64)
    it = iter(p)
    for ci0 in range(0, cin, 4 * gpc):
        for kz in range(kd):
            for g in range(gpc):
                for mb in range(mb_n):
                    for q in range(4):
                        for lane in range(64):
                            for e in range(4):
                                U[4 * q + e, kz, ci0 + 4 * g + lane // 16, mb * 16 + lane % 16] = next(it)
    assert next(it, None) is None
    return U


def _wino_conv(x, U, kd):
    """x [Cin,D,H,W] (H, W even) -> [Cout,D,H,W]: F(2x2,3x3) per plane, direct over the depth taps."""
    cin, D, H, W = x.shape
    cout = U.shape[-1]
    xp = np.pad(x.astype(np.float64), ((0, 0), (kd // 2, kd // 2), (1, 1), (1, 1)))
    y = np.zeros((cout, D, H, W))
    for z in range(D):
        for ty in range(H // 2):
            for tx in range(W // 2):
                M = np.zeros((16, cout))
                for kz in range(kd):
                    d = xp[:, z + kz, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]            # [Cin,4,4]
                    V = np.einsum("ay,cyx,bx->cab", BT, d, BT).reshape(cin, 16)     # B^T d B
                    M += np.einsum("cx,xco->xo", V, U[:, kz])
                Y = np.einsum("ia,abo,jb->oij", AT, M.reshape(4, 4, cout), AT)
                y[:, z, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = Y
    return y


@pytest.mark.parametrize("cin,cout,kd,gpc", [(16, 16, 3, 1), (32, 32, 3, 1), (64, 64, 1, 1), (16, 16, 1, 2), (32, 16, 1, 1)])
def test_packed_filters_are_g_w_gt_and_reproduce_the_convolution(cin, cout, kd, gpc):
    g = np.random.Generator(np.random.PCG64(cin + cout + kd))
    w = (g.standard_normal((cout, cin) + ((3, 3, 3) if kd == 3 else (3, 3))) / np.sqrt(9 * kd * cin)).astype(np.float32)
    U = _unpack(_pack(w, cin, cout, kd), cin, cout, kd, gpc)
    w5 = w.reshape(cout, cin, kd, 3, 3).astype(np.float64)
    want_U = np.einsum("ay,ockyx,bx->abkco", G, w5, G).reshape(16, kd, cin, cout)
    np.testing.assert_allclose(U, want_U, rtol=0, atol=1e-7)        # formed in double, rounded once to fp32
    x = g.standard_normal((cin, 3 if kd == 3 else 2, 6, 8)).astype(np.float32)
    got = _wino_conv(x, U, kd)
    ref = F.conv3d(torch.from_numpy(x)[None].double(), torch.from_numpy(w5), None, 1, (kd // 2, 1, 1))[0].numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)


def test_conv0_packing_pairs_channels_with_depth_taps():
    """Cin = 2: k-slot = (channel l/16 & 1, depth selector l/32); step 0 = taps 0 / 1, step 1 = tap 2 and zeros."""
    g = np.random.Generator(np.random.PCG64(7))
    w = (g.standard_normal((16, 2, 3, 3, 3)) / np.sqrt(54)).astype(np.float32)
    p = _pack(w, 2, 16, 3).reshape(2, 4, 64, 4)
    want_U = np.einsum("ay,ockyx,bx->abkco", G, w.astype(np.float64), G).reshape(16, 3, 2, 16)
    for st in range(2):
        for lane in range(64):
            co, ci, zsel = lane % 16, (lane // 16) & 1, lane // 32
            got = p[st, :, lane, :].reshape(16)
            if st == 1 and zsel == 1:
                assert not got.any()
            else:
                np.testing.assert_allclose(got, want_U[:, 2 if st else zsel, ci, co], rtol=0, atol=1e-7)


def test_fpn_composite_filters_reproduce_the_level3_merge():
    """dmvs_pack_conv_weights_wino_fpn: out3(b + W_lat.lat + up2(td)) as lateral-composite + ones-plane + 9-position
    top-down products (the arithmetic of fpn_wino_kernel) against the three reference ops."""
    lib = _lib.load()
    g = np.random.Generator(np.random.PCG64(11))
    w3 = (g.standard_normal((16, 32, 3, 3)) / np.sqrt(288)).astype(np.float32)
    wl = (0.3 * g.standard_normal((32, 8))).astype(np.float32)
    bl = (0.2 * g.standard_normal(32)).astype(np.float32)
    n = lib.dmvs_conv3d_wino_fpn_weight_floats()
    p = np.empty(n, dtype=np.float32)
    assert lib.dmvs_pack_conv_weights_wino_fpn(*(ctypes.c_void_p(a.ctypes.data) for a in (w3, wl, bl, p))) == 0
    lat_w = p[:3 * 4 * 256].reshape(3, 4, 64, 4).astype(np.float64)     # [group][quarter][lane][xi % 4]
    td_w = p[3 * 4 * 256:].reshape(8, 3, 64, 4).astype(np.float64)
    H, W = 8, 12
    lat = g.standard_normal((8, H, W)).astype(np.float32)
    td = g.standard_normal((32, H // 2, W // 2)).astype(np.float32)
    intra = F.conv2d(torch.from_numpy(lat)[None], torch.from_numpy(wl)[:, :, None, None], torch.from_numpy(bl)) + \
        F.interpolate(torch.from_numpy(td)[None], scale_factor=2, mode="nearest")
    ref = F.conv2d(intra.double(), torch.from_numpy(w3).double(), None, 1, 1)[0].numpy()
    chans = np.concatenate((lat.astype(np.float64), np.ones((1, H, W))), 0)      # 8 lateral planes + the ones plane
    cp = np.pad(chans, ((0, 0), (1, 1), (1, 1)))
    tp = np.pad(td.astype(np.float64), ((0, 0), (1, 1), (1, 1)))
    pos = (0, 1, 3)
    out = np.zeros((16, H, W))
    for ty in range(H // 2):
        for tx in range(W // 2):
            M = np.zeros((16, 16))                                               # [xi][co]
            for c in range(3):
                for lk in range(4):
                    ch = 4 * c + lk if c < 2 else 8
                    V = (BT @ cp[ch, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4] @ BT.T).reshape(16)
                    for co in range(16):
                        M[:, co] += V * lat_w[c, :, lk * 16 + co, :].reshape(16)
            for gk in range(8):
                for lk in range(4):
                    T = tp[4 * gk + lk, ty:ty + 3, tx:tx + 3]
                    c3 = np.stack((T[0] - T[1], T[1], T[1] - T[2]))
                    v9 = np.stack((c3[:, 0] - c3[:, 1], c3[:, 1], c3[:, 1] - c3[:, 2]), 1).reshape(9)
                    for co in range(16):
                        wj = td_w[gk, :, lk * 16 + co, :].reshape(12)
                        assert not wj[9:].any()
                        for j in range(9):
                            M[4 * pos[j // 3] + pos[j % 3], co] += v9[j] * wj[j]
            out[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("ia,abo,jb->oij", AT, M.reshape(4, 4, 16), AT)
    np.testing.assert_allclose(out, ref, rtol=0, atol=3e-6)


def test_wino_plan_tune_and_unsupported_shapes():
    """Host-side entry logic that needs no GPU: which layer shapes K3w is compiled for, the workgroup count the dispatch
    policy reads, the dmvs_tune knobs' argument checks, and null-pointer rejection before any launch."""
    lib = _lib.load()
    for cin, cout, kd in ((16, 16, 3), (32, 32, 3), (64, 64, 3), (64, 64, 1), (16, 16, 1), (32, 32, 1), (32, 16, 1), (2, 16, 3)):
        assert lib.dmvs_conv3d_wino_weight_floats(cin, cout, kd) > 0
    for cin, cout, kd in ((8, 16, 3), (8, 8, 1), (16, 8, 3), (16, 32, 3), (4, 8, 1)):
        assert lib.dmvs_conv3d_wino_weight_floats(cin, cout, kd) == 0
        assert lib.dmvs_conv3d_wino_plan(cin, cout, 4, 16, 32, kd) == _lib.EUNSUPPORTED
    # conv2 on the stage-2 volume of config 2: 2 planes x 8 rows x 32 columns per workgroup
    assert lib.dmvs_conv3d_wino_plan(16, 16, 16, 296, 400, 3) == 13 * 37 * 8
    assert lib.dmvs_conv3d_wino_plan(2, 16, 8, 1184, 1600, 3) == 50 * 148 * 4
    assert lib.dmvs_conv3d_wino_plan(16, 16, 16, 296, 402, 3) == _lib.EUNSUPPORTED      # W % 4 != 0: the 16-byte loader
    assert lib.dmvs_tune(b"wino_stages", 3) == _lib.EINVAL and lib.dmvs_tune(b"wino_stages", 0) == 0
    assert lib.dmvs_tune(b"wino_conv0_grid", 100) == _lib.EINVAL and lib.dmvs_tune(b"wino_conv0_grid", 512) == 0
    assert lib.dmvs_tune(b"wino_persistent", 1) == 0
    assert lib.dmvs_tune(b"no_such_knob", 1) == _lib.EUNSUPPORTED
    assert lib.dmvs_conv3d_wino(None, None, None, None, None, 16, 16, 4, 16, 32, 3, 1, None) == _lib.EINVAL
    assert lib.dmvs_conv3d_wino_fpn2(None, None, None, None, None, None, None, 1, 16, 32, 0, None) == _lib.EINVAL


# ------------------------------------------------------------------------------------------------ K3r (csrc/conv3d_coarse.hip)
def _pack_coarse(w, cin, cout, kd):
    lib = _lib.load()
    n = lib.dmvs_conv3d_coarse_weight_floats(cin, cout, kd)
    assert n > 0
    out = np.empty(n, dtype=np.float32)
    wc = np.ascontiguousarray(w, dtype=np.float32)
    assert lib.dmvs_pack_conv_weights_coarse(ctypes.c_void_p(wc.ctypes.data), ctypes.c_void_p(out.ctypes.data), cin, cout, kd) == 0
    return out


@pytest.mark.parametrize("cin,cout,kd,ncb", [(32, 32, 3, 2), (64, 64, 3, 1), (32, 32, 1, 2), (64, 64, 1, 2)])
def test_coarse_packing_and_split_transforms_reproduce_the_convolution(cin, cout, kd, ncb):
    """K3r's host side and dataflow on the CPU.  The packed filters are read back in the kernel's register order
    [cout group][wave = (transform row i, channel half h)][stage][k-group][kz][block][lane][position p]; then the kernel's
    arithmetic is restated per WAVE: wave (i, h) forms row i of B^T d from two patch rows, the column transform, its 4 positions'
    products over ITS channels, the column half of the output transform (M[i][:] A); the 8 partial results are summed in the
    kernel's fixed order (P_i = (i,0) + (i,1); row 0 = (P0 + P1) + P2, row 1 = (P1 - P2) - P3).  Must equal the direct
    convolution (module.py:120-157 semantics)."""
    g = np.random.Generator(np.random.PCG64(cin + cout + kd + 5))
    w = (g.standard_normal((cout, cin) + ((3, 3, 3) if kd == 3 else (3, 3))) / np.sqrt(9 * kd * cin)).astype(np.float32)
    cps = 16 if kd == 3 else 32
    nst, gph, ncg = cin // cps, cps // 8, cout // (16 * ncb)
    p = _pack_coarse(w, cin, cout, kd).reshape(ncg, 8, nst, gph, kd, ncb, 64, 4).astype(np.float64)
    w5 = w.reshape(cout, cin, kd, 3, 3).astype(np.float64)
    want_U = np.einsum("ay,ockyx,bx->abkco", G, w5, G)          # [i][p][kz][ci][co]
    for cg in range(ncg):
        for wave in range(8):
            i, h = wave & 3, wave >> 2
            for s in range(nst):
                for gg in range(gph):
                    for lane in (0, 17, 38, 63):
                        ci, co0 = s * cps + (h * gph + gg) * 4 + lane // 16, lane % 16
                        for nb in range(ncb):
                            np.testing.assert_allclose(p[cg, wave, s, gg, :, nb, lane, :].T,
                                                       want_U[i, :, :, ci, (cg * ncb + nb) * 16 + co0], rtol=0, atol=1e-7)
    # the split dataflow on one 8 x 8-output group (4 x 4 tiles) per plane
    D, H, W = (2 if kd == 3 else 1), 8, 8
    x = g.standard_normal((cin, D, H, W)).astype(np.float32)
    xp = np.pad(x.astype(np.float64), ((0, 0), (kd // 2, kd // 2), (1, 1), (1, 1)))
    rows = ((0, 2, -1.0), (1, 2, 1.0), (2, 1, -1.0), (1, 3, -1.0))          # row i of B^T d = d[ra] + sg * d[rb]
    y = np.zeros((cout, D, H, W))
    for z in range(D):
        for ty in range(4):
            for tx in range(4):
                S = np.zeros((8, cout, 2))                                  # per wave: the two output COLUMNS of its transform row
                for wave in range(8):
                    i, h = wave & 3, wave >> 2
                    ra, rb, sg = rows[i]
                    M = np.zeros((4, cout))
                    for s in range(nst):
                        for gg in range(gph):
                            for k in range(4):
                                ci = s * cps + (h * gph + gg) * 4 + k
                                for kz in range(kd):
                                    d = xp[ci, z + kz, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
                                    t = d[ra] + sg * d[rb]
                                    v = np.array([t[0] - t[2], t[1] + t[2], t[2] - t[1], t[1] - t[3]])
                                    for cg in range(ncg):
                                        for nb in range(ncb):
                                            c0 = (cg * ncb + nb) * 16
                                            for co in range(16):
                                                M[:, c0 + co] += v * p[cg, wave, s, gg, kz, nb, k * 16 + co, :]
                    S[wave, :, 0] = (M[0] + M[1]) + M[2]
                    S[wave, :, 1] = (M[1] - M[2]) - M[3]
                P = S[:4] + S[4:]
                y[:, z, 2 * ty, 2 * tx:2 * tx + 2] = (P[0] + P[1]) + P[2]
                y[:, z, 2 * ty + 1, 2 * tx:2 * tx + 2] = (P[1] - P[2]) - P[3]
    ref = F.conv3d(torch.from_numpy(x)[None].double(), torch.from_numpy(w5), None, 1, (kd // 2, 1, 1))[0].numpy()
    np.testing.assert_allclose(y, ref, rtol=0, atol=3e-6)


def test_coarse_entry_logic_without_a_gpu():
    """Which shapes K3r is compiled for, argument checks before any launch, and the `k3r_grid` knob."""
    lib = _lib.load()
    for cin, cout, kd, n in ((32, 32, 3, 1 * 8 * 24 * 256), (64, 64, 3, 4 * 8 * 24 * 256), (32, 32, 1, 1 * 8 * 8 * 256), (64, 64, 1, 2 * 8 * 16 * 256)):
        assert lib.dmvs_conv3d_coarse_weight_floats(cin, cout, kd) == n
    for cin, cout, kd in ((16, 16, 3), (32, 64, 3), (64, 32, 1), (2, 16, 3), (32, 32, 2)):
        assert lib.dmvs_conv3d_coarse_weight_floats(cin, cout, kd) == 0
    assert lib.dmvs_conv3d_coarse(None, None, None, None, None, 32, 32, 4, 16, 32, 3, 1, None) == _lib.EINVAL
    assert lib.dmvs_tune(b"k3r_grid", 100) == _lib.EINVAL and lib.dmvs_tune(b"k3r_grid", 256) == 0


# ------------------------------------------------------------------------------------------ K3z (csrc/conv3d_zmarch.hip)
def test_zmarch_packing_and_the_marching_schedule_reproduce_the_convolution():
    """K3z on the CPU: the packed filters decoded from the kernel's register order -- [wave = transform row i][k-group][kz][lane
    (cout = l % 16, channel = 4 kg + l / 16)][position p] = (G g G^T)[i][p] -- and a NumPy walk of the kernel's schedule: an input
    plane is transformed once (row i of B^T d by wave i, then the column transform), serves depth tap 0 / 1 / 2 of the output planes
    pz + 1 / pz / pz - 1, the wave's four positions are reduced to the two output columns (M[i][:] A) and the four waves' partial
    results are summed over i (A^T): any cut of the planes into z segments (a.zs, the balanced tail) gives the direct convolution."""
    lib = _lib.load()
    g = np.random.Generator(np.random.PCG64(160))
    w = (g.standard_normal((16, 16, 3, 3, 3)) / np.sqrt(27 * 16)).astype(np.float32)
    n = lib.dmvs_conv3d_zmarch_weight_floats(16, 16, 3)
    assert n == 4 * 12 * 256
    p = np.empty(n, dtype=np.float32)
    assert lib.dmvs_pack_conv_weights_zmarch(ctypes.c_void_p(w.ctypes.data), ctypes.c_void_p(p.ctypes.data), 16, 16, 3) == 0
    assert lib.dmvs_conv3d_zmarch_weight_floats(32, 32, 3) == 0 and lib.dmvs_conv3d_zmarch_weight_floats(16, 16, 1) == 0
    U = np.zeros((4, 4, 3, 16, 16))   # [i][p][kz][ci][co]
    it = iter(p)
    for i in range(4):
        for kg in range(4):
            for kz in range(3):
                for lane in range(64):
                    for pp in range(4):
                        U[i, pp, kz, 4 * kg + lane // 16, lane % 16] = next(it)
    want = np.einsum("ay,ockyx,bx->abkco", G, w.astype(np.float64), G)
    np.testing.assert_allclose(U, want, rtol=0, atol=1e-7)

    D, H, W = 5, 4, 6
    x = g.standard_normal((16, D, H, W)).astype(np.float32)
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (1, 1), (1, 1)))
    ref = F.conv3d(torch.from_numpy(x)[None].double(), torch.from_numpy(w).double(), None, 1, 1)[0].numpy()
    for zs in (1, 2, 3, 5):
        y = np.zeros((16, D, H, W))
        for z0 in range(0, D, zs):
            zse = min(zs, D - z0)
            for ty in range(H // 2):
                for tx in range(W // 2):
                    acc = {}                                       # output plane -> [i][p][co] Winograd-domain sums
                    for t in range(zse + 2):                       # one stage per input plane z0 - 1 + t
                        pz = z0 - 1 + t
                        if 0 <= pz < D:
                            d = xp[:, pz, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
                            V = np.einsum("ay,cyx,bx->abc", BT, d, BT)             # [i][p][ci]: wave i holds row i
                        else:
                            V = np.zeros((4, 4, 16))
                        for kz in range(3):
                            u = t - kz                             # local output plane this plane is depth tap kz of
                            if 0 <= u < zse:
                                acc.setdefault(u, np.zeros((4, 4, 16)))
                                acc[u] += np.einsum("ipc,ipco->ipo", V, U[:, :, kz])
                        if t >= 2:                                 # output plane t - 2 is complete
                            M = acc.pop(t - 2)
                            S = np.einsum("ipo,jp->ijo", M, AT)                     # per wave: its 4 positions -> 2 output columns
                            Y = np.einsum("ri,ijo->orj", AT, S)                     # finish: row sum over the waves
                            y[:, z0 + t - 2, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = Y
                    assert not acc
        np.testing.assert_allclose(y, ref, rtol=0, atol=2e-6, err_msg=f"zs = {zs}")


def test_split_probe_weights_are_three_exact_bf16_terms():
    """The bf16-split probe's packer (csrc/conv3d_split.hip): every fp32 weight is the EXACT sum of its three truncated bf16 terms,
    stored in MFMA B-operand order [term][step][lane = k-slice * 16 + cout][8 input channels], tap = 4 step + k-slice, tap 27 zero."""
    lib = _lib.load()
    g = np.random.Generator(np.random.PCG64(8))
    w = (g.standard_normal((16, 8, 3, 3, 3)) * 0.07).astype(np.float32)
    n = lib.dmvs_conv3d_split_weight_floats(8, 16)
    assert n == 3 * 7 * 64 * 4 and lib.dmvs_conv3d_split_weight_floats(16, 16) == 0
    out = np.empty(n, dtype=np.float32)
    assert lib.dmvs_pack_conv_weights_split(ctypes.c_void_p(w.ctypes.data), ctypes.c_void_p(out.ctypes.data), 8, 16) == 0
    bits = out.view(np.uint16).reshape(3, 7, 64, 8).astype(np.uint32) << 16
    terms = bits.view(np.float32).astype(np.float64)               # [term][step][lane][ci]
    wf = w.reshape(16, 8, 27).astype(np.float64)
    for s in range(7):
        for lane in range(64):
            tap, co = 4 * s + lane // 16, lane % 16
            tot = terms[:, s, lane, :].sum(0)
            if tap < 27:
                np.testing.assert_array_equal(tot, wf[co, :, tap])   # h + m + l == w, exactly
                assert np.all(np.abs(terms[1, s, lane]) <= np.abs(terms[0, s, lane]) * 2.0 ** -7 + 1e-45)
            else:
                assert not tot.any()
