"""Row N4: geometric-consistency kernel vs its oracle restatement, and the fusion host logic on an analytic scene
(a fronto-parallel plane seen by the synthetic cameras: every pixel must be consistent)."""
import numpy as np
import pytest
import torch

from dmvsnet_amd import synth


def _plane_depths(H, W, V, z0=600.0):
    """Depth maps of the world plane z = z0 for the synthetic cameras (extrinsics are world -> camera)."""
    cams = synth.synth_cameras(H, W, V)["stage3"][0].numpy()
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    depths = []
    for v in range(V):
        K, E = cams[v, 1, :3, :3].astype(np.float64), cams[v, 0].astype(np.float64)
        R, t = E[:3, :3], E[:3, 3]
        rays = np.linalg.inv(K) @ np.stack((xs.ravel(), ys.ravel(), np.ones(H * W)))       # camera rays
        # world point = R^T (d*ray - t); its z must be z0  ->  d = (z0 + (R^T t)_z) / (R^T ray)_z
        d = (z0 + (R.T @ t)[2]) / (R.T @ rays)[2]
        depths.append(d.reshape(H, W).astype(np.float32))
    return cams, depths


def test_fold_projection_matches_stepwise():
    from dmvsnet_amd.fusion import fold_projection
    cams, _ = _plane_depths(64, 96, 2)
    P = fold_projection(cams[0, 1, :3, :3], cams[0, 0], cams[1, 1, :3, :3], cams[1, 0]).astype(np.float64)
    Kr, Er, Ks, Es = (cams[0, 1, :3, :3].astype(np.float64), cams[0, 0].astype(np.float64),
                      cams[1, 1, :3, :3].astype(np.float64), cams[1, 0].astype(np.float64))
    x = np.array([40.0, 20.0, 1.0]); d = 612.0
    step = Ks @ ((Es @ np.linalg.inv(Er)) @ np.append(np.linalg.inv(Kr) @ x * d, 1.0))[:3]
    np.testing.assert_allclose(P[:9].reshape(3, 3) @ x * d + P[9:12], step, rtol=1e-5)


def test_oracle_plane_is_consistent():
    from oracle import fusion_oracle as FO
    cams, depths = _plane_depths(64, 96, 3)
    T = torch.from_numpy
    m, rep, dist, rel = FO.check_geometric_consistency(T(depths[0]), T(cams[0, 1, :3, :3]), T(cams[0, 0]), T(depths[1]),
                                                        T(cams[1, 1, :3, :3]), T(cams[1, 0]))
    inside = rep > 0
    assert inside.float().mean() > 0.5                    # most of the view overlaps
    assert dist[inside].max() < 0.05 and rel[inside].max() < 1e-3
    # an inconsistent source depth (10 % off) must be rejected everywhere
    m2, _, _, _ = FO.check_geometric_consistency(T(depths[0]), T(cams[0, 1, :3, :3]), T(cams[0, 0]), T(depths[1] * 1.1),
                                                  T(cams[1, 1, :3, :3]), T(cams[1, 0]))
    assert m2.float().mean() < 0.01


def test_ply_writer(tmp_path):
    from dmvsnet_amd.fusion import write_ply
    p = str(tmp_path / "a.ply")
    write_ply(p, np.array([[1, 2, 3], [4, 5, 6]], np.float32), np.array([[255, 0, 0], [0, 255, 0]], np.uint8))
    raw = open(p, "rb").read()
    head, body = raw.split(b"end_header\n")
    assert b"element vertex 2" in head and b"format binary_little_endian 1.0" in head and len(body) == 2 * 15
    assert np.frombuffer(body[:12], "<f4").tolist() == [1.0, 2.0, 3.0] and body[12:15] == bytes([255, 0, 0])


@pytest.mark.gpu
def test_geo_consistency_kernel_vs_oracle():
    from dmvsnet_amd.fusion import check_geometric_consistency
    from oracle import fusion_oracle as FO
    H, W = 160, 224
    cams, depths = _plane_depths(H, W, 3)
    rng = np.random.default_rng(0)
    T = torch.from_numpy
    for v, noise in ((1, 0.0), (2, 0.004), (1, 0.02)):     # exact, borderline (0.4 % noise vs 1 % gate), mostly rejected
        d_src = (depths[v] * (1 + noise * rng.standard_normal((H, W)))).astype(np.float32)
        d_ref = depths[0].copy(); d_ref[:4, :4] = 0.0        # zero reference depths take the 1e-4 path
        m_o, rep_o, dist, rel = FO.check_geometric_consistency(T(d_ref), T(cams[0, 1, :3, :3]), T(cams[0, 0]), T(d_src),
                                                                T(cams[v, 1, :3, :3]), T(cams[v, 0]))
        votes = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        dsum = torch.zeros((H, W), device="cuda")
        m, rep = check_geometric_consistency(T(d_ref).cuda(), cams[0, 1, :3, :3], cams[0, 0], T(d_src).cuda(),
                                             cams[v, 1, :3, :3], cams[v, 0], vote_sum=votes, depth_sum=dsum)
        m, rep = m.cpu().bool(), rep.cpu()
        # identical away from the thresholds; pixels within rounding of a gate may flip
        near_gate = ((dist - 1.0).abs() < 1e-3) | ((rel - 0.01).abs() < 1e-5)
        assert (m != m_o)[~near_gate].sum() == 0
        assert (m != m_o).float().mean() < 1e-3
        both = m & m_o
        np.testing.assert_allclose(rep[both].numpy(), rep_o[both].numpy(), rtol=2e-5)
        assert torch.equal(votes.cpu(), m.int()) and torch.allclose(dsum.cpu(), rep)
        if noise == 0.0:
            assert m.float().mean() > 0.5
        if noise == 0.02:  # 2 % depth noise against the 1 % gate: a good part of the overlap is rejected
            assert m.float().mean() < 0.8 * (rep_o > -1).float().mean()


@pytest.mark.gpu
def test_filter_depth_on_plane():
    from dmvsnet_amd.fusion import filter_depth
    H, W, V = 96, 128, 4
    cams, depths = _plane_depths(H, W, V, z0=650.0)
    cam = lambda v: (cams[v, 1, :3, :3], cams[v, 0])
    img = synth.synth_images(H, W, 1, 0)[0, 0].permute(1, 2, 0).numpy()
    conf = np.full((H, W), 0.9, np.float32); conf[:, :10] = 0.0
    xyz, rgb, stats = filter_depth(depths[0], conf, cam(0), img, depths[1:], [cam(v) for v in range(1, V)], 0.1, 2)
    assert stats["photo"] == pytest.approx(1 - 10 / W) and 0.3 < stats["final"] <= stats["geo"]
    assert xyz.shape[1] == 3 and rgb.dtype == np.uint8 and len(xyz) == len(rgb) > 1000
    # fused points lie on the plane z = 650; the few at the overlap border mix zero padding into the bilinear
    # sample of the source depth and still pass the 1 % gate (the reference algorithm's own behaviour)
    err = np.abs(xyz[:, 2] - 650.0)
    assert (err < 0.05).mean() > 0.995 and err.max() < 6.5
