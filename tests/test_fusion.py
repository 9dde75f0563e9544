"""Row N4: the fusion filter.  CPU: the oracle restatement against golden vectors produced by RUNNING the reference's
filter/pcd.py and filter/dypcd_tanks.py (tests/golden/make_golden_fusion.py).  GPU: the HIP kernels and the host driver
against the same vectors and the oracle; an analytic plane scene as a sanity check."""
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


PAIRS = [(0, [1, 2, 3, 4]), (1, [0, 2, 3]), (2, [1, 3, 0, 4])]   # the scene of make_golden_fusion.py
CONF = [0.1, 0.2, 0.3]


def _scene():
    H, W, V = 96, 128, 5
    cams, depths, confs, imgs = synth.synth_fusion_scene(H, W, V, seed=0)
    cam = lambda v: (cams[v, 1, :3, :3].copy(), cams[v, 0].copy())   # noqa: E731
    return cams, depths, confs, imgs, cam


def _same_mask(a, b, slack=3):
    """Masks are identical on the machine that generated the golden files; another CPU's ATen matmul / division may
    round a value sitting on a gate the other way, so a few pixels of slack are allowed."""
    return int((np.asarray(a) != np.asarray(b)).sum()) <= slack


def test_oracle_pair_check_matches_reference(golden):
    """check_geometric_consistency (torch and NumPy flavours, pcd.py:203-242) and the nine-gate ladder
    (dypcd_tanks.py:164-184): masks (bit for bit up to gate-rounding), reprojected depths to fp32 rounding."""
    from oracle import fusion_oracle as FO
    g = golden("fusion_geo.npz")
    cams, depths, _, _, cam = _scene()
    T = torch.from_numpy
    for v in range(1, 5):
        m, rep, _, _ = FO.check_geometric_consistency(T(depths[0].copy()), T(cam(0)[0]), T(cam(0)[1]), T(depths[v]),
                                                      T(cam(v)[0]), T(cam(v)[1]))
        assert _same_mask(m.numpy(), g[f"torch.mask.{v}"])
        ok = m.numpy() & g[f"torch.mask.{v}"]
        np.testing.assert_allclose(rep.numpy()[ok], g[f"torch.rep.{v}"][ok], rtol=2e-6)
        m2, rep2 = FO.check_numpy(depths[0].copy(), *cam(0), depths[v], *cam(v))
        assert _same_mask(m2, g[f"np.mask.{v}"])
        ok = m2 & g[f"np.mask.{v}"]
        np.testing.assert_allclose(rep2[ok], g[f"np.rep.{v}"][ok], rtol=2e-6)
        masks, rep3 = FO.check_ladder(depths[0].copy(), *cam(0), depths[v], *cam(v), 1 / 4, 1 / 1300)
        assert _same_mask(masks, g[f"dy.masks.{v}"], slack=9)
        ok = masks[-1] & g[f"dy.masks.{v}"][-1]
        np.testing.assert_allclose(rep3[ok], g[f"dy.rep.{v}"][ok], rtol=2e-6)
        assert 0.2 < m.float().mean() < 0.98      # the scene exercises both outcomes of the gate


@pytest.mark.parametrize("tag,dynamic", [("pcd", False), ("dy", True)])
def test_oracle_filter_depth_matches_reference(golden, tag, dynamic):
    """filter_depth of both drivers run by the reference on a scene folder: the three masks of every reference view
    and the fused point cloud."""
    from oracle import fusion_oracle as FO
    g = golden("fusion_scene.npz")
    _, depths, confs, _, cam = _scene()
    xyz, rgb = [], []
    for r, srcs in PAIRS:
        img = g[f"img.{r}"].astype(np.float32) / 255.0     # the JPEG round trip is part of the reference's pipeline
        out = FO.filter_view(depths[r], cam(r), confs[r], CONF, img, [depths[s] for s in srcs], [cam(s) for s in srcs],
                             thres_view=2, dynamic=dynamic)
        for kind in ("photo", "geo", "final"):
            assert _same_mask(out[kind], g[f"{tag}.mask.{r}.{kind}"], 0 if kind == "photo" else 3), (r, kind)
        xyz.append(out["xyz"]); rgb.append(out["rgb"])
    v = g[f"{tag}.vertex"]
    xyz, rgb = np.concatenate(xyz), np.concatenate(rgb)
    assert abs(len(xyz) - len(v)) <= 9
    if len(xyz) == len(v):   # always the case on the generating machine: same pixels in the same order
        np.testing.assert_allclose(xyz, np.stack((v["x"], v["y"], v["z"]), 1), rtol=1e-5, atol=1e-3)
        assert np.array_equal(rgb, np.stack((v["red"], v["green"], v["blue"]), 1))


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
def test_ladder_kernel_vs_reference(golden):
    """dmvs_geo_consistency_ladder vs the nine masks the reference's dypcd_tanks produced (pixels within rounding of a
    gate may flip; everything else is identical), and the single-gate kernel vs pcd.py's mask."""
    from dmvsnet_amd.fusion import N_LEVELS, check_geometric_consistency
    from oracle import fusion_oracle as FO
    g = golden("fusion_geo.npz")
    _, depths, _, _, cam = _scene()
    T = torch.from_numpy
    H, W = depths[0].shape
    for v in range(1, 5):
        _, _, dist, rel = FO.check_geometric_consistency(T(depths[0].copy()), T(cam(0)[0]), T(cam(0)[1]), T(depths[v]),
                                                         T(cam(v)[0]), T(cam(v)[1]))
        m, rep = check_geometric_consistency(T(depths[0]).cuda(), *cam(0), T(depths[v]).cuda(), *cam(v))
        near = ((dist - 1.0).abs() < 1e-3) | ((rel - 0.01).abs() < 1e-5)
        want = torch.from_numpy(g[f"np.mask.{v}"])
        assert ((m.cpu().bool() != want) & ~near).sum() == 0
        both = (m.cpu().bool() & want).numpy()
        np.testing.assert_allclose(rep.cpu().numpy()[both], g[f"np.rep.{v}"][both], rtol=5e-5)
        lv = torch.zeros((N_LEVELS, H, W), dtype=torch.int32, device="cuda")
        votes = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        m10, _ = check_geometric_consistency(T(depths[0]).cuda(), *cam(0), T(depths[v]).cuda(), *cam(v), 0.25, 1 / 1300,
                                             vote_sum=votes, level_votes=lv)
        masks = g[f"dy.masks.{v}"]
        for i in range(2, 11):
            near_i = ((dist - i * 0.25).abs() < 1e-3) | ((rel - i / 1300).abs() < 1e-5)
            diff = (lv[i - 2].cpu().bool() != torch.from_numpy(masks[i - 2])) & ~near_i
            assert diff.sum() == 0, (v, i, int(diff.sum()))
        assert torch.equal(votes.cpu(), lv[8].cpu()) and torch.equal(m10.cpu().int(), lv[8].cpu())


@pytest.mark.gpu
@pytest.mark.parametrize("tag,dynamic", [("pcd", False), ("dy", True)])
def test_fuse_scene_vs_reference(golden, tmp_path, tag, dynamic):
    """The file-based driver on the scene folder the reference was run on: mask PNGs and point cloud vs the reference's
    (a pixel within rounding of a gate may flip in one source view; the final masks then differ in at most a handful
    of pixels)."""
    from PIL import Image
    from dmvsnet_amd import eval_io
    from dmvsnet_amd.fusion import fuse_scene, read_pair_file
    g = golden("fusion_scene.npz")
    cams, depths, confs, _, cam = _scene()
    for sub in ("cams", "images", "depth_est", "confidence"):
        (tmp_path / sub).mkdir()
    for v in range(5):
        K, E = cam(v)
        eval_io.write_cam(str(tmp_path / "cams/{:0>8}_cam.txt".format(v)), np.stack((E, np.pad(K, ((0, 1), (0, 1))))))
        Image.fromarray(g[f"img.{v}"]).save(str(tmp_path / "images/{:0>8}.jpg".format(v)), quality=100, subsampling=0)
        eval_io.save_pfm(str(tmp_path / "depth_est/{:0>8}.pfm".format(v)), depths[v])
        for suffix, c in (("_stage1", confs[v][0]), ("_stage2", confs[v][1]), ("", confs[v][2])):
            eval_io.save_pfm(str(tmp_path / "confidence/{:0>8}{}.pfm".format(v, suffix)), c)
    with open(tmp_path / "pair.txt", "w") as f:
        f.write(f"{len(PAIRS)}\n")
        for r, srcs in PAIRS:
            f.write(f"{r}\n{len(srcs)} " + " ".join(f"{s} 1.0" for s in srcs) + "\n")
    pairs = read_pair_file(str(tmp_path / "pair.txt"))
    assert pairs == PAIRS
    fuse_scene(pairs, str(tmp_path), str(tmp_path / "out.ply"), conf=CONF, thres_view=2, dynamic=dynamic)
    npix = 0
    for r, _ in PAIRS:
        for kind in ("photo", "geo", "final"):
            got = np.array(Image.open(tmp_path / "mask/{:0>8}_{}.png".format(r, kind))) > 0
            want = g[f"{tag}.mask.{r}.{kind}"]
            assert (got != want).sum() <= (0 if kind == "photo" else 12), (r, kind, int((got != want).sum()))
        npix += int(g[f"{tag}.mask.{r}.final"].sum())
        if dynamic:
            assert (tmp_path / "depth_est/{:0>8}_averaged.pfm".format(r)).exists()
    raw = open(tmp_path / "out.ply", "rb").read()
    head, body = raw.split(b"end_header\n")
    n = int(head.split(b"element vertex ")[1].split()[0])
    assert abs(n - len(g[f"{tag}.vertex"])) <= 36 and len(body) == n * 15
    # the points both clouds share (same pixels in the same order unless a mask pixel flipped): compare as sets of
    # rounded coordinates
    v = g[f"{tag}.vertex"]
    pts = np.frombuffer(body, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    key = lambda x, y, z: set(zip(np.round(x, 1).tolist(), np.round(y, 1).tolist(), np.round(z, 1).tolist()))   # noqa: E731
    a, b = key(pts["x"], pts["y"], pts["z"]), key(v["x"], v["y"], v["z"])
    assert len(a & b) > 0.99 * len(b)


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
