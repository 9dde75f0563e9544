"""CPU: eval-side I/O (row N3): PFM format, cam files, the test-mode dataset on a synthetic on-disk scene.
The reference's loader cannot be imported here (cv2 is absent), so these tests pin the FORMAT (PFM spec: header,
negative scale = little endian, rows bottom-up; the MVSNet cam.txt layout) and the loader's arithmetic contract
(general_eval.py:69,97-105,178-198)."""
import os

import numpy as np
import pytest
import torch

from dmvsnet_amd import eval_io, synth


def test_pfm_bytes_and_roundtrip(tmp_path):
    a = np.arange(6, dtype=np.float32).reshape(2, 3) + 0.5
    p = str(tmp_path / "a.pfm")
    eval_io.save_pfm(p, a)
    raw = open(p, "rb").read()
    assert raw.startswith(b"Pf\n3 2\n-1.000000\n")                       # grey, W H, little-endian
    body = np.frombuffer(raw[len(b"Pf\n3 2\n-1.000000\n"):], "<f4").reshape(2, 3)
    assert np.array_equal(body, a[::-1])                                   # bottom row first
    b, scale = eval_io.read_pfm(p)
    assert scale == 1.0 and np.array_equal(b, a)
    c = np.random.default_rng(0).random((4, 5, 3)).astype(np.float32)
    eval_io.save_pfm(p, c)
    assert open(p, "rb").read(3) == b"PF\n"
    assert np.array_equal(eval_io.read_pfm(p)[0], c)
    with pytest.raises(Exception):
        eval_io.save_pfm(p, a.astype(np.float64))


def _write_scene(root, scan, H, W, V, depth_line="425.0 2.5"):
    from PIL import Image
    os.makedirs(os.path.join(root, scan, "cams"))
    os.makedirs(os.path.join(root, scan, "images"))
    imgs = synth.synth_images(H, W, V, seed=3)[0]
    cams = synth.synth_cameras(H, W, V)["stage3"][0].numpy()              # full-resolution intrinsics
    for v in range(V):
        Image.fromarray((imgs[v].permute(1, 2, 0).numpy() * 255).astype(np.uint8)).save(
            os.path.join(root, scan, "images", f"{v:08d}.jpg"), quality=95)
        with open(os.path.join(root, scan, "cams", f"{v:08d}_cam.txt"), "w") as f:
            f.write("extrinsic\n")
            for r in range(4):
                f.write(" ".join(repr(float(x)) for x in cams[v, 0, r]) + "\n")
            f.write("\nintrinsic\n")
            for r in range(3):
                f.write(" ".join(repr(float(x)) for x in cams[v, 1, r, :3]) + "\n")
            f.write("\n" + depth_line + "\n")
    with open(os.path.join(root, scan, "pair.txt"), "w") as f:
        f.write(f"{V}\n")
        for v in range(V):
            others = [u for u in range(V) if u != v]
            f.write(f"{v}\n{len(others)} " + " ".join(f"{u} 1.0" for u in others) + "\n")
    return cams


def test_dataset_identity_branch(tmp_path):
    H, W, V = 64, 96, 3
    cams = _write_scene(str(tmp_path), "scan1", H, W, V)
    ds = eval_io.MVSDataset(str(tmp_path), ["scan1"], "test", 3, 192, 1.06, max_h=1200, max_w=1600)
    assert len(ds) == V
    s = ds[0]
    assert s["imgs"].shape == (3, 3, H, W) and 0 <= s["imgs"].min() and s["imgs"].max() <= 1
    assert s["filename"].format("depth_est", ".pfm") == "scan1/depth_est/00000000.pfm"
    # intrinsics: stage1 = K/4, stage2 = K/2, stage3 = K (general_eval.py:69,189-198); extrinsics untouched
    K = cams[0, 1, :3, :3]
    np.testing.assert_allclose(s["proj_matrices"]["stage3"][0, 1, :2, :3], K[:2], rtol=1e-6)
    np.testing.assert_allclose(s["proj_matrices"]["stage1"][0, 1, :2, :3], K[:2] / 4, rtol=1e-6)
    np.testing.assert_allclose(s["proj_matrices"]["stage2"][0, 1, :2, :3], K[:2] / 2, rtol=1e-6)
    np.testing.assert_allclose(s["proj_matrices"]["stage1"][:, 0], cams[:, 0], rtol=1e-6)
    assert s["proj_matrices"]["stage1"][0, 1, 2, 2] == 1.0
    # depth values: depth_min + i * interval * interval_scale, 192 of them (general_eval.py:183)
    dv = s["depth_values"]
    assert dv.shape == (192,) and dv.dtype == np.float32
    # (np.arange in float32 accumulates its rounded step exactly as the reference's identical call does)
    np.testing.assert_allclose(dv, 425.0 + 2.5 * 1.06 * np.arange(192), rtol=1e-5)
    np.testing.assert_allclose(dv, synth.synth_depth_values()[0].numpy(), rtol=1e-5)
    # inverse-depth sampling (general_eval.py:178-181)
    dsi = eval_io.MVSDataset(str(tmp_path), ["scan1"], "test", 3, 192, 1.06, inverse_depth=True, max_h=1200, max_w=1600)
    dvi = dsi[0]["depth_values"]
    assert dvi[0] == pytest.approx(425.0) and np.all(np.diff(1.0 / dvi) < 0)
    # source views: the first nviews-1 of pair.txt
    assert ds.metas[1] == ("scan1", 1, [0, 2])


def test_dataset_resize_branch_and_depth_range_line(tmp_path):
    H, W, V = 80, 120, 2
    cams = _write_scene(str(tmp_path), "s", H, W, V, depth_line="400.0 2.0 100 700.0")
    ds = eval_io.MVSDataset(str(tmp_path), ["s"], "test", 3, 192, 1.0, max_h=1200, max_w=1600)
    s = ds[0]
    assert s["imgs"].shape == (3, 3, 64, 96)              # floored to multiples of 32; 1 source view repeated
    K = cams[0, 1, :3, :3]
    np.testing.assert_allclose(s["proj_matrices"]["stage3"][0, 1, 0, :3], K[0] * (96 / 120), rtol=1e-6)
    np.testing.assert_allclose(s["proj_matrices"]["stage3"][0, 1, 1, :3], K[1] * (64 / 80), rtol=1e-6)
    # 3-token depth line: interval = (min + n*itv - min) / ndepths (general_eval.py:73-76)
    np.testing.assert_allclose(np.diff(s["depth_values"])[:5], 100 * 2.0 / 192, rtol=1e-4)
    # larger than max: scaled down keeping aspect, then floored to the base
    ds2 = eval_io.MVSDataset(str(tmp_path), ["s"], "test", 2, 192, 1.0, max_h=64, max_w=64)
    assert ds2[0]["imgs"].shape[-2:] == (32, 64)


def test_resize_linear_known_answers():
    """cv2.resize(INTER_LINEAR) semantics on float32 (OpenCV's published algorithm; cv2 itself is absent): hand-computed
    vectors with exactly representable weights, then the product against the oracle's independent NumPy restatement
    on the ratios the loader meets (1200 -> 1184 rows, a 2048 x 1080 frame into 1920 x 1024 ...)."""
    from oracle import resize_oracle as RO
    r = eval_io.resize_linear
    # x2 up: s = (d + .5) / 2 - .5 = -.25, .25, .75, 1.25 -> edge, 3:1, 1:3, edge
    np.testing.assert_array_equal(r(np.array([[0.0, 8.0]], np.float32), 1, 4), [[0.0, 2.0, 6.0, 8.0]])
    # x2 down: s = 2 d + .5 -> the mean of the pair
    np.testing.assert_array_equal(r(np.arange(8, dtype=np.float32)[None], 1, 4), [[0.5, 2.5, 4.5, 6.5]])
    # 4x4 -> 2x2: separable means
    np.testing.assert_array_equal(r(np.arange(16, dtype=np.float32).reshape(4, 4), 2, 2), [[2.5, 4.5], [10.5, 12.5]])
    # 4 -> 3: s = (d + .5) * 4/3 - .5 = 1/6, 3/2, 17/6 : weights 5/6:1/6, 1/2:1/2, 1/6:5/6
    got = r(np.array([[0.0, 6.0, 12.0, 18.0]], np.float32), 1, 3)
    np.testing.assert_allclose(got, [[1.0, 9.0, 17.0]], rtol=1e-6)
    # colour images keep their channel axis; identity is the same object
    a = np.random.default_rng(0).random((5, 7, 3)).astype(np.float32)
    assert r(a, 5, 7) is a and r(a, 10, 14).shape == (10, 14, 3)
    for (h, w), (nh, nw) in (((1200, 1600), (1184, 1600)), ((108, 204), (96, 192)), ((37, 53), (64, 96)), ((64, 96), (37, 53))):
        img = np.random.default_rng(h).random((h, w, 3)).astype(np.float32)
        np.testing.assert_allclose(r(img, nh, nw), RO.resize_linear(img, nh, nw), rtol=0, atol=2e-7)


def test_resize_policy_targets():
    """general_eval.py:97-110: multiples of 32, rounded down, inside (max_h, max_w) keeping the aspect ratio."""
    P = eval_io.ResizePolicy(1200, 1600)
    assert P.target(1200, 1600) == (1184, 1600) and P.target(1184, 1600) == (1184, 1600) and P.target(80, 120) == (64, 96)
    assert eval_io.ResizePolicy(1024, 1920).target(1080, 2048) == (992, 1920)      # T&T frame: 1024/1080 would give 1941 columns -> 1920/2048: 1012.5 rows -> 992
    assert eval_io.ResizePolicy(64, 64).target(80, 120) == (32, 64)
    img, K = P.apply(np.zeros((1200, 1600, 3), np.float32), np.array([[2892.33, 0, 823.2], [0, 2883.18, 619.07], [0, 0, 1]], np.float32))
    assert img.shape == (1184, 1600, 3) and K[0, 0] == np.float32(2892.33) and K[1, 1] == pytest.approx(2883.18 * 1184 / 1200)


def test_write_cam_layout(tmp_path):
    cam = np.zeros((2, 4, 4), np.float32)
    cam[0] = np.eye(4)
    cam[1, :3, :3] = [[100, 0, 50], [0, 100, 40], [0, 0, 1]]
    p = str(tmp_path / "c.txt")
    eval_io.write_cam(p, cam)
    lines = open(p).read().split("\n")
    assert lines[0] == "extrinsic" and lines[6] == "intrinsic" and lines[1].split() == ["1.0", "0.0", "0.0", "0.0"]
    assert lines[7].split() == ["100.0", "0.0", "50.0"]


@pytest.mark.gpu
def test_save_depth_maps_end_to_end(tmp_path):
    """Scene on disk -> loader -> HIP network -> PFM/cam files (Model.test step 1)."""
    from dmvsnet_amd import MVSNet
    _write_scene(str(tmp_path / "data"), "scan9", 64, 96, 3)
    net = MVSNet([16, 8, 8], [3, 2, 1], verbose=False)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 1))
    net = net.cuda()
    net.return_prob_volume = False
    out = eval_io.save_depth_maps(net, str(tmp_path / "data"), ["scan9"], str(tmp_path / "out"), 3, 1200, 1600)
    assert len(out) == 3 and all(os.path.exists(p) for p in out)
    d, _ = eval_io.read_pfm(out[0])
    c, _ = eval_io.read_pfm(out[0].replace("depth_est", "confidence"))
    assert d.shape == (64, 96) and np.isfinite(d).all() and 0 <= c.min() and c.max() < 1
    ds = eval_io.MVSDataset(str(tmp_path / "data"), ["scan9"], "test", 3, 192, 1.06, max_h=1200, max_w=1600)
    s = ds[0]
    ref = net(torch.from_numpy(s["imgs"])[None].cuda(), {k: torch.from_numpy(v)[None].cuda() for k, v in s["proj_matrices"].items()},
              torch.from_numpy(s["depth_values"])[None].cuda())
    assert np.array_equal(d, ref["depth"][0].cpu().numpy())
    assert os.path.exists(out[0].replace("depth_est", "cams").replace(".pfm", "_cam.txt"))


@pytest.mark.gpu
def test_run_test_both_steps(tmp_path):
    """Model.test end to end: scene on disk -> depth maps -> fusion filter -> PLY (both filter methods)."""
    from dmvsnet_amd import MVSNet
    _write_scene(str(tmp_path / "data"), "scan9", 64, 96, 4)
    net = MVSNet([16, 8, 8], [3, 2, 1], verbose=False)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 1))
    net = net.cuda()
    net.return_prob_volume = False
    for method in ("pcd", "dypcd"):
        out = str(tmp_path / ("out_" + method))
        stats = eval_io.run_test(net, str(tmp_path / "data"), ["scan9"], out, 4, 1200, 1600, conf=(0.0, 0.0, 0.0),
                                 filter_method=method)
        ply = os.path.join(out, "pcd", "mvsnet009_l3.ply")
        assert os.path.exists(ply) and set(stats["scan9"]) == {"photo", "geo", "final"}
        head = open(ply, "rb").read(200).split(b"end_header")[0]
        assert b"element vertex" in head
        for kind in ("photo", "geo", "final"):
            assert os.path.exists(os.path.join(out, "scan9", "mask", "00000000_{}.png".format(kind)))


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("tag,scan,nviews,inverse", [("A_lin", "scanA", 3, False), ("A_inv", "scanA", 3, True),
                                                      ("B_pad", "scanB", 5, False)])
def test_dataset_equals_reference_loader_bit_for_bit(tag, scan, nviews, inverse):
    """N3 pinned: ``eval_io.MVSDataset`` against the sample dicts the REFERENCE's ``datasets/general_eval.py`` produced
    on the committed on-disk scenes (tests/golden/make_golden_eval.py ran it in the build container) -- images, the three
    proj_matrices scales, depth_values (linear and inverse sampling, the 2-field and the >= 3-field depth line), the
    padded source list and the filename template, bit for bit.  The scenes are base-32 sized, so every cv2.resize of the
    reference is a same-size call; cv2's interpolation for images that do get resized stays unpinned."""
    g = np.load(os.path.join(GOLDEN, "eval_dataset.npz"))
    ds = eval_io.MVSDataset(os.path.join(GOLDEN, "eval_scene"), [scan], "test", nviews, 192, 1.06, inverse_depth=inverse,
                            max_h=1200, max_w=1600)
    assert len(ds) == int(g[f"{tag}.n"])
    for i in range(len(ds)):
        s = ds[i]
        assert s["filename"] == str(g[f"{tag}.{i}.filename"])
        for k, got in (("imgs", s["imgs"]), ("depth_values", s["depth_values"]), ("stage1", s["proj_matrices"]["stage1"]),
                       ("stage2", s["proj_matrices"]["stage2"]), ("stage3", s["proj_matrices"]["stage3"])):
            want = g[f"{tag}.{i}.{k}"]
            assert got.dtype == want.dtype and got.shape == want.shape, (k, got.dtype, want.dtype, got.shape, want.shape)
            assert np.array_equal(got, want), (tag, i, k, np.abs(got.astype(np.float64) - want).max())
