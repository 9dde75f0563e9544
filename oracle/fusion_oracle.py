"""ORACLE -- test infrastructure only.  CPU restatement of the fusion filter (/root/reference/filter/pcd.py:151-361
and the dynamic-threshold variant filter/dypcd_tanks.py:164-326), stepwise as the reference computes it (the same
ATen / NumPy ops on the CPU).

PINNED: tests/test_fusion.py checks it against tests/golden/fusion_geo.npz and fusion_scene.npz, which
tests/golden/make_golden_fusion.py produced by running the reference's own functions in the build container (masks bit
for bit, depths and points to fp32 rounding).  One caveat, stated there too: dypcd_tanks samples the source depth with
cv2.remap, which is absent from the image; the golden run (and this restatement) use the grid_sample reprojection of
the same reference file instead, so cv2.remap's 1/32-pixel coordinate quantisation is the one unpinned detail."""
import numpy as np
import torch
import torch.nn.functional as F


def reproject_with_depth(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src):
    """pcd.py:151-200.  depth [H,W]; intrinsics [3,3]; extrinsics [4,4] (all float32 tensors)."""
    height, width = depth_ref.shape
    y_ref, x_ref = torch.meshgrid(torch.arange(0, height), torch.arange(0, width), indexing="ij")
    x_ref, y_ref = x_ref.reshape(-1), y_ref.reshape(-1)
    ones = torch.ones_like(x_ref)
    xyz_ref = torch.linalg.inv(intrinsics_ref) @ (torch.vstack((x_ref, y_ref, ones)) * depth_ref.reshape(-1))
    xyz_src = ((extrinsics_src @ torch.linalg.inv(extrinsics_ref)) @ torch.vstack((xyz_ref, ones.float())))[:3]
    K_xyz_src = intrinsics_src @ xyz_src
    xy_src = K_xyz_src[:2] / K_xyz_src[2:3]
    x_src = xy_src[0] / ((width - 1) / 2) - 1
    y_src = xy_src[1] / ((height - 1) / 2) - 1
    grid = torch.stack((x_src, y_src), dim=-1).view(1, height, width, 2)
    sampled = F.grid_sample(depth_src[None, None], grid, mode="bilinear", padding_mode="zeros", align_corners=True)[0, 0]
    xyz_src = torch.linalg.inv(intrinsics_src) @ (torch.vstack((xy_src, ones.float())) * sampled.reshape(-1))
    xyz_rep = ((extrinsics_ref @ torch.linalg.inv(extrinsics_src)) @ torch.vstack((xyz_src, ones.float())))[:3]
    depth_rep = xyz_rep[2].reshape(height, width)
    K_xyz_rep = intrinsics_ref @ xyz_rep
    z = K_xyz_rep[2:3]
    z = torch.where(z == 0, z + 0.00001, z)
    xy_rep = K_xyz_rep[:2] / z
    return depth_rep, xy_rep[0].reshape(height, width), xy_rep[1].reshape(height, width)


def check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src,
                                dist_thresh=1.0, rel_thresh=0.01):
    """pcd.py:224-242 -> (mask bool [H,W], depth_reprojected [H,W] with rejected pixels zeroed)."""
    height, width = depth_ref.shape
    y_ref, x_ref = torch.meshgrid(torch.arange(0, height), torch.arange(0, width), indexing="ij")
    depth_rep, x_rep, y_rep = reproject_with_depth(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src,
                                                   extrinsics_src)
    dist = torch.sqrt((x_rep - x_ref) ** 2 + (y_rep - y_ref) ** 2)
    dref = torch.where(depth_ref == 0, torch.full_like(depth_ref, 1e-4), depth_ref)
    rel = (depth_rep - dref).abs() / dref
    mask = (dist < dist_thresh) & (rel < rel_thresh)
    return mask, torch.where(mask, depth_rep, torch.zeros_like(depth_rep)), dist, rel


def _np(a):
    return a.numpy() if torch.is_tensor(a) else np.asarray(a)


def check_numpy(depth_ref, K_ref, E_ref, depth_src, K_src, E_src):
    """pcd.py:226-242 (the NumPy wrapper filter_depth calls): float64 distances on float32 reprojections.
    Mutates ``depth_ref`` like the reference (zeros -> 1e-4).  -> (mask, depth_reprojected)."""
    T = torch.from_numpy
    H, W = depth_ref.shape
    x_ref, y_ref = np.meshgrid(np.arange(0, W), np.arange(0, H))
    rep, xr, yr = reproject_with_depth(T(depth_ref.copy()), T(K_ref), T(E_ref), T(depth_src), T(K_src), T(E_src))
    rep, xr, yr = rep.numpy(), xr.numpy(), yr.numpy()
    dist = np.sqrt((xr - x_ref) ** 2 + (yr - y_ref) ** 2)
    depth_ref[depth_ref == 0] = 1e-4
    rel = np.abs(rep - depth_ref) / depth_ref
    mask = np.logical_and(dist < 1, rel < 0.01)
    rep[~mask] = 0
    return mask, rep


def check_ladder(depth_ref, K_ref, E_ref, depth_src, K_src, E_src, dist_base, rel_base):
    """dypcd_tanks.py:164-184: nine gates i = 2..10; the reprojected depth is zeroed outside the LAST one.
    No zero-depth patch (division by zero -> inf / nan -> False).  -> (masks [9,H,W], depth_reprojected)."""
    T = torch.from_numpy
    H, W = depth_ref.shape
    x_ref, y_ref = np.meshgrid(np.arange(0, W), np.arange(0, H))
    rep, xr, yr = reproject_with_depth(T(depth_ref.copy()), T(K_ref), T(E_ref), T(depth_src), T(K_src), T(E_src))
    rep, xr, yr = rep.numpy(), xr.numpy(), yr.numpy()
    dist = np.sqrt((xr - x_ref) ** 2 + (yr - y_ref) ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.abs(rep - depth_ref) / depth_ref
    masks = [np.logical_and(dist < i * dist_base, rel < i * rel_base) for i in range(2, 11)]
    rep[~masks[-1]] = 0
    return np.stack(masks), rep


def filter_view(ref_depth, ref_cam, confs3, conf_thresh, ref_img, src_depths, src_cams, thres_view=2, dynamic=False,
                dist_base=0.25, rel_base=1.0 / 1300):
    """One reference view of filter_depth (pcd.py:256-344; dypcd_tanks.py:198-304 when ``dynamic``).
    confs3 = (confidence1, confidence2, confidence) of the three stages; conf_thresh = args.conf.
    -> dict(photo, geo, final masks, depth_averaged, xyz [N,3] float32, rgb [N,3] uint8)."""
    K, E = ref_cam
    c1, c2, c3 = confs3
    photo = np.logical_and(np.logical_and(c3 > conf_thresh[2], c2 > conf_thresh[1]), c1 > conf_thresh[0])
    ref_depth = ref_depth.copy()
    reps, vote = [], 0
    level = [0] * (len(src_depths) - 1)
    for d_src, (Ks, Es) in zip(src_depths, src_cams):
        if dynamic:
            masks, rep = check_ladder(ref_depth, K, E, d_src, Ks, Es, dist_base, rel_base)
            m = masks[-1]
            for i in range(2, len(src_depths) + 1):
                level[i - 2] = level[i - 2] + masks[i - 2].astype(np.int32)
        else:
            m, rep = check_numpy(ref_depth, K, E, d_src, Ks, Es)   # patches ref_depth in place
        vote = vote + m.astype(np.int32)
        reps.append(rep)
    avg = (sum(reps) + ref_depth) / (vote + 1)
    if dynamic:
        geo = vote >= len(src_depths) + 1
        for i in range(2, len(src_depths) + 1):
            geo = np.logical_or(geo, level[i - 2] >= i)
    else:
        geo = vote >= thres_view
    final = np.logical_and(photo, geo)
    H, W = avg.shape
    x, y = np.meshgrid(np.arange(0, W), np.arange(0, H))
    x, y, depth = x[final], y[final], avg[final]
    xyz_ref = np.matmul(np.linalg.inv(K), np.vstack((x, y, np.ones_like(x))) * depth)
    xyz_world = np.matmul(np.linalg.inv(E), np.vstack((xyz_ref, np.ones_like(x))))[:3]
    return {"photo": photo, "geo": geo, "final": final, "depth_averaged": avg,
            "xyz": xyz_world.transpose((1, 0)).astype(np.float32), "rgb": (ref_img[final] * 255).astype(np.uint8)}
