"""ORACLE -- test infrastructure only.  CPU restatement of the fusion filter's geometric-consistency check
(/root/reference/filter/pcd.py:151-242), stepwise as the reference computes it (torch ops on CPU instead of
.cuda()).  The reference module itself cannot be imported here (cv2 / plyfile are absent), so this restatement
is NOT pinned against reference outputs ("parity unpinned" for row N4, stated in DESIGN.md)."""
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
