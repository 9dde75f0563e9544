"""Depth-map fusion after the network: photometric mask + geometric consistency over
the source views + back-projection to a coloured point cloud.

Mirrors /root/reference/filter/pcd.py: ``check_geometric_consistency`` (151-242, one fused HIP kernel here,
``dmvs_geo_consistency``), ``filter_depth`` (244-361) and the PLY output.  The per-pixel work runs on the GPU; file
handling stays in Python.  cv2 / plyfile are not needed (masks are not written as PNG; the PLY is written directly).
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .eval_io import read_pfm


def fold_projection(intrinsics_ref, extrinsics_ref, intrinsics_src, extrinsics_src) -> np.ndarray:
    """The 33 floats ``dmvs_geo_consistency`` takes (fp64 products, rounded once); see include/dmvs.h."""
    Kr, Er = np.asarray(intrinsics_ref, np.float64), np.asarray(extrinsics_ref, np.float64)
    Ks, Es = np.asarray(intrinsics_src, np.float64), np.asarray(extrinsics_src, np.float64)
    rel = Es @ np.linalg.inv(Er)      # ref camera -> src camera
    back = Er @ np.linalg.inv(Es)     # src camera -> ref camera
    A1 = Ks @ rel[:3, :3] @ np.linalg.inv(Kr)
    b1 = Ks @ rel[:3, 3]
    A2 = back[:3, :3] @ np.linalg.inv(Ks)
    t2 = back[:3, 3]
    return np.concatenate([A1.ravel(), b1, A2.ravel(), t2, Kr.ravel()]).astype(np.float32)


def check_geometric_consistency(depth_ref: torch.Tensor, intrinsics_ref, extrinsics_ref, depth_src: torch.Tensor,
                                intrinsics_src, extrinsics_src, dist_thresh: float = 1.0, rel_thresh: float = 0.01,
                                vote_sum: torch.Tensor = None, depth_sum: torch.Tensor = None):
    """-> (mask uint8 [H,W], depth_reprojected [H,W]); optionally accumulates into vote_sum (int32) / depth_sum."""
    if not depth_ref.is_cuda:
        raise _lib.DmvsError("fusion kernels need tensors on a HIP device (no CPU fallback)")
    H, W = depth_ref.shape
    P = torch.from_numpy(fold_projection(intrinsics_ref, extrinsics_ref, intrinsics_src, extrinsics_src)).to(depth_ref.device)
    mask = torch.empty((H, W), dtype=torch.uint8, device=depth_ref.device)
    rep = torch.empty((H, W), dtype=torch.float32, device=depth_ref.device)
    p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    code = _lib.load().dmvs_geo_consistency(p(depth_ref.contiguous()), p(depth_src.contiguous()), p(P), H, W,
                                            float(dist_thresh), float(rel_thresh), p(mask), p(rep), p(vote_sum),
                                            p(depth_sum), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(code, "dmvs_geo_consistency")
    return mask, rep


def filter_depth(ref_depth, ref_conf, ref_cam, ref_img, src_depths: Sequence, src_cams: Sequence, conf_thresh=0.1,
                 thres_view: int = 2, device="cuda") -> Tuple[np.ndarray, np.ndarray, Dict[str, float]]:
    """One reference view of filter_depth (pcd.py:256-335).  cams are (intrinsics [3,3], extrinsics [4,4]) pairs.
    Returns (xyz_world [N,3] float32, rgb [N,3] uint8, mask statistics)."""
    Kr, Er = ref_cam
    d_ref = torch.from_numpy(np.ascontiguousarray(ref_depth, dtype=np.float32)).to(device)
    conf = torch.from_numpy(np.ascontiguousarray(ref_conf, dtype=np.float32)).to(device)
    photo_mask = conf > conf_thresh
    votes = torch.zeros(d_ref.shape, dtype=torch.int32, device=device)
    dsum = torch.zeros_like(d_ref)
    for d_src, (Ks, Es) in zip(src_depths, src_cams):
        d_src = torch.from_numpy(np.ascontiguousarray(d_src, dtype=np.float32)).to(device)
        check_geometric_consistency(d_ref, Kr, Er, d_src, Ks, Es, vote_sum=votes, depth_sum=dsum)
    # the reference overwrites zero reference depths with 1e-4 before averaging (pcd.py:219 mutates in place)
    d_avg = (dsum + torch.where(d_ref == 0, torch.full_like(d_ref, 1e-4), d_ref)) / (votes + 1).float()
    geo_mask = votes >= thres_view
    final = photo_mask & geo_mask
    ys, xs = torch.nonzero(final, as_tuple=True)
    depth = d_avg[final].double()
    pts = torch.linalg.inv(torch.from_numpy(np.asarray(Kr, np.float64)).to(device)) @ (
        torch.stack((xs.double(), ys.double(), torch.ones_like(depth))) * depth)
    pts = (torch.linalg.inv(torch.from_numpy(np.asarray(Er, np.float64)).to(device)) @ torch.cat((pts, torch.ones_like(depth)[None])))[:3]
    rgb = (np.asarray(ref_img)[final.cpu().numpy()] * 255).astype(np.uint8)
    stats = {"photo": photo_mask.float().mean().item(), "geo": geo_mask.float().mean().item(),
             "final": final.float().mean().item()}
    return pts.T.float().cpu().numpy(), rgb, stats


def write_ply(filename: str, xyz: np.ndarray, rgb: np.ndarray) -> None:
    """Binary little-endian PLY with x,y,z float32 + red,green,blue uint8 vertices (what PlyData writes, pcd.py:346-360)."""
    v = np.empty(len(xyz), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    v["x"], v["y"], v["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    v["red"], v["green"], v["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    with open(filename, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                 "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % len(v)).encode())
        f.write(v.tobytes())


def read_camera_parameters(filename):
    """(intrinsics [3,3], extrinsics [4,4]) from a *_cam.txt as written by eval_io.write_cam (pcd.py:33-44)."""
    with open(filename) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    return intrinsics, extrinsics


def pcd_filter_scene(pair_data: Sequence[Tuple[int, List[int]]], out_folder: str, plyfilename: str, conf_thresh=0.1,
                     thres_view=2, device="cuda") -> Dict[str, float]:
    """filter_depth over a scene folder written by eval_io.save_depth_maps (depth_est/, confidence/, cams/, images/)."""
    from PIL import Image
    load = lambda v: (read_pfm(os.path.join(out_folder, "depth_est/{:0>8}.pfm".format(v)))[0],
                      read_camera_parameters(os.path.join(out_folder, "cams/{:0>8}_cam.txt".format(v))))
    pts, cols, stats = [], [], {}
    for ref_view, src_views in pair_data:
        d_ref, cam_ref = load(ref_view)
        conf = read_pfm(os.path.join(out_folder, "confidence/{:0>8}.pfm".format(ref_view)))[0]
        img = np.array(Image.open(os.path.join(out_folder, "images/{:0>8}.jpg".format(ref_view))), dtype=np.float32) / 255.0
        srcs = [load(v) for v in src_views]
        xyz, rgb, stats = filter_depth(d_ref, conf, cam_ref, img, [s[0] for s in srcs], [s[1] for s in srcs], conf_thresh,
                                       thres_view, device)
        pts.append(xyz)
        cols.append(rgb)
    write_ply(plyfilename, np.concatenate(pts), np.concatenate(cols))
    return stats
