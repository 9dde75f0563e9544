"""Depth-map fusion after the network: photometric mask, geometric consistency over
the source views, depth averaging and back-projection to a coloured point cloud.

What the reference does per reference view (/root/reference/filter/pcd.py:244-361 ``filter_depth``; the
Tanks&Temples variant with a ladder of thresholds, filter/dypcd_tanks.py:164-326) becomes:

  ``ViewFilter``         the per-view state on the GPU: photometric mask from the three stage confidences, vote /
                         depth accumulators, one fused kernel launch per source view (``dmvs_geo_consistency`` or
                         ``dmvs_geo_consistency_ladder``), then the static (``votes >= thres_view``) or dynamic
                         (``any_i votes_i >= i``) geometric mask, the averaged depth and the world points
  ``fuse_scene``         the file-based driver over a folder written by ``eval_io.save_depth_maps`` (depth_est/,
                         confidence/, cams/, images/): mask PNGs, the averaged depth PFM of the dynamic variant, the PLY

The per-pixel work runs on the GPU (no CPU fallback); file handling stays in Python.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .eval_io import read_pfm, save_pfm

N_LEVELS = 9   # gates i = 2..10 of the dynamic variant (dypcd_tanks.py:178)


def fold_projection(intrinsics_ref, extrinsics_ref, intrinsics_src, extrinsics_src) -> np.ndarray:
    """The 33 floats the consistency kernels take (fp64 products, rounded once); see include/dmvs.h."""
    Kr, Er = np.asarray(intrinsics_ref, np.float64), np.asarray(extrinsics_ref, np.float64)
    Ks, Es = np.asarray(intrinsics_src, np.float64), np.asarray(extrinsics_src, np.float64)
    rel = Es @ np.linalg.inv(Er)      # ref camera -> src camera
    back = Er @ np.linalg.inv(Es)     # src camera -> ref camera
    A1 = Ks @ rel[:3, :3] @ np.linalg.inv(Kr)
    b1 = Ks @ rel[:3, 3]
    A2 = back[:3, :3] @ np.linalg.inv(Ks)
    t2 = back[:3, 3]
    return np.concatenate([A1.ravel(), b1, A2.ravel(), t2, Kr.ravel()]).astype(np.float32)


def _dev_f32(a, device, what):
    t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    t = t.to(device=device, dtype=torch.float32).contiguous()
    if t.dim() != 2:
        raise _lib.DmvsError(f"{what}: expected an [H,W] map, got {tuple(t.shape)}")
    return t


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def check_geometric_consistency(depth_ref: torch.Tensor, intrinsics_ref, extrinsics_ref, depth_src: torch.Tensor,
                                intrinsics_src, extrinsics_src, dist_thresh: float = 1.0, rel_thresh: float = 0.01,
                                vote_sum: Optional[torch.Tensor] = None, depth_sum: Optional[torch.Tensor] = None,
                                level_votes: Optional[torch.Tensor] = None):
    """One (reference, source) pair, pcd.py:151-242 -> (mask uint8 [H,W], depth_reprojected [H,W]).
    ``vote_sum`` (int32 [H,W]) / ``depth_sum`` (float32 [H,W]) are accumulated when given.
    ``level_votes`` (int32 [9,H,W]) selects the dynamic ladder (dypcd_tanks.py:164-184): ``dist_thresh`` /
    ``rel_thresh`` are then the BASES of the nine gates and mask / sums refer to the last one."""
    if not depth_ref.is_cuda:
        raise _lib.DmvsError("fusion kernels need tensors on a HIP device (no CPU fallback)")
    H, W = depth_ref.shape
    for name, t, dt, shape in (("depth_ref", depth_ref, torch.float32, (H, W)), ("depth_src", depth_src, torch.float32, (H, W)),
                               ("vote_sum", vote_sum, torch.int32, (H, W)), ("depth_sum", depth_sum, torch.float32, (H, W)),
                               ("level_votes", level_votes, torch.int32, (N_LEVELS, H, W))):
        if t is None:
            continue
        # the kernel indexes every map with the reference's H, W: a smaller source map would be read out of bounds
        if tuple(t.shape) != shape or t.dtype != dt or not t.is_contiguous() or t.device != depth_ref.device:
            raise _lib.DmvsError(f"{name}: need a contiguous {dt} tensor of shape {shape} on {depth_ref.device}, got "
                                 f"{t.dtype} {tuple(t.shape)} on {t.device}")
    P = torch.from_numpy(fold_projection(intrinsics_ref, extrinsics_ref, intrinsics_src, extrinsics_src)).to(depth_ref.device)
    mask = torch.empty((H, W), dtype=torch.uint8, device=depth_ref.device)
    rep = torch.empty((H, W), dtype=torch.float32, device=depth_ref.device)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib = _lib.load()
    if level_votes is None:
        code = lib.dmvs_geo_consistency(_ptr(depth_ref), _ptr(depth_src), _ptr(P), H, W, float(dist_thresh),
                                        float(rel_thresh), _ptr(mask), _ptr(rep), _ptr(vote_sum), _ptr(depth_sum), st)
    else:
        code = lib.dmvs_geo_consistency_ladder(_ptr(depth_ref), _ptr(depth_src), _ptr(P), H, W, float(dist_thresh),
                                               float(rel_thresh), _ptr(level_votes), _ptr(mask), _ptr(rep),
                                               _ptr(vote_sum), _ptr(depth_sum), st)
    _lib.check(code, "dmvs_geo_consistency")
    return mask, rep


@dataclass
class FusedView:
    xyz: np.ndarray            # [N,3] float32 world points
    rgb: np.ndarray            # [N,3] uint8
    photo_mask: np.ndarray     # [H,W] bool
    geo_mask: np.ndarray
    final_mask: np.ndarray
    depth_averaged: np.ndarray  # [H,W] float32
    stats: Dict[str, float]


class ViewFilter:
    """Fusion state of ONE reference view on the GPU.  ``conf`` thresholds are per stage ``(c1, c2, c3)`` like
    ``args.conf`` (pcd.py:268-274): the mask is ``conf3 > c3 & conf2 > c2 & conf1 > c1``; a scalar applies one
    threshold to the final confidence only; missing stage maps default to the final one, as in the reference."""

    def __init__(self, depth, cam, confidence, conf=(0.1, 0.1, 0.1), confidence2=None, confidence1=None,
                 dynamic: bool = False, device="cuda"):
        self.device = torch.device(device)
        self.K, self.E = np.asarray(cam[0]), np.asarray(cam[1])
        self.depth = _dev_f32(depth, self.device, "depth")
        c3 = _dev_f32(confidence, self.device, "confidence")
        c2 = c3 if confidence2 is None else _dev_f32(confidence2, self.device, "confidence2")
        c1 = c3 if confidence1 is None else _dev_f32(confidence1, self.device, "confidence1")
        if np.isscalar(conf):
            self.photo_mask = c3 > float(conf)
        else:
            t1, t2, t3 = (float(v) for v in conf)
            self.photo_mask = (c3 > t3) & (c2 > t2) & (c1 > t1)
        self.dynamic = dynamic
        self.votes = torch.zeros(self.depth.shape, dtype=torch.int32, device=self.device)
        self.depth_sum = torch.zeros_like(self.depth)
        self.level_votes = torch.zeros((N_LEVELS,) + tuple(self.depth.shape), dtype=torch.int32, device=self.device) if dynamic else None
        self.nsrc = 0

    def add_source(self, depth_src, cam_src, dist=None, rel=None):
        """Accumulate one source view.  Static: gates ``dist`` px / ``rel`` (default 1 / 0.01, pcd.py:236);
        dynamic: bases of the ladder (default 1/4 px, 1/1300; the tank scripts' ``--dist_base`` / ``--rel_diff_base``)."""
        d_src = _dev_f32(depth_src, self.device, "depth_src")
        if self.dynamic:
            dist, rel = (0.25 if dist is None else dist), (1.0 / 1300 if rel is None else rel)
        else:
            dist, rel = (1.0 if dist is None else dist), (0.01 if rel is None else rel)
        check_geometric_consistency(self.depth, self.K, self.E, d_src, cam_src[0], cam_src[1], dist, rel,
                                    vote_sum=self.votes, depth_sum=self.depth_sum, level_votes=self.level_votes)
        self.nsrc += 1

    def finish(self, image, thres_view: int = 2) -> FusedView:
        """Masks, averaged depth and the world points of the view (pcd.py:299-344 / dypcd_tanks.py:258-304).
        ``image`` [H,W,3] float in [0,1] supplies the colours."""
        d_ref = self.depth
        if self.dynamic:
            # at least i views under gate i, for any i in [2, nsrc]; the all-views test on the last gate is
            # `votes >= nsrc + 1` in the reference and can never hold (dypcd_tanks.py:233,258-260)
            geo = self.votes >= self.nsrc + 1
            for i in range(2, self.nsrc + 1):
                geo |= self.level_votes[i - 2] >= i
            d_base = d_ref
        else:
            geo = self.votes >= thres_view
            # the reference's check overwrites zero reference depths with 1e-4 IN PLACE before averaging (pcd.py:235)
            d_base = torch.where(d_ref == 0, torch.full_like(d_ref, 1e-4), d_ref) if self.nsrc else d_ref
        # float32 sum, promoted to float64 only by the division with the int32 count (NumPy: pcd.py:299, dypcd_tanks.py:248)
        d_avg = (self.depth_sum + d_base).double() / (self.votes + 1).double()
        final = self.photo_mask & geo
        ys, xs = torch.nonzero(final, as_tuple=True)
        depth = d_avg[final]
        Kinv = torch.from_numpy(np.linalg.inv(np.asarray(self.K, np.float32)).astype(np.float64)).to(self.device)
        Einv = torch.from_numpy(np.linalg.inv(np.asarray(self.E, np.float32)).astype(np.float64)).to(self.device)
        pts = Kinv @ (torch.stack((xs.double(), ys.double(), torch.ones_like(depth))) * depth)
        pts = (Einv @ torch.cat((pts, torch.ones_like(depth)[None])))[:3]
        fm = final.cpu().numpy()
        rgb = (np.asarray(image)[fm] * 255).astype(np.uint8)
        stats = {"photo": self.photo_mask.float().mean().item(), "geo": geo.float().mean().item(),
                 "final": final.float().mean().item()}
        return FusedView(pts.T.float().cpu().numpy(), rgb, self.photo_mask.cpu().numpy(), geo.cpu().numpy(), fm,
                         d_avg.float().cpu().numpy(), stats)


def filter_depth(ref_depth, ref_conf, ref_cam, ref_img, src_depths: Sequence, src_cams: Sequence, conf_thresh=0.1,
                 thres_view: int = 2, device="cuda", confidence2=None, confidence1=None, dynamic: bool = False
                 ) -> Tuple[np.ndarray, np.ndarray, Dict[str, float]]:
    """One reference view (pcd.py:256-344): (xyz_world [N,3] float32, rgb [N,3] uint8, mask statistics).
    ``conf_thresh``: scalar or the per-stage triple of ``args.conf``."""
    vf = ViewFilter(ref_depth, ref_cam, ref_conf, conf_thresh, confidence2, confidence1, dynamic, device)
    for d_src, cam in zip(src_depths, src_cams):
        vf.add_source(d_src, cam)
    out = vf.finish(ref_img, thres_view)
    return out.xyz, out.rgb, out.stats


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
    """(intrinsics [3,3], extrinsics [4,4]) from a *_cam.txt as written by eval_io.write_cam (pcd.py:68-78)."""
    with open(filename) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    return intrinsics, extrinsics


def read_pair_file(filename) -> List[Tuple[int, List[int]]]:
    """pair.txt -> [(ref_view, [src_view, ...]), ...]; views without sources are skipped (pcd.py:82-93)."""
    data = []
    with open(filename) as f:
        for _ in range(int(f.readline())):
            ref_view = int(f.readline().rstrip())
            src_views = [int(x) for x in f.readline().rstrip().split()[1::2]]
            if src_views:
                data.append((ref_view, src_views))
    return data


def fuse_scene(pair_data: Sequence[Tuple[int, List[int]]], out_folder: str, plyfilename: str, conf=(0.1, 0.15, 0.7),
               thres_view: int = 5, dynamic: bool = False, num_stage: int = 3, device="cuda",
               write_masks: bool = True, dist_base: Optional[float] = None, rel_diff_base: Optional[float] = None) -> Dict[str, float]:
    """``filter_depth`` over a scene folder written by eval_io.save_depth_maps (depth_est/, confidence/ incl. the
    optional ``_stage1`` / ``_stage2`` maps, cams/, images/): writes mask/%08d_{photo,geo,final}.png
    (pcd.py:307-310), depth_est/%08d_averaged.pfm for the dynamic variant (dypcd_tanks.py:255) and the PLY.
    Defaults = main.py:60-61 (``--conf 0.1 0.15 0.7``, ``--thres_view 5``); ``dist_base`` / ``rel_diff_base``: the dynamic
    filter's ladder bases (main.py:63-64), ignored by the static one.  Views of one scene must share one size (the
    reference's filters assume it too: they index the source maps with reference-sized grids); a scene with mixed
    image sizes needs ``fix_res`` in step 1."""
    from PIL import Image

    def pfm(sub, v, suffix=""):
        return read_pfm(os.path.join(out_folder, sub, "{:0>8}{}.pfm".format(v, suffix)))[0]

    def cam(v):
        return read_camera_parameters(os.path.join(out_folder, "cams/{:0>8}_cam.txt".format(v)))

    pts, cols, stats = [], [], {}
    for ref_view, src_views in pair_data:
        has_stages = os.path.exists(os.path.join(out_folder, "confidence/{:0>8}_stage2.pfm".format(ref_view)))
        vf = ViewFilter(pfm("depth_est", ref_view), cam(ref_view), pfm("confidence", ref_view), conf,
                        pfm("confidence", ref_view, "_stage2") if has_stages else None,
                        pfm("confidence", ref_view, "_stage1") if has_stages else None, dynamic, device)
        for v in src_views:
            if dynamic:
                vf.add_source(pfm("depth_est", v), cam(v), dist_base, rel_diff_base)
            else:
                vf.add_source(pfm("depth_est", v), cam(v))
        img = np.array(Image.open(os.path.join(out_folder, "images/{:0>8}.jpg".format(ref_view))), dtype=np.float32) / 255.0
        step = 2 ** (3 - num_stage)     # 1- / 2-stage nets stop at 1/4 / 1/2 resolution (pcd.py:332-337)
        out = vf.finish(img[1::step, 1::step] if step > 1 else img, thres_view)
        if write_masks:
            os.makedirs(os.path.join(out_folder, "mask"), exist_ok=True)
            for kind, m in (("photo", out.photo_mask), ("geo", out.geo_mask), ("final", out.final_mask)):
                Image.fromarray(m.astype(np.uint8) * 255).save(os.path.join(out_folder, "mask/{:0>8}_{}.png".format(ref_view, kind)))
        if dynamic:
            save_pfm(os.path.join(out_folder, "depth_est/{:0>8}_averaged.pfm".format(ref_view)), out.depth_averaged)
        pts.append(out.xyz)
        cols.append(out.rgb)
        stats = out.stats
    write_ply(plyfilename, np.concatenate(pts), np.concatenate(cols))
    return stats
