#!/usr/bin/env python3
"""Golden vectors for the fusion filter (row N4) by RUNNING THE REFERENCE's filter/pcd.py and filter/dypcd_tanks.py
(build container only; /root/reference is never copied).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_fusion.py

The reference modules import cv2 / plyfile / tomlkit / yacs, which this image lacks, and call ``.cuda()``.  None of
that is on the arithmetic path that is pinned here, so the generator registers EMPTY placeholder modules for those
names (they provide nothing: any use would raise), lets ``torch.Tensor.cuda`` return the tensor unchanged (the same
ATen ops then run on the CPU), restores the ``np.bool`` alias the reference still uses, and captures the vertex
array the reference hands to ``PlyElement.describe``.  Everything numeric below is computed by the reference's own
functions on dmvsnet_amd.synth.synth_fusion_scene:
  fusion_geo.npz      check_geometric_consistency_pytorch / check_geometric_consistency (pcd.py:203-242) per view pair;
                      dypcd_tanks.check_geometric_consistency (164-184): the nine threshold masks (on the module's
                      grid_sample reprojection, see below)
  fusion_scene.npz    filter_depth (pcd.py:244-361) and the dynamic variant (dypcd_tanks.py:186-326) run on a scene
                      folder on disk: the three masks of every reference view and the fused point cloud
"""
import argparse
import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from dmvsnet_amd import eval_io, synth  # noqa: E402

captured = {}


class _PlyElement:
    @staticmethod
    def describe(arr, name):
        captured["vertex"] = np.array(arr)
        return arr


class _PlyData:
    def __init__(self, els):
        pass

    def write(self, filename):
        captured["ply"] = filename


class _CN(dict):   # the few attribute-style lookups filter/tank_test_config.py performs at import time
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


for name in ("cv2", "tomlkit", "plyfile", "yacs", "yacs.config"):
    sys.modules[name] = types.ModuleType(name)
# datasets/__init__.py pulls in the training loaders (torchvision ...): expose the package directory without running it,
# the only thing the filter needs from it is the real datasets/data_io.py
sys.modules["datasets"] = types.ModuleType("datasets")
sys.modules["datasets"].__path__ = ["/root/reference/datasets"]
sys.modules["tomlkit"].value = None
sys.modules["plyfile"].PlyData, sys.modules["plyfile"].PlyElement = _PlyData, _PlyElement
sys.modules["yacs.config"].CfgNode = _CN
torch.Tensor.cuda = lambda self, *a, **k: self
np.bool = bool   # removed from NumPy 1.24+, still used by the reference's save_mask

with contextlib.redirect_stdout(io.StringIO()):
    from filter import dypcd_tanks as ref_dy  # noqa: E402
    from filter import pcd as ref_pcd  # noqa: E402

# dypcd_tanks.check_geometric_consistency calls the module's cv2.remap reprojection (dypcd_tanks.py:62-99); cv2 is absent,
# so it is pointed at the SAME module's own grid_sample reprojection (reproject_with_depth_pytorch, :101-163 -- the
# one filter/pcd.py uses).  The threshold ladder, the dynamic voting rule and the averaging that are pinned below are
# the reference's code either way; what stays unpinned is cv2.remap's 1/32-pixel coordinate quantisation.
ref_dy.reproject_with_depth = ref_dy.reproject_with_depth_pytorch

H, W, V = 96, 128, 5
cams, depths, confs, imgs = synth.synth_fusion_scene(H, W, V, seed=0)
K = lambda v: cams[v, 1, :3, :3].copy()   # noqa: E731
E = lambda v: cams[v, 0].copy()           # noqa: E731

out = {}
args_dy = argparse.Namespace(dist_base=1 / 4, rel_diff_base=1 / 1300)
for v in range(1, V):
    m, rep, xs, ys = ref_pcd.check_geometric_consistency_pytorch(
        torch.from_numpy(depths[0].copy()), torch.from_numpy(K(0)), torch.from_numpy(E(0)),
        torch.from_numpy(depths[v].copy()), torch.from_numpy(K(v)), torch.from_numpy(E(v)))
    out[f"torch.mask.{v}"], out[f"torch.rep.{v}"] = m.numpy(), rep.numpy()
    m2, rep2, _, _ = ref_pcd.check_geometric_consistency(depths[0].copy(), K(0), E(0), depths[v].copy(), K(v), E(v))
    out[f"np.mask.{v}"], out[f"np.rep.{v}"] = m2, rep2
    masks, last, rep3, _, _ = ref_dy.check_geometric_consistency(args_dy, depths[0].copy(), K(0), E(0), depths[v].copy(), K(v), E(v))
    out[f"dy.masks.{v}"], out[f"dy.rep.{v}"] = np.stack(masks), rep3
np.savez_compressed(os.path.join(HERE, "fusion_geo.npz"), **out)
print("fusion_geo.npz", {k: v.shape for k, v in list(out.items())[:4]})

# ---- the file-based drivers on a scene folder
pairs = [(0, [1, 2, 3, 4]), (1, [0, 2, 3]), (2, [1, 3, 0, 4])]
scene = {}
with tempfile.TemporaryDirectory() as tmp:
    for sub in ("cams", "images", "depth_est", "confidence", "pcd"):
        os.makedirs(os.path.join(tmp, sub))
    from PIL import Image
    for v in range(V):
        eval_io.write_cam(os.path.join(tmp, "cams/{:0>8}_cam.txt".format(v)), np.stack((E(v), np.pad(K(v), ((0, 1), (0, 1))))))
        Image.fromarray((imgs[v] * 255).astype(np.uint8)).save(os.path.join(tmp, "images/{:0>8}.jpg".format(v)), quality=95)
        eval_io.save_pfm(os.path.join(tmp, "depth_est/{:0>8}.pfm".format(v)), depths[v])
        eval_io.save_pfm(os.path.join(tmp, "confidence/{:0>8}_stage1.pfm".format(v)), confs[v][0])
        eval_io.save_pfm(os.path.join(tmp, "confidence/{:0>8}_stage2.pfm".format(v)), confs[v][1])
        eval_io.save_pfm(os.path.join(tmp, "confidence/{:0>8}.pfm".format(v)), confs[v][2])
    with open(os.path.join(tmp, "pair.txt"), "w") as f:
        f.write(f"{len(pairs)}\n")
        for r, srcs in pairs:
            f.write(f"{r}\n{len(srcs)} " + " ".join(f"{s} 1.0" for s in srcs) + "\n")
    args = argparse.Namespace(ndepths=[64, 32, 8], conf=[0.1, 0.2, 0.3], thres_view=2, display=False,
                              dist_base=1 / 4, rel_diff_base=1 / 1300)

    def masks_of(tag):
        for r, _ in pairs:
            for kind in ("photo", "geo", "final"):
                scene[f"{tag}.mask.{r}.{kind}"] = np.array(Image.open(os.path.join(tmp, "mask/{:0>8}_{}.png".format(r, kind)))) > 0

    with contextlib.redirect_stdout(io.StringIO()):
        ref_pcd.filter_depth(args, tmp, tmp, tmp, os.path.join(tmp, "pcd/a.ply"))
    scene["pcd.vertex"] = captured["vertex"]
    masks_of("pcd")
    for r, _ in pairs:   # the dynamic driver skips views whose geo mask + averaged depth already exist
        os.remove(os.path.join(tmp, "mask/{:0>8}_geo.png".format(r)))
    with contextlib.redirect_stdout(io.StringIO()):
        ref_dy.filter_depth(args, tmp, tmp, tmp, os.path.join(tmp, "pcd/b.ply"))
    scene["dy.vertex"] = captured["vertex"]
    masks_of("dy")
    # the jpg round trip of the colours is part of the reference's pipeline: keep the decoded images as inputs
    for v in range(V):
        scene[f"img.{v}"] = np.array(Image.open(os.path.join(tmp, "images/{:0>8}.jpg".format(v))))
scene["pairs"] = np.array([[r] + s + [-1] * (4 - len(s)) for r, s in pairs])
np.savez_compressed(os.path.join(HERE, "fusion_scene.npz"), **scene)
print("fusion_scene.npz", len(scene["pcd.vertex"]), "points (static),", len(scene["dy.vertex"]), "points (dynamic)")
