"""Eval-side I/O either side of the network boundary.

Mirrors, with the same names and argument meaning, what the reference's eval path uses:
  read_pfm / save_pfm          /root/reference/datasets/data_io.py:6-71
  write_cam                    /root/reference/tools.py:40-57
  MVSDataset (mode "test")     /root/reference/datasets/general_eval.py:9-203
  save_depth_maps              step 1 of Model.test, /root/reference/model.py:323-380

The reference resizes with ``cv2.resize`` (INTER_LINEAR on float32 images); cv2 is absent from this image, so
``resize_linear`` restates OpenCV's published algorithm (half-pixel centres, no anti-aliasing, separable fp32 passes)
and is checked against hand-computed vectors and an independent NumPy restatement in the oracle -- not against cv2
output (that one comparison stays unpinned).  Inputs whose size already is a multiple of 32 within (max_h, max_w) take
the identity branch, which is exact.
"""
from __future__ import annotations

import os
import re
import sys
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

GLOBAL_BASE = 32  # general_eval.py:7


# ------------------------------------------------------------------------------------------ PFM
def read_pfm(filename):
    """-> (float32 array [H,W] or [H,W,3] with row 0 = top, scale).  data_io.py:6-42."""
    with open(filename, "rb") as f:
        header = f.readline().decode("utf-8").rstrip()
        if header not in ("PF", "Pf"):
            raise Exception("Not a PFM file.")
        m = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not m:
            raise Exception("Malformed PFM header.")
        width, height = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.frombuffer(f.read(), dtype=endian + "f4")
    shape = (height, width, 3) if header == "PF" else (height, width)
    return np.flipud(data.reshape(shape)), abs(scale)


def save_pfm(filename, image, scale=1):
    """float32 [H,W] / [H,W,1] / [H,W,3]; rows stored bottom-up, little-endian => negative scale.  data_io.py:45-71."""
    image = np.flipud(np.asarray(image))
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    if image.ndim == 3 and image.shape[2] == 3:
        color = True
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        color = False
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    endian = image.dtype.byteorder
    if endian == "<" or (endian == "=" and sys.byteorder == "little"):
        scale = -scale
    with open(filename, "wb") as f:
        f.write(b"PF\n" if color else b"Pf\n")
        f.write("{} {}\n".format(image.shape[1], image.shape[0]).encode("utf-8"))
        f.write(("%f\n" % scale).encode("utf-8"))
        f.write(np.ascontiguousarray(image).tobytes())


def write_cam(file, cam):
    """cam [2,4,4] (extrinsic; intrinsic in [1,:3,:3], depth range in [1,3,:]).  tools.py:40-57."""
    with open(file, "w") as f:
        f.write("extrinsic\n")
        for i in range(4):
            f.write(" ".join(str(cam[0][i][j]) for j in range(4)) + " \n")
        f.write("\nintrinsic\n")
        for i in range(3):
            f.write(" ".join(str(cam[1][i][j]) for j in range(3)) + " \n")
        f.write("\n" + " ".join(str(cam[1][3][j]) for j in range(4)) + "\n")


# ------------------------------------------------------------------------------------------ dataset
def resize_linear(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """``cv2.resize(img, (new_w, new_h))`` (INTER_LINEAR, the default) for float32 images, restated from OpenCV's
    published algorithm (resizeGeneric / HResizeLinear + VResizeLinear): half-pixel centres
    ``s = (d + 0.5) * (src / dst) - 0.5``, no anti-aliasing, edge replication, a horizontal fp32 pass followed by a
    vertical one.  cv2 itself is absent from the image: checked against hand-computed vectors and the oracle's
    independent NumPy restatement (tests/test_eval_io.py), not against cv2 output."""
    h, w = img.shape[:2]
    if (h, w) == (new_h, new_w):
        return img

    def taps(n_out, n_in):
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        f = f - i0.astype(np.float32)
        lo, hi = i0 < 0, i0 >= n_in - 1
        f[lo | hi] = 0.0
        i0 = np.clip(i0, 0, n_in - 1)
        return torch.from_numpy(i0), torch.from_numpy(np.minimum(i0 + 1, n_in - 1)), torch.from_numpy(f)

    t = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))
    squeeze = t.dim() == 2
    if squeeze:
        t = t[..., None]
    x0, x1, fx = taps(new_w, w)
    rows = t[:, x0] * (1.0 - fx)[None, :, None] + t[:, x1] * fx[None, :, None]
    y0, y1, fy = taps(new_h, h)
    out = rows[y0] * (1.0 - fy)[:, None, None] + rows[y1] * fy[:, None, None]
    return (out[..., 0] if squeeze else out).contiguous().numpy()


@dataclass
class CamFile:
    """One ``%08d_cam.txt``: extrinsic 4x4, intrinsic 3x3 (as stored: full image resolution) and the depth line
    ``depth_min depth_interval [num_depth [depth_max]]``."""
    extrinsics: np.ndarray
    intrinsics: np.ndarray
    depth_min: float
    depth_interval: float
    num_depth: Optional[int] = None

    @staticmethod
    def parse(filename) -> "CamFile":
        with open(filename) as f:
            lines = [line.rstrip() for line in f.readlines()]
        extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
        intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
        parts = lines[11].split()
        return CamFile(extrinsics, intrinsics, float(parts[0]), float(parts[1]),
                       int(float(parts[2])) if len(parts) >= 3 else None)

    def for_network(self, ndepths: int, interval_scale: float):
        """(stage-1 intrinsics = K/4, extrinsics, depth_min, depth interval) as the loader hands them on
        (general_eval.py:57-80: a three-field depth line spreads its range over ``ndepths`` planes)."""
        K = self.intrinsics.copy()
        K[:2, :] /= 4.0
        interval = self.depth_interval
        if self.num_depth is not None:
            interval = (self.depth_min + self.num_depth * interval - self.depth_min) / ndepths
        return K, self.extrinsics.copy(), self.depth_min, interval * interval_scale


@dataclass
class ResizePolicy:
    """Image sizes the network accepts (general_eval.py:97-110): both sides multiples of ``base`` (32), inside
    ``max_h x max_w`` (shrunk preserving the aspect ratio if larger), rounded DOWN."""
    max_h: int
    max_w: int
    base: int = 32

    def target(self, h: int, w: int):
        if h > self.max_h or w > self.max_w:
            scale = 1.0 * self.max_h / h
            if scale * w > self.max_w:
                scale = 1.0 * self.max_w / w
            new_w, new_h = scale * w // self.base * self.base, scale * h // self.base * self.base
        else:
            new_w, new_h = 1.0 * w // self.base * self.base, 1.0 * h // self.base * self.base
        return int(new_h), int(new_w)

    def apply(self, img: np.ndarray, intrinsics: np.ndarray, size=None):
        """Resize ``img`` to ``size`` (default: the policy's target) and scale the intrinsics' rows with it."""
        h, w = img.shape[:2]
        new_h, new_w = self.target(h, w) if size is None else size
        K = intrinsics.copy()
        K[0, :] *= 1.0 * new_w / w
        K[1, :] *= 1.0 * new_h / h
        return resize_linear(img, new_h, new_w), K


def read_pairs(filename, nviews: Optional[int] = None):
    """pair.txt -> [(ref_view, [src views, best first])]; views without sources are dropped; with ``nviews`` short
    lists are padded with the best source view (general_eval.py:38-47)."""
    out = []
    with open(filename) as f:
        for _ in range(int(f.readline())):
            ref_view = int(f.readline().rstrip())
            src_views = [int(x) for x in f.readline().rstrip().split()[1::2]]
            if not src_views:
                continue
            if nviews is not None and len(src_views) < nviews - 1:
                src_views += [src_views[0]] * (nviews - len(src_views))
            out.append((ref_view, src_views))
    return out


def depth_hypothesis_values(depth_min: float, depth_interval: float, ndepths: int, inverse_depth: bool) -> np.ndarray:
    """The ``depth_values`` vector of a sample (general_eval.py:178-184); the network only reads its ends and length."""
    if inverse_depth:
        depth_end = depth_interval * ndepths + depth_min
        return (1.0 / np.linspace(1.0 / depth_min, 1.0 / depth_end, ndepths, endpoint=False)).astype(np.float32)
    return np.arange(depth_min, depth_interval * (ndepths - 0.5) + depth_min, depth_interval, dtype=np.float32)


class MVSDataset(torch.utils.data.Dataset):
    """Eval loader for DTU / Tanks&Temples style scenes (``<scan>/pair.txt``, ``<scan>/cams/%08d_cam.txt``,
    ``<scan>/images(_post)/%08d.jpg``) with the constructor arguments and the sample dict of the reference's test-mode
    loader (general_eval.py:9-203): imgs [V,3,H,W] in [0,1]; proj_matrices {"stage1|2|3": [V,2,4,4]} (intrinsics K/4,
    K/2, K); depth_values [ndepths]; filename pattern ``<scan>/{}/<ref id>{}``.  Built from the pieces above: pair
    index, ``CamFile``, ``ResizePolicy``; every view of a sample ends up at the size of its reference view (or, with
    ``fix_res``, of the first view ever loaded)."""

    def __init__(self, datapath, listfile, mode, nviews, ndepths=192, interval_scale=1.06, inverse_depth=False, **kwargs):
        super().__init__()
        assert mode == "test"
        self.datapath, self.listfile, self.mode, self.nviews, self.ndepths = datapath, listfile, mode, nviews, ndepths
        self.policy = ResizePolicy(kwargs["max_h"], kwargs["max_w"], GLOBAL_BASE)
        self.inverse_depth = inverse_depth
        self.scene_size = None if not kwargs.get("fix_res", False) else "first"   # None | "first" | (h, w)
        self.interval_scale = {s: (interval_scale if isinstance(interval_scale, float) else interval_scale[s])
                               for s in listfile}
        self.metas = [(scan, ref, srcs) for scan in listfile
                      for ref, srcs in read_pairs(os.path.join(datapath, scan, "pair.txt"), nviews)]

    def __len__(self):
        return len(self.metas)

    def _view(self, scan, vid):
        img_filename = os.path.join(self.datapath, "{}/images_post/{:0>8}.jpg".format(scan, vid))
        if not os.path.exists(img_filename):
            img_filename = os.path.join(self.datapath, "{}/images/{:0>8}.jpg".format(scan, vid))
        from PIL import Image
        img = np.array(Image.open(img_filename), dtype=np.float32) / 255.0
        cam = CamFile.parse(os.path.join(self.datapath, "{}/cams/{:0>8}_cam.txt".format(scan, vid)))
        return img, cam.for_network(self.ndepths, self.interval_scale[scan])

    def __getitem__(self, idx):
        scan, ref_view, src_views = self.metas[idx]
        view_ids = [ref_view] + src_views[: self.nviews - 1]
        imgs, proj_matrices, depth_values, size = [], [], None, None
        for i, vid in enumerate(view_ids):
            img, (K, E, depth_min, depth_interval) = self._view(scan, vid)
            img, K = self.policy.apply(img, K)
            if self.scene_size == "first":
                self.scene_size = img.shape[:2]
            if i == 0:
                size = self.scene_size if isinstance(self.scene_size, tuple) else img.shape[:2]
                depth_values = depth_hypothesis_values(depth_min, depth_interval, self.ndepths, self.inverse_depth)
            if img.shape[:2] != tuple(size):
                img, K = self.policy.apply(img, K, size)
            imgs.append(img)
            proj_mat = np.zeros((2, 4, 4), dtype=np.float32)
            proj_mat[0], proj_mat[1, :3, :3] = E, K
            proj_matrices.append(proj_mat)
        proj_matrices = np.stack(proj_matrices)
        ms = {}
        for k, mul in (("stage1", 1), ("stage2", 2), ("stage3", 4)):
            ms[k] = proj_matrices.copy()
            ms[k][:, 1, :2, :] = proj_matrices[:, 1, :2, :] * mul
        return {"imgs": np.stack(imgs).transpose([0, 3, 1, 2]), "proj_matrices": ms, "depth_values": depth_values,
                "filename": scan + "/{}/" + "{:0>8}".format(view_ids[0]) + "{}"}


# ------------------------------------------------------------------------------------------ eval driver
@torch.no_grad()
def save_depth_maps(network, datapath: str, testlist: Sequence[str], outdir: str, num_view: int, max_h: int, max_w: int,
                    numdepth: int = 192, interval_scale: float = 1.06, inverse_depth: bool = False, device="cuda",
                    write_images: bool = True, fix_res: bool = False, scene_cfg: Optional[Dict[str, dict]] = None) -> List[str]:
    """Step 1 of Model.test (model.py:323-380): run ``network`` on every reference view of every scene and write
    ``<outdir>/<scan>/depth_est/%08d.pfm``, ``confidence/%08d.pfm``, ``cams/%08d_cam.txt`` (and ``images/%08d.jpg``).
    ``scene_cfg``: optional per-scene overrides ``{scene: {"max_h": .., "max_w": ..}}`` -- the reference's ``tank_cfg``
    table (model.py:325-328).  ``fix_res``: main.py's ``--fix_res``.  Returns the list of depth files written."""
    network.eval()
    num_stage = len(network.ndepths)
    written = []
    for scene in testlist:
        sc = (scene_cfg or {}).get(scene, {})
        ds = MVSDataset(datapath, [scene], "test", num_view, numdepth, interval_scale, inverse_depth=inverse_depth,
                        max_h=sc.get("max_h", max_h), max_w=sc.get("max_w", max_w), fix_res=fix_res)
        loader = torch.utils.data.DataLoader(ds, 1, shuffle=False, num_workers=0, drop_last=False)
        for sample in loader:
            imgs = sample["imgs"].to(device)
            proj = {k: v.to(device) for k, v in sample["proj_matrices"].items()}
            outputs = network(imgs, proj, sample["depth_values"].to(device))
            depth = outputs["depth"].cpu().numpy()
            conf = outputs["photometric_confidence"].cpu().numpy()
            cams = sample["proj_matrices"]["stage{}".format(num_stage)].numpy()
            for b, filename in enumerate(sample["filename"]):
                paths = {k: os.path.join(outdir, filename.format(k, ext)) for k, ext in
                         (("depth_est", ".pfm"), ("confidence", ".pfm"), ("cams", "_cam.txt"), ("images", ".jpg"))}
                for p in paths.values():
                    os.makedirs(os.path.dirname(p), exist_ok=True)
                save_pfm(paths["depth_est"], depth[b])
                save_pfm(paths["confidence"], conf[b])
                write_cam(paths["cams"], cams[b][0])
                if write_images:
                    from PIL import Image
                    img = np.clip(np.transpose(sample["imgs"][b, 0].numpy(), (1, 2, 0)) * 255, 0, 255).astype(np.uint8)
                    Image.fromarray(img).save(paths["images"])
                written.append(paths["depth_est"])
    return written


@torch.no_grad()
def run_test(network, datapath: str, testlist: Sequence[str], outdir: str, num_view: int, max_h: int, max_w: int,
             numdepth: int = 192, interval_scale: float = 1.06, inverse_depth: bool = False, conf=(0.1, 0.15, 0.7),
             thres_view: int = 5, filter_method: str = "pcd", device="cuda", fix_res: bool = False,
             dist_base: float = 1 / 4, rel_diff_base: float = 1 / 1300,
             scene_cfg: Optional[Dict[str, dict]] = None) -> Dict[str, Dict[str, float]]:
    """Both steps of ``Model.test`` (model.py:297-390): depth / confidence maps of every reference view, then the
    fusion filter per scene -- ``filter_method`` "pcd" (filter/pcd.py) or "dypcd" (the dynamic-threshold variant,
    filter/dypcd_tanks.py) -- into ``<outdir>/pcd/<name>.ply`` (``mvsnet%03d_l3.ply`` for DTU ``scanN`` names,
    pcd.py:365-370).  Defaults are main.py's: ``--conf 0.1 0.15 0.7``, ``--thres_view 5`` (main.py:60-61), ``--dist_base
    1/4``, ``--rel_diff_base 1/1300`` (main.py:63-64; the dynamic filter's ladder bases).  NOTE: step 1 writes only the
    final ``confidence.pfm`` (model.py:372-375), so -- exactly as in the reference -- all three thresholds are applied to
    that one map and the effective photometric gate is its largest value (0.7 by default).  ``scene_cfg``: per-scene
    overrides ``{scene: {"max_h", "max_w", "conf"}}`` (the reference's ``tank_cfg``: model.py:325-328, pcd.py:375-377).
    Returns the mask statistics of the last reference view per scene."""
    from . import fusion
    save_depth_maps(network, datapath, testlist, outdir, num_view, max_h, max_w, numdepth, interval_scale, inverse_depth,
                    device, fix_res=fix_res, scene_cfg=scene_cfg)
    os.makedirs(os.path.join(outdir, "pcd"), exist_ok=True)
    stats = {}
    for scan in testlist:
        name = "mvsnet{:0>3}_l3.ply".format(int(scan[4:])) if scan.startswith("scan") and scan[4:].isdigit() else "{}.ply".format(scan)
        pairs = fusion.read_pair_file(os.path.join(datapath, scan, "pair.txt"))
        sc = (scene_cfg or {}).get(scan, {})
        stats[scan] = fusion.fuse_scene(pairs, os.path.join(outdir, scan), os.path.join(outdir, "pcd", name),
                                        conf=sc.get("conf", conf), thres_view=thres_view, dynamic=filter_method == "dypcd",
                                        num_stage=len(network.ndepths), device=device, dist_base=dist_base,
                                        rel_diff_base=rel_diff_base)
    return stats
