"""Eval-side I/O either side of the network boundary.

Mirrors, with the same names and argument meaning, what the reference's eval path uses:
  read_pfm / save_pfm          /root/reference/datasets/data_io.py:6-71
  write_cam                    /root/reference/tools.py:40-57
  MVSDataset (mode "test")     /root/reference/datasets/general_eval.py:9-203
  save_depth_maps              step 1 of Model.test, /root/reference/model.py:323-380

The reference resizes with cv2 (absent from this image); here bilinear resizing is torch's
``F.interpolate(mode="bilinear", align_corners=False)`` which has the same half-pixel-centre, non-antialiased
definition as ``cv2.resize(..., INTER_LINEAR)``.  cv2 not being importable, the loader's resize branch is NOT
pinned against the reference (parity unpinned for that branch); inputs whose size already is a multiple of 32
and within (max_h, max_w) take the identity branch, which is exact.
"""
from __future__ import annotations

import os
import re
import sys
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

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
def _resize_bilinear(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    if img.shape[0] == new_h and img.shape[1] == new_w:
        return img
    t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1)[None]
    t = F.interpolate(t, (new_h, new_w), mode="bilinear", align_corners=False)
    return t[0].permute(1, 2, 0).contiguous().numpy()


class MVSDataset(torch.utils.data.Dataset):
    """Eval loader for DTU / Tanks&Temples style scenes: ``<scan>/pair.txt``, ``<scan>/cams/%08d_cam.txt``,
    ``<scan>/images(_post)/%08d.jpg``.  Sample dict identical to general_eval.py:200-203:
    imgs [V,3,H,W] in [0,1]; proj_matrices {"stage1|2|3": [V,2,4,4]} (stage1 intrinsics = K/4, x2, x4);
    depth_values [ndepths]; filename pattern ``<scan>/{}/<ref id>{}``."""

    def __init__(self, datapath, listfile, mode, nviews, ndepths=192, interval_scale=1.06, inverse_depth=False, **kwargs):
        super().__init__()
        assert mode == "test"
        self.datapath, self.listfile, self.mode, self.nviews, self.ndepths = datapath, listfile, mode, nviews, ndepths
        self.max_h, self.max_w = kwargs["max_h"], kwargs["max_w"]
        self.fix_res = kwargs.get("fix_res", False)
        self.fix_wh = False
        self.inverse_depth = inverse_depth
        self._std_hw = (0, 0)
        self.interval_scale = {s: (interval_scale if isinstance(interval_scale, float) else interval_scale[s])
                               for s in listfile}
        self.metas = self.build_list()

    def build_list(self):
        metas = []
        for scan in self.listfile:
            with open(os.path.join(self.datapath, scan, "pair.txt")) as f:
                for _ in range(int(f.readline())):
                    ref_view = int(f.readline().rstrip())
                    src_views = [int(x) for x in f.readline().rstrip().split()[1::2]]
                    if len(src_views) > 0:
                        if len(src_views) < self.nviews - 1:  # fill to nviews with the best source view
                            src_views += [src_views[0]] * (self.nviews - len(src_views))
                        metas.append((scan, ref_view, src_views, scan))
        return metas

    def __len__(self):
        return len(self.metas)

    def read_cam_file(self, filename, interval_scale):
        with open(filename) as f:
            lines = [line.rstrip() for line in f.readlines()]
        extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
        intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
        intrinsics[:2, :] /= 4.0
        parts = lines[11].split()
        depth_min, depth_interval = float(parts[0]), float(parts[1])
        if len(parts) >= 3:
            depth_max = depth_min + int(float(parts[2])) * depth_interval
            depth_interval = (depth_max - depth_min) / self.ndepths
        return intrinsics, extrinsics, depth_min, depth_interval * interval_scale

    @staticmethod
    def read_img(filename):
        from PIL import Image
        return np.array(Image.open(filename), dtype=np.float32) / 255.0

    def scale_mvs_input(self, img, intrinsics, max_w, max_h, base=GLOBAL_BASE):
        h, w = img.shape[:2]
        if h > max_h or w > max_w:
            scale = 1.0 * max_h / h
            if scale * w > max_w:
                scale = 1.0 * max_w / w
            new_w, new_h = scale * w // base * base, scale * h // base * base
        else:
            new_w, new_h = 1.0 * w // base * base, 1.0 * h // base * base
        intrinsics[0, :] *= 1.0 * new_w / w
        intrinsics[1, :] *= 1.0 * new_h / h
        return _resize_bilinear(img, int(new_h), int(new_w)), intrinsics

    def __getitem__(self, idx):
        scan, ref_view, src_views, scene_name = self.metas[idx]
        view_ids = [ref_view] + src_views[: self.nviews - 1]
        imgs, proj_matrices, depth_values = [], [], None
        for i, vid in enumerate(view_ids):
            img_filename = os.path.join(self.datapath, "{}/images_post/{:0>8}.jpg".format(scan, vid))
            if not os.path.exists(img_filename):
                img_filename = os.path.join(self.datapath, "{}/images/{:0>8}.jpg".format(scan, vid))
            cam_filename = os.path.join(self.datapath, "{}/cams/{:0>8}_cam.txt".format(scan, vid))
            img = self.read_img(img_filename)
            intrinsics, extrinsics, depth_min, depth_interval = self.read_cam_file(cam_filename, self.interval_scale[scene_name])
            img, intrinsics = self.scale_mvs_input(img, intrinsics, self.max_w, self.max_h)
            if self.fix_res:  # one standard size for the whole scene
                self._std_hw = img.shape[:2]
                self.fix_res, self.fix_wh = False, True
            if i == 0 and not self.fix_wh:
                self._std_hw = img.shape[:2]
            s_h, s_w = self._std_hw
            c_h, c_w = img.shape[:2]
            if (c_h != s_h) or (c_w != s_w):
                img = _resize_bilinear(img, s_h, s_w)
                intrinsics[0, :] *= 1.0 * s_w / c_w
                intrinsics[1, :] *= 1.0 * s_h / c_h
            imgs.append(img)
            proj_mat = np.zeros((2, 4, 4), dtype=np.float32)
            proj_mat[0, :4, :4] = extrinsics
            proj_mat[1, :3, :3] = intrinsics
            proj_matrices.append(proj_mat)
            if i == 0:
                if self.inverse_depth:
                    depth_end = depth_interval * self.ndepths + depth_min
                    dv = np.linspace(1.0 / depth_min, 1.0 / depth_end, self.ndepths, endpoint=False)
                    depth_values = (1.0 / dv).astype(np.float32)
                else:
                    depth_values = np.arange(depth_min, depth_interval * (self.ndepths - 0.5) + depth_min, depth_interval,
                                             dtype=np.float32)
        imgs = np.stack(imgs).transpose([0, 3, 1, 2])
        proj_matrices = np.stack(proj_matrices)
        ms = {"stage1": proj_matrices}
        for k, mul in (("stage2", 2), ("stage3", 4)):
            p = proj_matrices.copy()
            p[:, 1, :2, :] = proj_matrices[:, 1, :2, :] * mul
            ms[k] = p
        return {"imgs": imgs, "proj_matrices": ms, "depth_values": depth_values,
                "filename": scan + "/{}/" + "{:0>8}".format(view_ids[0]) + "{}"}


# ------------------------------------------------------------------------------------------ eval driver
@torch.no_grad()
def save_depth_maps(network, datapath: str, testlist: Sequence[str], outdir: str, num_view: int, max_h: int, max_w: int,
                    numdepth: int = 192, interval_scale: float = 1.06, inverse_depth: bool = False, device="cuda",
                    write_images: bool = True) -> List[str]:
    """Step 1 of Model.test (model.py:323-380): run ``network`` on every reference view of every scene and write
    ``<outdir>/<scan>/depth_est/%08d.pfm``, ``confidence/%08d.pfm``, ``cams/%08d_cam.txt`` (and ``images/%08d.jpg``).
    Returns the list of depth files written."""
    network.eval()
    num_stage = len(network.ndepths)
    written = []
    for scene in testlist:
        ds = MVSDataset(datapath, [scene], "test", num_view, numdepth, interval_scale, inverse_depth=inverse_depth,
                        max_h=max_h, max_w=max_w, fix_res=False)
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
