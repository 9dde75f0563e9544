#!/usr/bin/env python3
"""Golden vectors for the eval input pipeline (row N3) by RUNNING THE REFERENCE's loader, datasets/general_eval.py
(build container only; /root/reference is never copied).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eval.py

The reference module imports cv2, which this image lacks.  It needs it for exactly two things: the ``INTER_NEAREST``
default argument of ``scale_depth_input`` (general_eval.py:115) and ``cv2.resize`` (:112, :124, :166).  The generator
registers a placeholder ``cv2`` with the ``INTER_*`` constants and a ``resize`` that ASSERTS the target size equals the
source size and returns its input: on base-32 images below max_h / max_w every resize of the loader is a same-size
call, so the placeholder is the identity cv2.resize is there too, and anything else raises.  What stays unpinned is
cv2.resize's interpolation for images that DO get resized (stated in DESIGN.md / INTEGRATION.md).

Writes tests/golden/eval_scene/<scan>/{images/*.jpg, cams/*_cam.txt, pair.txt} (the on-disk inputs: data) and
tests/golden/eval_dataset.npz: for every sample the reference's ``MVSDataset.__getitem__`` returns -- imgs, the three
proj_matrices scales, depth_values, filename -- for
  scanA  64 x 96, 4 views, cam files with "depth_min depth_interval"                 nviews 3, linear + inverse depth
  scanB  96 x 64, 3 views, cam files with "depth_min depth_interval num_depth ..."   nviews 5 (> views: source list padded)
"""
import contextlib
import io
import os
import shutil
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from dmvsnet_amd import synth  # noqa: E402


def write_scene(root, scan, H, W, V, depth_line):
    """A synthetic MVSNet-format scene (same writer as tests/test_eval_io.py::_write_scene)."""
    from PIL import Image
    os.makedirs(os.path.join(root, scan, "cams"))
    os.makedirs(os.path.join(root, scan, "images"))
    imgs = synth.synth_images(H, W, V, seed=3)[0]
    cams = synth.synth_cameras(H, W, V)["stage3"][0].numpy()
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


def main():
    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST, cv2.INTER_LINEAR, cv2.INTER_CUBIC, cv2.INTER_AREA = 0, 1, 2, 3

    def resize(img, size, interpolation=None):
        assert (int(size[0]), int(size[1])) == (img.shape[1], img.shape[0]), \
            f"placeholder cv2.resize: only same-size calls are pinned ({img.shape[:2]} -> {size})"
        return img
    cv2.resize = resize
    sys.modules["cv2"] = cv2
    # the package's __init__ imports the TRAINING loaders (torchvision, absent here): register the package by path only,
    # so that `datasets.general_eval` and its `from datasets.data_io import *` load from the reference tree
    pkg = types.ModuleType("datasets")
    pkg.__path__ = ["/root/reference/datasets"]
    sys.modules["datasets"] = pkg
    from datasets.general_eval import MVSDataset   # the reference

    scene_root = os.path.join(HERE, "eval_scene")
    shutil.rmtree(scene_root, ignore_errors=True)
    write_scene(scene_root, "scanA", 64, 96, 4, "425.0 2.5")
    write_scene(scene_root, "scanB", 96, 64, 3, "425.0 2.5 128 935.0")
    out = {}
    cases = (("A_lin", "scanA", 3, False), ("A_inv", "scanA", 3, True), ("B_pad", "scanB", 5, False))
    for tag, scan, nviews, inverse in cases:
        with contextlib.redirect_stdout(io.StringIO()):
            ds = MVSDataset(scene_root, [scan], "test", nviews, 192, 1.06, inverse_depth=inverse, max_h=1200, max_w=1600)
        out[f"{tag}.n"] = np.int64(len(ds))
        for i in range(len(ds)):
            s = ds[i]
            out[f"{tag}.{i}.imgs"] = s["imgs"]
            for k in ("stage1", "stage2", "stage3"):
                out[f"{tag}.{i}.{k}"] = s["proj_matrices"][k]
            out[f"{tag}.{i}.depth_values"] = s["depth_values"]
            out[f"{tag}.{i}.filename"] = np.array(s["filename"])
    np.savez_compressed(os.path.join(HERE, "eval_dataset.npz"), **out)
    print("wrote", os.path.join(HERE, "eval_dataset.npz"), len(out), "arrays;", scene_root)


if __name__ == "__main__":
    main()
