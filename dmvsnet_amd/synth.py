"""Deterministic synthetic inputs and weights for the DMVSNet hot path.

No dataset or checkpoint is reachable from the build or GPU boxes, so every
test / bench / golden vector uses the generators below.  Everything is driven
by NumPy ``PCG64`` (version-stable) and plain float32 NumPy arithmetic so the
same seed gives bit-identical tensors in the build container (where the golden
vectors are produced from the reference) and on the GPU box.

Shapes follow the reference's eval loader contract
(/root/reference/datasets/general_eval.py:69,172-198): ``imgs [B,V,3,H,W]`` in
[0,1], ``proj_matrices["stageK"] [B,V,2,4,4]`` with ``[...,0,:,:]`` the
world->camera extrinsic and ``[...,1,:3,:3]`` the intrinsics at that stage's
scale, ``depth_values [B,192]`` ascending.
"""
from __future__ import annotations

import zlib
from typing import Dict, Sequence

import numpy as np
import torch

__all__ = [
    "synth_fusion_scene",
    "synth_images",
    "synth_cameras",
    "synth_depth_values",
    "synth_inputs",
    "synth_state_dict",
    "CONFIGS",
]

# BASELINE.json configs, restated as numbers.
CONFIGS = {
    # plumbing config: image 128x160 -> 32x40 volume, 3 views, single stage D=8
    "c1": dict(H=128, W=160, V=3, ndepths=[8], ratios=[4]),
    # 3-stage variant of c1 used for end-to-end parity fixtures
    "c1s3": dict(H=128, W=160, V=3, ndepths=[16, 8, 8], ratios=[3, 2, 1]),
    # headline: DTU eval 1600x1184, 5 views, 64/32/8
    "c2": dict(H=1184, W=1600, V=5, ndepths=[64, 32, 8], ratios=[3, 2, 1]),
    "c3": dict(H=1184, W=1600, V=11, ndepths=[64, 32, 8], ratios=[3, 2, 1]),
    "c4": dict(H=1024, W=1920, V=11, ndepths=[64, 32, 8], ratios=[3, 2, 1]),
    # BASELINE configs[4] (BlendedMVS 2048x1536, 7 views, 4 stages 96/64/32/8): an EXTENSION -- the reference cannot
    # express a 4th stage; stages 1-2 run at the coarsest FPN level (MVSNet.stage_level)
    "c5": dict(H=1536, W=2048, V=7, ndepths=[96, 64, 32, 8], ratios=[4, 3, 2, 1]),
    # the reference's own eval recipes (not BASELINE lines).  DTU: scripts/dtu_test.sh:10-29 --
    # 48/32/8 planes, ratios 4/2/1, 5 views, --max_h 864 --max_w 1152 (the 1200x1600 frames shrink to 864x1152,
    # general_eval.py:97-110), --inverse_depth.  Tanks&Temples: scripts/tank_test.sh:10-23 -- 64/32/8, 3/2/1, 11 views,
    # filter/tank_test_config.py:10-11 max_h 1080 / max_w 2048: a 1080x2048 frame is only rounded down to base 32
    # (eval_io.ResizePolicy(1080, 2048).target(1080, 2048) == (1056, 2048)), linear depth sampling.
    "dtu": dict(H=864, W=1152, V=5, ndepths=[48, 32, 8], ratios=[4, 2, 1], inverse=True),
    "tnt": dict(H=1056, W=2048, V=11, ndepths=[64, 32, 8], ratios=[3, 2, 1]),
    # c3 at about a quarter of the linear size: what the multi-rank tests (several ranks sharing one GPU over gloo) run
    "c3_small": dict(H=288, W=416, V=11, ndepths=[16, 8, 8], ratios=[3, 2, 1]),
    # ... and config 2's view count at that size: 5 views on 8 ranks = 2 view groups x 4 (the default multi-GPU bench line)
    "c2_small": dict(H=288, W=416, V=5, ndepths=[16, 8, 8], ratios=[3, 2, 1]),
}


def _rng(seed: int, tag: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(tag.encode())]))


def _upsample_bilinear_np(a: np.ndarray, H: int, W: int) -> np.ndarray:
    """Separable bilinear upsample (half-pixel centres, edge clamp), float32."""
    h, w = a.shape[-2:]

    def axis_weights(n_out, n_in):
        src = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        src = np.clip(src, 0.0, n_in - 1)
        i0 = np.floor(src).astype(np.int64)
        i1 = np.minimum(i0 + 1, n_in - 1)
        f = (src - i0).astype(np.float32)
        return i0, i1, f

    y0, y1, fy = axis_weights(H, h)
    x0, x1, fx = axis_weights(W, w)
    rows = a[..., y0, :] * (1 - fy)[:, None] + a[..., y1, :] * fy[:, None]
    out = rows[..., :, x0] * (1 - fx) + rows[..., :, x1] * fx
    return out.astype(np.float32)


def synth_images(H: int, W: int, V: int, seed: int = 0) -> torch.Tensor:
    """[1,V,3,H,W] float32 in [0,1]: shared low-frequency texture + per-view noise."""
    g = _rng(seed, "imgs.base")
    base = g.random((3, max(H // 8, 2), max(W // 8, 2)), dtype=np.float32)
    base = _upsample_bilinear_np(base, H, W)
    imgs = np.empty((1, V, 3, H, W), dtype=np.float32)
    for v in range(V):
        noise = _rng(seed, f"imgs.noise.{v}").random((3, H, W), dtype=np.float32)
        imgs[0, v] = np.clip(base + np.float32(0.1) * noise, 0.0, 1.0)
    return torch.from_numpy(imgs)


def synth_cameras(H: int, W: int, V: int, num_stage: int = 3) -> Dict[str, torch.Tensor]:
    """DTU-like cameras.  Stage k (1-based) intrinsics = K_full / 2**(3-k) on rows 0,1
    (mirrors general_eval.py:189-198, where stage1 = K/4 ... stage3 = K)."""
    K = np.array(
        [[2892.33 * W / 1600.0, 0.0, W / 2.0], [0.0, 2883.18 * H / 1200.0, H / 2.0], [0.0, 0.0, 1.0]],
        dtype=np.float64,
    )
    out = {}
    for s in range(1, 4):
        scale = 2.0 ** (3 - s)
        P = np.zeros((1, V, 2, 4, 4), dtype=np.float32)
        Ks = K.copy()
        Ks[:2, :] /= scale
        for v in range(V):
            a = 0.03 * v
            R = np.array([[np.cos(a), 0.0, np.sin(a)], [0.0, 1.0, 0.0], [-np.sin(a), 0.0, np.cos(a)]])
            E = np.eye(4)
            E[:3, :3] = R
            E[:3, 3] = [-30.0 * v, 7.0 * v, 5.0 * v]
            P[0, v, 0] = E.astype(np.float32)
            P[0, v, 1, :3, :3] = Ks.astype(np.float32)
        out[f"stage{s}"] = torch.from_numpy(P)
    return out


def synth_depth_values(n: int = 192) -> torch.Tensor:
    """[1,n]: 425 + 2.5*1.06*i  (DTU depth_min, interval * --interval_scale)."""
    d = np.float32(425.0) + np.float32(2.5 * 1.06) * np.arange(n, dtype=np.float32)
    return torch.from_numpy(d[None].astype(np.float32))


def synth_inputs(H: int, W: int, V: int, seed: int = 0):
    return synth_images(H, W, V, seed), synth_cameras(H, W, V), synth_depth_values()


def synth_state_dict(template: Dict[str, torch.Tensor], seed: int = 0) -> Dict[str, torch.Tensor]:
    """Fill a state_dict *template* (keys + shapes) with a non-degenerate recipe.

    PyTorch default init + fresh BN statistics gives uniform softmax / confidence==1
    (SURVEY.md section 8c).  Recipe: conv weights ~ U(-b,b), b = 1/sqrt(fan_in);
    BN weight ~ U(.5,1.5), bias ~ N(0,.1), running_mean ~ N(0,.1), running_var ~ U(.5,1.5);
    every ``*.prob.weight`` multiplied by 20.  Each tensor is drawn from its own
    PCG64 stream keyed by (seed, crc32(key)) so the result is independent of key order.
    """
    out = {}
    for key, ref in template.items():
        shape = tuple(ref.shape)
        g = _rng(seed, key)
        if key.endswith("num_batches_tracked"):
            val = np.zeros(shape, dtype=np.int64)
        elif key.endswith("bn.weight") or key.endswith("running_var"):
            val = (0.5 + g.random(shape, dtype=np.float32)).astype(np.float32)
        elif key.endswith("bn.bias") or key.endswith("running_mean"):
            val = (np.float32(0.1) * g.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        elif key.endswith("weight"):
            fan_in = int(np.prod(shape[1:]))
            b = np.float32(1.0 / np.sqrt(fan_in))
            val = ((g.random(shape, dtype=np.float32) * 2 - 1) * b).astype(np.float32)
            if key.endswith("prob.weight"):
                val = val * np.float32(20.0)
        elif key.endswith("bias"):
            val = ((g.random(shape, dtype=np.float32) * 2 - 1) * np.float32(0.1)).astype(np.float32)
        else:
            raise KeyError(f"synth_state_dict: no recipe for {key}")
        out[key] = torch.from_numpy(np.ascontiguousarray(val))
    return out


def synth_fusion_scene(H: int, W: int, V: int, seed: int = 0, z0: float = 620.0):
    """A scene for the depth-map fusion filter (row N4): the world plane ``0.05 x + 0.03 y + z = z0`` seen by the
    synthetic cameras.  Returns ``(cams [V,2,4,4], depths, confs, imgs)``: per view an exact plane depth map perturbed
    by 0.3 % multiplicative noise (so that part of the pixels sit near the 1 % consistency gate), a block of zero
    depths, a block of gross outliers, three confidence maps (stage 1 / 2 / 3) and an RGB image in [0,1]."""
    cams = synth_cameras(H, W, V)["stage3"][0].numpy()
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    n, depths, confs, imgs = np.array([0.05, 0.03, 1.0]), [], [], []
    for v in range(V):
        K, E = cams[v, 1, :3, :3].astype(np.float64), cams[v, 0].astype(np.float64)
        R, t = E[:3, :3], E[:3, 3]
        rays = np.linalg.inv(K) @ np.stack((xs.ravel(), ys.ravel(), np.ones(H * W)))
        d = (z0 + n @ (R.T @ t)) / (n @ (R.T @ rays))                 # n . R^T (d ray - t) = z0
        g = _rng(seed, f"fusion.{v}")
        d = d.reshape(H, W) * (1.0 + 0.003 * g.standard_normal((H, W)))
        d[H // 3:H // 3 + 6, W // 4:W // 4 + 9] = 0.0                 # holes
        d[H // 2:H // 2 + 5, W // 2:W // 2 + 7] *= 1.2                # gross outliers
        depths.append(d.astype(np.float32))
        confs.append(g.random((3, H, W), dtype=np.float32))
        imgs.append(g.random((H, W, 3), dtype=np.float32))
    return cams, depths, confs, imgs
