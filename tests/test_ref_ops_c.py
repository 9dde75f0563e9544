"""CPU: the plain-C restatement of the ATen op semantics (oracle/ref_ops.c) against the torch oracle and the
reference's golden vectors.  Tiny shapes only (scalar loops)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import dmvs_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32P = ctypes.POINTER(ctypes.c_float)


@pytest.fixture(scope="module")
def clib():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    return ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle_ref.so"))


def fp(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(F32P)


def test_warp_corr_c(clib, golden):
    g = golden("op_costagg.npz")
    feats = [np.ascontiguousarray(g[f"feat{v}"][0]) for v in range(3)]
    proj = torch.from_numpy(g["proj"])
    p12 = []
    for v in (1, 2):
        rot, tr = O.relative_projection(proj[:, v], proj[:, 0])
        p12.append(np.concatenate([rot[0].numpy().ravel(), tr[0].numpy().ravel()]))
    p12 = np.ascontiguousarray(np.stack(p12).astype(np.float32))
    C, H, W = feats[0].shape
    depth = np.ascontiguousarray(g["depth"][0])
    D = depth.shape[0]
    sim = np.empty((2, D, H, W), np.float32)
    srcs = (F32P * 2)(fp(feats[1]), fp(feats[2]))
    clib.ref_warp_corr(fp(feats[0]), srcs, 2, fp(p12), fp(depth), fp(sim), C, D, H, W)
    np.testing.assert_allclose(sim, g["sim"][0], atol=2e-6)


@pytest.mark.parametrize("stride,kd", [(1, 3), (2, 3), (1, 1), (2, 1)])
def test_conv3d_c(clib, stride, kd):
    rng = np.random.default_rng(3)
    Cin, Cout, D, H, W = 3, 4, (5 if kd == 3 else 1), 6, 7
    x = rng.standard_normal((Cin, D, H, W)).astype(np.float32)
    w = rng.standard_normal((Cout, Cin, kd, 3, 3)).astype(np.float32)
    sd = stride if kd == 3 else 1
    ref = torch.nn.functional.conv3d(torch.from_numpy(x)[None], torch.from_numpy(w), None, (sd, stride, stride),
                                     (kd // 2, 1, 1))[0].numpy()
    out = np.empty_like(ref)
    clib.ref_conv3d(fp(x), fp(w), fp(out), Cin, Cout, D, H, W, kd, sd, stride)
    np.testing.assert_allclose(out, ref, atol=1e-5)


@pytest.mark.parametrize("kd", [3, 1])
def test_deconv3d_c(clib, kd):
    rng = np.random.default_rng(4)
    Cin, Cout, D, H, W = 3, 2, (3 if kd == 3 else 1), 4, 5
    x = rng.standard_normal((Cin, D, H, W)).astype(np.float32)
    w = rng.standard_normal((Cin, Cout, kd, 3, 3)).astype(np.float32)
    if kd == 3:
        ref = torch.nn.functional.conv_transpose3d(torch.from_numpy(x)[None], torch.from_numpy(w), None, 2, 1, 1)[0]
    else:
        ref = torch.nn.functional.conv_transpose2d(torch.from_numpy(x[:, 0])[None], torch.from_numpy(w[:, :, 0]), None,
                                                   2, 1, 1)[0].unsqueeze(1)
    ref = ref.numpy()
    out = np.empty_like(ref)
    clib.ref_deconv3d(fp(x), fp(w), fp(out), Cin, Cout, D, H, W, kd)
    np.testing.assert_allclose(out, ref, atol=1e-5)


def test_bn_relu_add_c(clib):
    rng = np.random.default_rng(5)
    C, n = 4, 30
    x = rng.standard_normal((C, n)).astype(np.float32)
    gamma, beta, mean = (rng.standard_normal(C).astype(np.float32) for _ in range(3))
    var = (rng.random(C) + 0.5).astype(np.float32)
    skip = rng.standard_normal((C, n)).astype(np.float32)
    ref = torch.relu(torch.nn.functional.batch_norm(torch.from_numpy(x)[None], torch.from_numpy(mean),
                                                    torch.from_numpy(var), torch.from_numpy(gamma),
                                                    torch.from_numpy(beta), False, 0.0, 1e-5))[0].numpy() + skip
    y = x.copy()
    clib.ref_bn_relu_add(fp(y), fp(gamma), fp(beta), fp(mean), fp(var), fp(skip), C, ctypes.c_size_t(n), 1)
    np.testing.assert_allclose(y, ref, atol=1e-6)


def test_depth_regress_c(clib, golden):
    g = golden("op_depthnet.npz")
    logits = np.ascontiguousarray(g["logits"][0])
    dv = np.ascontiguousarray(g["depth_values"][0])
    _, D, H, W = logits.shape
    dsp = np.empty((4, H, W), np.float32); sel = np.empty((4, H, W), np.float32); conf = np.empty((H, W), np.float32)
    clib.ref_depth_regress(fp(logits), fp(dv), ctypes.c_float(float(g["interval"])), ctypes.c_float(1.0), 0, D, H, W,
                           fp(dsp), fp(sel), fp(conf))
    np.testing.assert_allclose(dsp, g["dsp"][0], atol=2e-4)
    np.testing.assert_allclose(sel, g["hyps"][0], atol=2e-3)
    np.testing.assert_allclose(conf, g["conf"][0], atol=1e-5)
    lc = np.ascontiguousarray(g["logits_c"][0]); hy = np.ascontiguousarray(g["hyps"][0])
    dep = np.empty((H, W), np.float32)
    clib.ref_depth_regress(fp(lc), fp(hy), ctypes.c_float(float(g["interval"])), ctypes.c_float(5.0), 1, 4, H, W,
                           fp(dsp), fp(dep), fp(conf))
    np.testing.assert_allclose(dep, g["depth"][0], atol=2e-4)
    np.testing.assert_allclose(conf, g["conf_refine"][0], atol=1e-5)
