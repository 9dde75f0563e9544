"""ctypes binding of libdmvs_hip.so (the C ABI declared in include/dmvs.h).

There is NO fallback: if the shared library is missing or a symbol is absent the import of the compute
path fails loudly.  ``import torch`` happens first so that the library's libamdhip64.so.7 dependency binds
to the HIP runtime PyTorch has already loaded (same SONAME) -- device pointers and streams are then shared.
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  (must be loaded before the HIP library, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# DMVS_LIB: development override (knock-out / experiment builds of the same ABI, scripts/ko_build.sh)
LIB_PATH = os.environ.get("DMVS_LIB") or os.path.join(_HERE, "csrc", "libdmvs_hip.so")

ABI_VERSION = 140   # include/dmvs.h DMVS_VERSION

_p = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float

# name -> (restype, argtypes); mirrors include/dmvs.h one to one.
SIGNATURES = {
    "dmvs_version": (_i, []),
    "dmvs_error_string": (ctypes.c_char_p, [_i]),
    "dmvs_tune": (_i, [ctypes.c_char_p, _i]),
    "dmvs_nchw_to_hwc": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "dmvs_planar_to_hwc": (_i, [_p, ctypes.c_long, _i, _i, _i, _i, _p, _p]),
    "dmvs_relative_proj": (_i, [_p, _i, _p, _p]),
    "dmvs_hypotheses_first": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "dmvs_hypotheses_next": (_i, [_p, _i, _i, _p, _i, _f, _i, _i, _p, _p, _p]),
    "dmvs_hypotheses_next_up": (_i, [_p, _i, _i, _i, _p, _i, _f, _i, _i, _i, _p, _p, _p]),
    "dmvs_hypothesis_base_first": (_i, [_p, _i, _i, _i, _i, _p, _p, _p]),
    "dmvs_hypothesis_base_next": (_i, [_p, _i, _i, _p, _i, _f, _i, _p, _p, _p]),
    "dmvs_warp_corr_affine": (_i, [_p, ctypes.POINTER(_p), _i, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "dmvs_depth_regress_affine": (_i, [_p, _p, _p, _f, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "dmvs_warp_corr": (_i, [_p, ctypes.POINTER(_p), _i, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "dmvs_warp_corr_q4": (_i, [_p, ctypes.POINTER(_p), _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "dmvs_warp_corr_q4_f16": (_i, [_p, ctypes.POINTER(_p), _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "dmvs_conv3d_direct": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dmvs_conv3d_mfma": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dmvs_conv3d_wino": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dmvs_conv3d_wino_plan": (_i, [_i, _i, _i, _i, _i, _i]),
    "dmvs_conv3d_wino_fpn2": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "dmvs_conv3d_wino_fpn_weight_floats": (ctypes.c_long, []),
    "dmvs_pack_conv_weights_wino_fpn": (_i, [_p, _p, _p, _p]),
    "dmvs_conv3d_wino_weight_floats": (ctypes.c_long, [_i, _i, _i]),
    "dmvs_conv3d_coarse": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dmvs_conv3d_coarse_weight_floats": (ctypes.c_long, [_i, _i, _i]),
    "dmvs_pack_conv_weights_coarse": (_i, [_p, _p, _i, _i, _i]),
    "dmvs_conv3d_zmarch": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dmvs_conv3d_zmarch_weight_floats": (ctypes.c_long, [_i, _i, _i]),
    "dmvs_pack_conv_weights_zmarch": (_i, [_p, _p, _i, _i, _i]),
    "dmvs_conv3d_split_probe": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "dmvs_conv3d_split_weight_floats": (ctypes.c_long, [_i, _i]),
    "dmvs_pack_conv_weights_split": (_i, [_p, _p, _i, _i]),
    "dmvs_conv2d_c8": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "dmvs_conv2d_c8_weight_floats": (ctypes.c_long, [_i]),
    "dmvs_featurenet_conv0": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "dmvs_pack_conv_weights_c8": (_i, [_p, _p, _i]),
    "dmvs_pack_conv_weights_wino": (_i, [_p, _p, _i, _i, _i]),
    "dmvs_conv3d_mfma_plan": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "dmvs_conv3d_mfma_weight_floats": (ctypes.c_long, [_i, _i, _i, _i]),
    "dmvs_pack_conv_weights_mfma": (_i, [_p, _p, _i, _i, _i, _i]),
    "dmvs_geo_consistency": (_i, [_p, _p, _p, _i, _i, _f, _f, _p, _p, _p, _p, _p]),
    "dmvs_geo_consistency_ladder": (_i, [_p, _p, _p, _i, _i, _f, _f, _p, _p, _p, _p, _p, _p]),
    "dmvs_prob_regress": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _f, _p, _p]),
    "dmvs_depth_select": (_i, [_p, _p, _i, _i, _i, _p, _p, _p]),
    "dmvs_depth_regress": (_i, [_p, _p, _p, _f, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
}

EINVAL, EUNSUPPORTED = -1, -2   # include/dmvs.h

_lib = None


class DmvsError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load (once) and type the library.  Raises if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DmvsError(
            f"{LIB_PATH} not found: the HIP kernels are the only compute path (no CPU / PyTorch fallback). "
            "Build them with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C dmvsnet_amd/csrc`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud
        fn.restype = res
        fn.argtypes = args
    if hasattr(lib, "dmvs_dev_build") and os.environ.get("DMVS_ALLOW_DEV_BUILD") != "1":
        raise DmvsError(f"{LIB_PATH} is a DEVELOPMENT build (knock-out / trace / experiment switches compiled in, csrc/dev_guard.h): "
                        "its results may be wrong by design.  Set DMVS_ALLOW_DEV_BUILD=1 to load it anyway")
    if lib.dmvs_version() != ABI_VERSION:
        raise DmvsError(f"libdmvs_hip.so version {lib.dmvs_version()} does not match the Python host ({ABI_VERSION})")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().dmvs_error_string(code)
        raise DmvsError(f"{what} failed: {msg.decode() if msg else code} (code {code})")
