"""dmvsnet_amd -- MI355X-native (gfx950) implementation of DMVSNet's cost-volume hot path.

Public surface mirrors the reference's network interface for this path
(/root/reference/networks/mvsnet.py): ``MVSNet(ndepths, depth_interval_ratio, ...)`` with
``forward(imgs, proj_matrices, depth_values) -> dict``.  The compute path is the hand-written HIP
library behind include/dmvs.h; there is no CPU or PyTorch fallback.
"""
from .mvsnet import CostAgg, CostRegNet, DepthNet, FeatureNet, MVSNet, shard_source_views  # noqa: F401

from . import eval_io  # noqa: F401  (PFM / cam I/O, eval dataset, Model.test step 1)
from . import fusion   # noqa: F401  (geometric-consistency fusion filter, PLY)

__all__ = ["MVSNet", "CostAgg", "CostRegNet", "DepthNet", "FeatureNet", "shard_source_views", "eval_io", "fusion"]
