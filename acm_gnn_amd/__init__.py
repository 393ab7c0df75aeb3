"""acm_gnn_amd -- MI355X-native ACM graph-convolution layer.

The hot path of SitaoLuan/ACM-GNN (``GraphConvolution``: three/four-channel
filterbank + adaptive channel mixing) as hand-written gfx950 HIP kernels behind
a C ABI (``include/acm_hip.h``, ``libacm_hip.so``), exposed through a
``torch.autograd.Function`` and a module with the reference's interface.
"""
from .layers import GraphConvolution, MLP  # noqa: F401
from .models import GCN  # noqa: F401
from .graph import CsrGraph, FilterOperators, SparseFeatures, operators_for  # noqa: F401
from .optim import FusedAdam, FusedAdamW  # noqa: F401
from . import tuning  # noqa: F401

__all__ = ["GraphConvolution", "MLP", "GCN", "CsrGraph", "FilterOperators", "SparseFeatures", "operators_for", "FusedAdam", "FusedAdamW"]
