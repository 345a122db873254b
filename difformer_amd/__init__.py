"""difformer_amd -- MI355X-native DIFFormer propagation layer (drop-in for the reference `difformer` module).

    from difformer_amd import DIFFormer
    model = DIFFormer(in_channels, hidden_channels, out_channels, use_graph=True, kernel='simple').cuda()
    logits = model(x, edge_index)

The arithmetic lives in lib/libdifformer_hip.so (hand-written gfx950 kernels, C ABI in
include/difformer_hip.h).  Importing this package does not load the library; the first operator
call does, and fails loudly if it has not been built.
"""
from .difformer import DIFFormer, DIFFormerConv, full_attention_conv, gcn_conv  # noqa: F401
from .difformer_v2 import DIFFormer_v2, TransConv  # noqa: F401
from .dist import RowShard  # noqa: F401
from .graphs import GraphedForward, GraphedTrainStep, graphed_training  # noqa: F401

__all__ = ["DIFFormer", "DIFFormerConv", "full_attention_conv", "gcn_conv", "DIFFormer_v2", "TransConv", "RowShard",
           "GraphedForward", "GraphedTrainStep", "graphed_training"]
__version__ = "0.1.0"
