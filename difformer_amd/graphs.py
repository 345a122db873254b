"""hipGraph replay of a whole DIFFormer forward.

The forward of a small graph (Cora, a Pokec mini-batch) is a dozen kernels of a few microseconds each:
launch-bound when issued one by one from Python.  Every entry point of the C ABI only enqueues on the
stream it is given and never allocates or synchronises, so the complete forward can be captured once
into a hipGraph and replayed with a single launch.

    fwd = difformer_amd.GraphedForward(model, x, edge_index)     # eval mode, fixed shapes
    out = fwd(x_new)                                             # copies x_new in, replays, returns the output

The graph (and so the CSR built for `edge_index`) is fixed at capture time; capture again for another
graph or another input shape.  Single-GPU only (collectives are not captured).
"""
from __future__ import annotations

import torch

from . import ops


class GraphedForward:
    def __init__(self, model, x, edge_index=None, edge_weight=None, warmup=2):
        if model.training:
            raise RuntimeError("GraphedForward captures the inference forward: call model.eval() first")
        if any(getattr(c, "row_shard", None) is not None and c.row_shard.world > 1 for c in model.convs):
            raise NotImplementedError("GraphedForward: row-sharded (multi-GPU) forwards are not captured")
        if not x.is_cuda:
            raise RuntimeError("difformer_amd: operands must live on the MI355X (no CPU fallback)")
        self.model = model
        self.x = x.detach().clone()
        self.edge_index, self.edge_weight = edge_index, edge_weight
        dev = x.device
        with torch.no_grad():
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):          # loads the library, builds and caches the CSR
                    model(self.x, edge_index, edge_weight)
            torch.cuda.current_stream(dev).wait_stream(side)
            # the captured kernels hold raw pointers into the CSR: keep it alive with this object
            self._csr = [v[2] for v in ops.csr_cache.entries.values()] if edge_index is not None else []
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = model(self.x, edge_index, edge_weight)

    def __call__(self, x=None):
        if x is not None:
            if x.shape != self.x.shape:
                raise ValueError(f"GraphedForward was captured for x of shape {tuple(self.x.shape)}, got {tuple(x.shape)}")
            self.x.copy_(x)
        self.graph.replay()
        return self.out
