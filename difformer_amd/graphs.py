"""hipGraph replay of a whole DIFFormer forward.

The forward of a small graph (Cora, a Pokec mini-batch) is a dozen kernels of a few microseconds each:
launch-bound when issued one by one from Python.  Every entry point of the C ABI only enqueues on the
stream it is given and never allocates or synchronises, so the complete forward can be captured once
into a hipGraph and replayed with a single launch.

    fwd = difformer_amd.GraphedForward(model, x, edge_index)     # eval mode, fixed shapes
    out = fwd(x_new)                                             # copies x_new in, replays, returns the output

The graph (and so the CSR built for `edge_index`) is fixed at capture time; capture again for another
graph or another input shape.  Single-GPU only (collectives are not captured).

Training on a fixed graph (node classification/main.py:117-131: the same `model(x, edge_index)` every epoch) is launch-bound
the same way, forward AND backward (a Cora-sized step: ~150 launches, 2.0 ms, 0.6 ms of it kernels):

    model = difformer_amd.graphed_training(model, x, edge_index)  # one line after the model is built
    out = model(x, edge_index); loss = ...; loss.backward(); optimizer.step()    # unchanged loop

replays the training forward and its backward as two hipGraphs (torch.cuda.make_graphed_callables over the module: the
autograd functions of this package only enqueue on the current stream); evaluation calls (`model.eval()`) keep the
ordinary forward.
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
            be = ops.get_backend()
            self._pins = []                  # ... and into packed weight buffers (backend `capture_pins`)
            be.capture_pins = self._pins
            try:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self.out = model(self.x, edge_index, edge_weight)
            finally:
                be.capture_pins = None
            # the other derived tensors the captured kernels read: concatenated projections, weight-only factors, float32
            # copies of bfloat16 parameters -- a later eager call with changed parameters replaces them in their caches
            self._pins.append([(c._fused_wb, c._wide, c._narrow) for c in getattr(model, "convs", [])])
            self._pins.append([v[1] for v in ops._F32_PARAMS.values()])

    def __call__(self, x=None):
        if x is not None:
            if x.shape != self.x.shape:
                raise ValueError(f"GraphedForward was captured for x of shape {tuple(self.x.shape)}, got {tuple(x.shape)}")
            self.x.copy_(x)
        self.graph.replay()
        return self.out


def graphed_training(model, x, edge_index=None, edge_weight=None, warmup=3):
    """-> the same module, with its TRAINING forward and backward captured as hipGraphs for these operands (fixed shapes,
    the same graph every step: full-graph training as in main.py:117-131).  Calls in eval mode run the module's own
    forward.  The operands passed later must have the shapes (and, for `edge_index`, the content) used here; new values
    of x are copied into the captured input.  Parameters are updated in place by the optimiser as usual."""
    if not x.is_cuda:
        raise RuntimeError("difformer_amd: operands must live on the MI355X (no CPU fallback)")
    if any(getattr(c, "row_shard", None) is not None and c.row_shard.world > 1 for c in getattr(model, "convs", [])):
        raise NotImplementedError("graphed_training: row-sharded (multi-GPU) steps are not captured")
    was_training = model.training
    model.train()
    args = tuple(t for t in (x, edge_index, edge_weight) if t is not None)
    if edge_index is None and edge_weight is not None:
        raise ValueError("graphed_training: edge_weight without edge_index")
    with torch.no_grad():
        model(*args)                      # loads the library, builds and caches the CSR outside the capture
    graphed = torch.cuda.make_graphed_callables(model, args, num_warmup_iters=max(3, int(warmup)))
    # the captured kernels hold raw pointers into the CSR (and its adjoint, built by the warm-up backward passes)
    graphed._difformer_csr = [v[2] for v in ops.csr_cache.entries.values()] if edge_index is not None else []
    graphed.train(was_training)
    return graphed


class GraphedTrainStep:
    """One optimisation step -- forward, loss, backward, optimiser update (main.py:117-131) -- captured as ONE hipGraph.

        step = difformer_amd.GraphedTrainStep(model, optimizer, lambda out: F.nll_loss(F.log_softmax(out, 1)[idx], y[idx]),
                                              x, edge_index)
        for epoch in range(epochs):
            loss = step()                  # replays; `loss` is the captured scalar tensor (read it with float(loss))

    The optimiser must be capturable (torch.optim.Adam(..., capturable=True), SGD, ...) and `loss_fn` must only enqueue
    device work (no .item() / host syncs); operands keep their shapes and the graph its edges.  A Cora-sized step: 1.8-2.0 ms
    kernel by kernel, 0.9 ms with graphed_training, 0.67 ms this way."""

    def __init__(self, model, optimizer, loss_fn, x, edge_index=None, edge_weight=None, warmup=3):
        if not x.is_cuda:
            raise RuntimeError("difformer_amd: operands must live on the MI355X (no CPU fallback)")
        if any(getattr(c, "row_shard", None) is not None and c.row_shard.world > 1 for c in getattr(model, "convs", [])):
            raise NotImplementedError("GraphedTrainStep: row-sharded (multi-GPU) steps are not captured")
        model.train()
        self.model, self.optimizer = model, optimizer
        self.x, self.edge_index, self.edge_weight = x, edge_index, edge_weight
        dev = x.device

        def one_step():
            out = model(self.x, edge_index, edge_weight)
            loss = loss_fn(out)
            loss.backward()
            optimizer.step()
            return loss

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(3, int(warmup))):          # builds the CSR, its adjoint and the optimiser state
                optimizer.zero_grad(set_to_none=True)
                one_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        self._csr = [v[2] for v in ops.csr_cache.entries.values()] if edge_index is not None else []
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)             # gradients are re-created inside the capture's memory pool
        with torch.cuda.graph(self.graph):
            self.loss = one_step()

    def __call__(self, x=None):
        if x is not None and x is not self.x:
            if x.shape != self.x.shape:
                raise ValueError(f"GraphedTrainStep was captured for x of shape {tuple(self.x.shape)}, got {tuple(x.shape)}")
            self.x.copy_(x)
        self.graph.replay()
        return self.loss
