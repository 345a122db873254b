"""Drop-in replacement for `physical particle/difformer-v2.py` (imported there as `from difformer import DIFFormer_v2`,
parse.py:3) on MI355X: batches of small graphs stored back to back, `forward(x, edge_index, n_nodes)`.

Same public names, constructor arguments, defaults and `state_dict` keys.  The reference pads every graph to the
largest one ([B, max_node, H, D], :8-27, :87-91); here the per-graph attention runs unpadded inside one HIP kernel
(csrc/batched_attn.hip, csrc/sigmoid_attn.hip) and the rest of the layer reuses the kernels of `difformer.py`.
No CPU path: tensors must be float32 on the GPU.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from . import autograd_ops as ag
from . import staging
from .difformer import gcn_conv

__all__ = ["make_batch_mask", "make_batch", "to_pad", "gcn_conv", "TransConv", "DIFFormer_v2"]


# ---- the reference's padding helpers (:8-27).  The kernels do not need them; kept because they are public names of
# ---- the module.  Vectorised (the reference loops over graphs in Python), same results.
def make_batch_mask(n_nodes, device='cpu'):
    n_nodes = torch.as_tensor(n_nodes)
    max_node = int(n_nodes.max().item())
    mask = torch.arange(max_node, device=n_nodes.device).unsqueeze(0) < n_nodes.reshape(-1, 1)
    return mask.to(device), max_node


def make_batch(n_nodes, device='cpu'):
    n_nodes = torch.as_tensor(n_nodes).reshape(-1)
    return torch.repeat_interleave(torch.arange(n_nodes.numel(), device=n_nodes.device), n_nodes).long().to(device)


def to_pad(feat, mask, max_node, batch_size):
    n_heads, model_dim = feat.shape[-2:]
    new_feat = torch.zeros((batch_size, max_node, n_heads, model_dim)).to(feat)
    new_feat[mask] = feat
    return new_feat


class TransConv(nn.Module):
    """One propagation layer over a batch of graphs (reference: difformer-v2.py:47-159)."""

    def __init__(self, in_channels, out_channels, num_heads=1, kernel='simple', use_graph=True, use_weight=True,
                 graph_weight=-1):
        super().__init__()
        self.Wk = nn.Linear(in_channels, out_channels * num_heads)     # creation order as in the reference
        self.Wq = nn.Linear(in_channels, out_channels * num_heads)
        if use_weight:
            self.Wv = nn.Linear(in_channels, out_channels * num_heads)
        self.out_channels = out_channels
        self.num_heads = num_heads
        self.kernel = kernel
        self.use_graph = use_graph
        self.use_weight = use_weight
        self.graph_weight = graph_weight

    def __getstate__(self):
        state = super().__getstate__() if hasattr(super(), "__getstate__") else self.__dict__.copy()
        state = dict(state)
        for k in ("_cat_key", "_cat_w", "_cat_b"):
            state.pop(k, None)
        return state

    def reset_parameters(self):
        self.Wk.reset_parameters()
        self.Wq.reset_parameters()
        if self.use_weight:
            self.Wv.reset_parameters()

    def invalidate_caches(self):
        """Drop the cached concatenation of the projections (needed after parameter writes through `.data`, which do not
        bump the version counter the cache is keyed on)."""
        self._cat_key = None

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_caches()
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_caches()
        return out

    def _project(self, query_input, source_input):
        """Wq / Wk / Wv (:143-146).  Same input and nothing to differentiate: ONE narrow-Linear launch over the
        concatenated weights (x read once); q, k, v are then column slices of its [n, 3*H*D] result."""
        H, D = self.num_heads, self.out_channels
        params = (self.Wq.weight, self.Wq.bias, self.Wk.weight, self.Wk.bias, self.Wv.weight, self.Wv.bias)
        if query_input is source_input and source_input.shape[1] <= 64 and not ag._needs_grad(source_input, *params):
            key = ops.param_key(params)
            if key is None or getattr(self, "_cat_key", None) != key:
                self._cat_w = torch.cat([self.Wq.weight, self.Wk.weight, self.Wv.weight], dim=0).detach().contiguous()
                self._cat_b = torch.cat([self.Wq.bias, self.Wk.bias, self.Wv.bias], dim=0).detach().contiguous()
                self._cat_key = key
            qkv = ops.linear(source_input, self._cat_w, self._cat_b)
            return (qkv[:, : H * D].reshape(-1, H, D), qkv[:, H * D: 2 * H * D].reshape(-1, H, D),
                    qkv[:, 2 * H * D:].reshape(-1, H, D))
        return (ag.linear(query_input, self.Wq.weight, self.Wq.bias).reshape(-1, H, D),
                ag.linear(source_input, self.Wk.weight, self.Wk.bias).reshape(-1, H, D),
                ag.linear(source_input, self.Wv.weight, self.Wv.bias).reshape(-1, H, D))

    def full_attention(self, qs, ks, vs, kernel, n_nodes):
        """qs, ks, vs [N,H,D]; n_nodes [B] -> [N,H,D]  (:71-137)."""
        return ag.batched_attention(qs, ks, vs, ops.layout_cache.get(n_nodes, qs.device), kernel)

    def _layer(self, query_input, source_input, n_nodes, edge_index, edge_weight, prev=None, alpha=0.5, ln_weight=None,
               ln_bias=None, eps=1e-5, relu=False):
        H, D = self.num_heads, self.out_channels
        if not self.use_weight:
            # difformer-v2.py:145-148: `value` is only bound under use_weight, the reference dies with
            # UnboundLocalError right here
            raise UnboundLocalError("TransConv needs use_weight=True: difformer-v2.py:145-148 leaves `value` unbound "
                                    "otherwise")
        q, k, v = self._project(query_input, source_input)                               # :143-146
        attn = self.full_attention(q, k, v, self.kernel, n_nodes)                        # :148
        if not self.use_graph:
            return ag.layer_tail(attn, None, prev, alpha, ln_weight, ln_bias, eps, relu)
        if edge_index is None:
            raise ValueError("use_graph=True needs an edge_index")
        csr = ops.csr_cache.get(edge_index, edge_weight, v.shape[0], H * D * v.element_size(), elem_size=v.element_size())
        if self.graph_weight > 0:                                                        # :151-152
            a_s, g_s = 1.0 - self.graph_weight, float(self.graph_weight)
        else:                                                                            # :154
            a_s, g_s = 1.0, 1.0
        g_s *= csr.weight_scale                                                          # a constant edge_weight
        return ag.gcn_aggregate_tail(csr, v, attn, a_s, g_s, None, None, prev, alpha, ln_weight, ln_bias, eps, relu)

    def forward(self, query_input, source_input, n_nodes, edge_index=None, edge_weight=None):
        return self._layer(query_input, source_input, n_nodes, edge_index, edge_weight)  # head mean, :157


class DIFFormer_v2(nn.Module):
    """Reference: difformer-v2.py:161-223.  x [N, in_channels] (all graphs of the batch back to back),
    edge_index [2, E] int64 with batch-global node ids, n_nodes [B] -> [N, out_channels]."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers=2, kernel='simple', alpha=0.5, dropout=0.5,
                 use_bn=True, use_residual=True, use_weight=True, use_graph=True, graph_weight=-1):
        super().__init__()
        self.convs = nn.ModuleList()
        self.fcs = nn.ModuleList()
        self.fcs.append(nn.Linear(in_channels, hidden_channels))
        self.bns = nn.ModuleList()
        self.bns.append(nn.LayerNorm(hidden_channels))
        for _ in range(num_layers):
            self.convs.append(TransConv(hidden_channels, hidden_channels, kernel=kernel, use_graph=use_graph,
                                        use_weight=use_weight, graph_weight=graph_weight))
            self.bns.append(nn.LayerNorm(hidden_channels))
        self.fcs.append(nn.Linear(hidden_channels, out_channels))
        self.dropout = dropout
        self.activation = F.relu
        self.use_bn = use_bn
        self.residual = use_residual
        self.alpha = alpha

    def reset_parameters(self):
        for conv in self.convs:
            conv.reset_parameters()
        for bn in self.bns:
            bn.reset_parameters()
        for fc in self.fcs:
            fc.reset_parameters()

    def __getstate__(self):
        state = super().__getstate__() if hasattr(super(), "__getstate__") else self.__dict__.copy()
        state = dict(state)
        state.pop("_staged", None)
        return state

    def invalidate_caches(self):
        for conv in self.convs:
            conv.invalidate_caches()
        staging.drop(self)

    def forward(self, x, edge_index, n_nodes):
        dev = staging.staging_device(self, (x, edge_index))
        if dev is not None:      # model and batch in host memory: staged onto the GPU, result back on the host
            return staging.staged_forward(self, dev, lambda m, *a: m.forward(*a), x, edge_index, n_nodes)
        layer_ = []
        bn = self.bns[0] if self.use_bn else None                                        # :197-200 in one kernel
        x = ag.linear(x, self.fcs[0].weight, self.fcs[0].bias, bn.weight if bn is not None else None,
                      bn.bias if bn is not None else None, bn.eps if bn is not None else 1e-5, relu=True)
        x = F.dropout(x, p=self.dropout, training=self.training)                         # :201
        layer_.append(x)
        for i, conv in enumerate(self.convs):
            bn = self.bns[i + 1] if self.use_bn else None
            lnw, lnb, eps = (bn.weight, bn.bias, bn.eps) if bn is not None else (None, None, 1e-5)
            prev = layer_[i] if self.residual else None
            if self.training and self.dropout > 0:
                # :212-217: residual -> norm -> dropout -> ReLU; dropout sits between norm and ReLU
                x = conv._layer(x, x, n_nodes, edge_index, None, prev, self.alpha, lnw, lnb, eps, relu=False)
                x = self.activation(F.dropout(x, p=self.dropout, training=True))
            else:
                x = conv._layer(x, x, n_nodes, edge_index, None, prev, self.alpha, lnw, lnb, eps, relu=True)
            layer_.append(x)
        x_out = ag.linear(x, self.fcs[-1].weight, self.fcs[-1].bias)                     # :221
        return F.dropout(x_out, p=self.dropout, training=self.training)                  # :222
