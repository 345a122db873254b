"""Device-side versions of the graph preprocessing calls the reference's drivers make around the model
(SURVEY.md section 8f rank 2).  Same argument order and return values as the torch_geometric.utils functions the
scripts import, so a driver can switch by changing one import; all of them keep the graph on the GPU.

    subgraph          node classification/main-batch.py:131   (HIP kernels: mark, flag, scan, emit)
    add_self_loops    main.py:76, main-batch.py:98            (tensor plumbing)
    remove_self_loops main.py:75, main-batch.py:97            (tensor plumbing)
    to_undirected     main.py:73                              (tensor plumbing: both directions, duplicates coalesced)
"""
from __future__ import annotations

import torch

from . import ops


def subgraph(subset, edge_index, edge_attr=None, relabel_nodes=False, num_nodes=None):
    """Edges induced by `subset` (int64 node ids or a boolean mask), in their original order.
    relabel_nodes=True renumbers subset[i] -> i (what main-batch.py:131 asks for)."""
    n = int(num_nodes) if num_nodes is not None else int(edge_index.max().item()) + 1
    if subset.dtype == torch.bool:
        subset = subset.nonzero(as_tuple=False).view(-1)
    ei, ew = ops.get_backend().subgraph(subset, edge_index, edge_attr, n)
    if not relabel_nodes:
        ei = subset[ei]
    return ei, ew


def add_self_loops(edge_index, edge_weight=None, fill_value=1.0, num_nodes=None):
    n = int(num_nodes) if num_nodes is not None else int(edge_index.max().item()) + 1
    loops = torch.arange(n, device=edge_index.device, dtype=edge_index.dtype).repeat(2, 1)
    if edge_weight is not None:
        edge_weight = torch.cat([edge_weight, edge_weight.new_full((n,), fill_value)])
    return torch.cat([edge_index, loops], dim=1), edge_weight


def remove_self_loops(edge_index, edge_attr=None):
    keep = edge_index[0] != edge_index[1]
    return edge_index[:, keep], (None if edge_attr is None else edge_attr[keep])


def to_undirected(edge_index, num_nodes=None):
    """Both directions of every edge, duplicates removed, sorted by (row, col)."""
    n = int(num_nodes) if num_nodes is not None else int(edge_index.max().item()) + 1
    both = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    key = torch.unique(both[0] * n + both[1])
    return torch.stack([key // n, key % n])
