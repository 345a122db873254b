"""Device-side versions of the graph preprocessing calls the reference's drivers make around the model
(SURVEY.md section 8f rank 2).  Same argument order and return values as the torch_geometric.utils functions the
scripts import, so a driver can switch by changing one import; all of them keep the graph on the GPU.

    subgraph          node classification/main-batch.py:131   (HIP kernels: mark, flag, scan, emit)
    subgraph_batches  main-batch.py:121-131, all batches of an epoch in one pass (HIP: mark, key, stable radix pass, emit)
    add_self_loops    main.py:76, main-batch.py:98            (HIP: dif_graph_prepare)
    remove_self_loops main.py:75, main-batch.py:97            (HIP: dif_graph_prepare; mask when attributes ride along)
    to_undirected     main.py:73                              (HIP: pairs in both directions, two stable radix sorts, coalesce)
    prepare_graph     main.py:72-76 in one call
"""
from __future__ import annotations

import weakref

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


class _LongestRows:
    """Longest row of every batch of ONE subgraph_batches call, on its way to the host: a device reduction and a copy into
    pinned memory are enqueued behind the CSR sort, and the first batch that needs its value WAITS for the copy (by then it
    has landed: the wait is a completed-event check) and keeps the values of all batches.  The kernel choice that depends
    on the statistic is therefore the same on every run; nothing polls, and the epoch has no extra host synchronisation in
    front of its first forward."""
    _pinned = None          # one pinned buffer for the process (a pinned allocation costs ~0.1 ms), re-used per epoch ...
    _in_flight = None       # ... once the previous epoch's values have been taken out of it

    def __init__(self, per_batch):
        n = int(per_batch.numel())
        self.values = None
        if not per_batch.is_cuda:
            self.values = per_batch.tolist()
            return
        cls = _LongestRows
        if cls._in_flight is not None:
            cls._in_flight.resolve()
        if cls._pinned is None or cls._pinned.numel() < n:
            cls._pinned = torch.empty(max(n, 1024), dtype=torch.int32, pin_memory=True)
        self.n, self.host = n, cls._pinned
        self.host[:n].copy_(per_batch.to(torch.int32), non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(per_batch.device))
        cls._in_flight = self

    def resolve(self):
        if self.values is None:
            self.event.synchronize()
            self.values = self.host[: self.n].tolist()
            if _LongestRows._in_flight is self:
                _LongestRows._in_flight = None
        return self.values


def subgraph_batches(perm, batch_size, edge_index, edge_attr=None, num_nodes=None, build_csr=True):
    """Every mini-batch subgraph of an epoch from ONE pass over the edge list (csrc/gcn_csr.hip, dif_subgraph_batches_*).

    main-batch.py:121-131 cuts a permutation of the training nodes into batches and calls
    `subgraph(idx_i, edge_index, num_nodes=n, relabel_nodes=True)` once per batch; with
        batches = subgraph_batches(train_idx[perm], batch_size, edge_index, num_nodes=n)
    before the loop, `edge_index_i, _ = batches[i]` returns the identical tensors (same edges, same order, same
    relabelling) without touching the edge list again.  Returns a list of (edge_index_b [2, E_b], edge_attr_b | None).
    build_csr: the normalised CSR of every batch comes out of one more sort of the surviving edges and is registered in
    the CSR cache under its batch's edge_index, so `model(x_i, edge_index_i)` builds nothing (keep the returned list
    alive for the epoch: the registration follows the tensors' lifetime)."""
    n = int(num_nodes) if num_nodes is not None else int(edge_index.max().item()) + 1
    ei, ew, ptr, csr = ops.get_backend().subgraph_batches(perm, batch_size, edge_index, edge_attr, n, build_csr)
    nb = len(ptr) - 1
    out = [(ei[:, ptr[b]: ptr[b + 1]], None if ew is None else ew[ptr[b]: ptr[b + 1]]) for b in range(nb)]
    if csr is not None:
        rowptr, src, val = csr
        m, bs = int(perm.numel()), int(batch_size)
        ops.csr_cache.reserve(nb + ops.csr_cache.capacity)
        # longest row of every batch (the layer kernel's choice of path): one reduction per EPOCH, read when first needed
        deg = rowptr[1:] - rowptr[:-1]
        longest = None
        if m > 0:
            longest = _LongestRows(torch.nn.functional.pad(deg, (0, nb * bs - m)).view(nb, bs).max(dim=1).values)
        for b, (eb, wb) in enumerate(out):
            lo, hi = b * bs, min((b + 1) * bs, m)
            if ops.csr_cache.blocking(eb, wb, hi - lo) != (1, 0):
                continue        # a dense batch: `get` builds the tiled CSR of the sliced product itself (different key)
            rp = (rowptr[lo: hi + 1] - ptr[b]).contiguous()
            e0, e1 = ptr[b], ptr[b + 1]
            g = ops.GraphCSR(rp, None, 1, src[e0: max(e1, e0 + 1)], val[e0: max(e1, e0 + 1)], hi - lo, e1 - e0)
            g.weighted = wb is not None
            g._edges = (weakref.ref(eb), None if wb is None else weakref.ref(wb))
            if longest is None:
                g._max_degree = 0
            else:
                g._max_source = (longest, b)
            g._format_checked = True            # a sparse batch by construction (`blocking` above): nothing to build at first use
            ops.csr_cache.put(eb, wb, hi - lo, g)
    return out


def _n(edge_index, num_nodes):
    """Node count: the caller's, else max id + 1 (one host sync, as torch_geometric's maybe_num_nodes; 0 for no edges)."""
    if num_nodes is not None:
        return int(num_nodes)
    return int(edge_index.max().item()) + 1 if edge_index.numel() else 0


def add_self_loops(edge_index, edge_weight=None, fill_value=1.0, num_nodes=None):
    """main.py:76, main-batch.py:98: the N loops appended (HIP: dif_graph_prepare); weights ride along as tensor plumbing."""
    n = _n(edge_index, num_nodes)
    if edge_weight is not None:
        edge_weight = torch.cat([edge_weight, edge_weight.new_full((n,), fill_value)])
    if edge_index.numel() == 0:
        loops = torch.arange(n, device=edge_index.device, dtype=edge_index.dtype)
        return torch.stack([loops, loops]), edge_weight
    return ops.get_backend().graph_prepare(edge_index, n, add_loops=True), edge_weight


def remove_self_loops(edge_index, edge_attr=None):
    """main.py:75, main-batch.py:97: edges (v, v) dropped, order kept (HIP: dif_graph_prepare).  With attributes the
    filter is a mask (the attributes have to follow the same selection)."""
    if edge_attr is not None or edge_index.numel() == 0:
        # a mask needs no node count (no host sync) and handles the empty edge list
        keep = edge_index[0] != edge_index[1]
        return edge_index[:, keep], (None if edge_attr is None else edge_attr[keep])
    return ops.get_backend().graph_prepare(edge_index, _n(edge_index, None), remove_loops=True), None


def to_undirected(edge_index, num_nodes=None):
    """main.py:73: both directions of every edge, duplicates removed, sorted by (row, col) (HIP: dif_graph_prepare)."""
    if edge_index.numel() == 0:
        return edge_index
    return ops.get_backend().graph_prepare(edge_index, _n(edge_index, num_nodes), undirected=True)


def prepare_graph(edge_index, num_nodes=None):
    """main.py:72-76 in one call: to_undirected -> remove_self_loops -> add_self_loops."""
    return ops.get_backend().graph_prepare(edge_index, _n(edge_index, num_nodes), True, True, True)
