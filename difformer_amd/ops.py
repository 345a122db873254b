"""Operators of the DIFFormer propagation layer on MI355X (shard-aware, no arithmetic of their own).

    simple_attention / sigmoid_attention  <- full_attention_conv   (difformer.py:10-61)
    GraphCSR + gcn_aggregate              <- gcn_conv              (difformer.py:63-79)
    layer_tail                            <- difformer.py:137-140, :200-203

Each function drives the HIP kernels through `backend_hip.HipBackend`; with a `RowShard` it
adds the single exchange step the operator needs (dist.py).
"""
from __future__ import annotations

import weakref
from collections import OrderedDict
from typing import Optional

import math
import os

import torch

from .dist import RowShard

_BACKEND = None


def tensor_version(t):
    """`t._version`, or -1 for tensors that do not track one (created under torch.inference_mode())."""
    try:
        return t._version
    except RuntimeError:
        return -1


def param_key(params):
    """Identity + version key of a list of parameters for the inference-time caches (concatenated projections,
    weight-only factors of the closed form, float32 copies), or None when a version cannot be read (inference tensors):
    the caller then rebuilds instead of caching.  NOTE: writes through `.data` (`p.data.copy_()`, EMA / weight averaging
    done on `.data`) do NOT bump the version counter -- call `model.invalidate_caches()` after such an update
    (`load_state_dict` and `.to()` / `.half()` / ... do it themselves)."""
    key = []
    for t in params:
        v = tensor_version(t)
        if v < 0:
            return None
        key.append((t.data_ptr(), v, t.dtype, t.device))
    return tuple(key)


def get_backend():
    """The HIP backend (created on first use; import fails loudly if the .so is missing)."""
    global _BACKEND
    if _BACKEND is None:
        from .backend_hip import HipBackend
        _BACKEND = HipBackend()
    return _BACKEND


# ------------------------------------------------------------------------------------------
# a1 / a2
# ------------------------------------------------------------------------------------------
def simple_attention(qs, ks, vs, shard: Optional[RowShard] = None):
    """qs, ks [n,H,M], vs [n,H,D] (this rank's rows) -> [n,H,D].  difformer.py:18-39."""
    if qs.shape[0] != vs.shape[0] or ks.shape[0] != vs.shape[0]:
        # difformer.py:29 adds a [L,H,D] tensor to a [N,H,D] one: the reference raises too
        raise RuntimeError(f"simple kernel needs as many queries as sources (N={qs.shape[0]}, L={vs.shape[0]}; "
                           "difformer.py:29)")
    be = get_backend()
    reduced = be.simple_reduce(qs, ks, vs)
    n_global = qs.shape[0]
    if shard is not None and shard.world > 1:
        shard.all_reduce_sum(reduced)          # the one exchange step: H*(M*D+M+D)+2 floats
        n_global = shard.n_global
    return be.simple_apply(qs, reduced, n_global, vs.shape[2])


def project_simple_attention(x, Wq, bq, Wk, bk, Wv, bv, H, D, shard: Optional[RowShard] = None,
                             gather_values=False):
    """x [n,C] (local rows) -> (attn [n,H,D], v [n,H,D]).  Projections (difformer.py:115-118) fused with
    stage 1 of the simple kernel; k never reaches HBM.  Needs C <= 64 and D <= 64."""
    be = get_backend()
    q, v, reduced = be.project_reduce(x, Wq, bq, Wk, bk, Wv, bv, H, D)
    n_global = x.shape[0]
    if shard is not None and shard.world > 1:
        n_global = shard.n_global
        if gather_values:
            # start moving the value rows now, and the record all-reduce on the second communicator; the part of the
            # SpMM that needs only this rank's own rows runs under both (gcn_aggregate), then `apply`, then the rest
            v = GatheredRows(v, shard.all_gather_rows_async(v.reshape(v.shape[0], H * D)))
            work = shard.all_reduce_sum_async(reduced)

            def finish():
                if work is not None:
                    work.wait()
                return be.simple_apply(q, reduced, n_global, D)
            return LazyAttention(finish, q.shape[0], H, D), v
        shard.all_reduce_sum(reduced)
    return be.simple_apply(q, reduced, n_global, D), v


class LazyAttention:
    """The `simple` attention of a row-sharded layer whose record all-reduce is still in flight: materialize() waits
    for it and runs the apply kernel.  gcn_aggregate calls it after part 0 of the SpMM, which needs neither."""

    def __init__(self, finish, n, H, D):
        self._finish, self._value = finish, None
        self.shape = (n, H, D)

    def materialize(self):
        if self._value is None:
            self._value = self._finish()
            self._finish = None
        return self._value


class GatheredRows:
    """Local value rows [n,H,D] plus the in-flight all-gather of them (row-sharded runs)."""

    def __init__(self, local, handle):
        self.local, self.handle = local, handle
        self.shape = local.shape


def sigmoid_attention(qs, ks, vs, shard: Optional[RowShard] = None):
    """qs [n,H,M] local queries; ks, vs local sources -> [n,H,D].  difformer.py:45-56."""
    if shard is not None and shard.world > 1:
        ks = shard.all_gather_rows(ks)
        vs = shard.all_gather_rows(vs)
    return get_backend().sigmoid_attention(qs, ks, vs)


# ------------------------------------------------------------------------------------------
# a3
# ------------------------------------------------------------------------------------------
L2_SLICE_BYTES = 2.5 * 2 ** 20   # target size of the x slice one source block covers (L2 is 4 MiB / XCD)


def choose_source_blocks(num_nodes, row_bytes, nnz):
    """Source blocks for the blocked SpMM: 1 (plain CSR) when x fits L2 or rows are too sparse for
    the per-(row, block) groups to amortise; otherwise ~2.5 MiB of x per block, at most 64."""
    total = float(num_nodes) * float(row_bytes)
    if total <= 4 * 2 ** 20 or nnz < 64 * num_nodes:
        return 1
    nb = int(round(total / L2_SLICE_BYTES))
    # the sweep prefetches up to 64 entries of a (row, block) group; longer groups fall off that fast path
    nb = max(nb, -(-int(nnz) // (int(num_nodes) * 48)))
    nb = max(2, min(nb, 64))
    # keep >= ~24 entries per (row, block) group on average
    while nb > 2 and nnz / (num_nodes * nb) < 24:
        nb -= 1
    return nb


def choose_shard_blocks(num_nodes, row_bytes, nnz, shard):
    """(n_blocks, block_rows) whose block boundaries fall on the rank boundaries of `shard`, or None.  The blocked SpMM
    of a row-sharded run can then sweep the blocks holding this rank's OWN value rows while the all-gather of the
    others is still in flight (gcn_aggregate).  Every rank's c rows (a multiple of 8, dist.split_rows) are cut into
    m in {1, 2, 4, 8} blocks of c / m rows, ~2.5 MiB of x each."""
    if shard is None or shard.world <= 1 or choose_source_blocks(num_nodes, row_bytes, nnz) <= 1:
        return None
    c = shard.counts[0]
    if any(shard.offsets[r] != r * c for r in range(shard.world)) or c % 8 != 0:
        return None
    for m in (1, 2, 4, 8):
        rows = c // m
        n_blocks = -(-int(num_nodes) // rows)
        if rows * row_bytes <= 1.15 * L2_SLICE_BYTES and n_blocks <= 64:
            # the sweep prefetches up to 64 entries of a (row, block) group: keep the average group at <= ~48 entries
            if nnz / (float(num_nodes) * n_blocks) <= 48 or m == 8:
                return n_blocks, rows
    return None


SLICED_MIN_DEGREE = 48      # entries per row from which the LDS-staged product beats the gather kernels
SLICED_MIN_ROWS = 8192


def sliced_tiling(num_nodes, F, nnz, edge_weight, shard, elem_size):
    """(n_tiles, tile_rows) the feature-sliced product would use for this graph, or None when it is not a candidate
    (edge weights, non-fp32 rows, sparse or small graph).  csr_cache builds the CSR with that blocking.  A row shard
    has its own tiling (source splits: a multiple of 2, 4 or 8 tiles, include/difformer_hip.h)."""
    if edge_weight is not None or elem_size != 4:
        return None
    if F % 4 or F > 256 or num_nodes < SLICED_MIN_ROWS or nnz < SLICED_MIN_DEGREE * num_nodes:
        return None
    be = get_backend()
    n_rows = shard.n_local if (shard is not None and shard.world > 1) else num_nodes
    plan = be.sliced_plan(num_nodes, n_rows, F) if hasattr(be, "sliced_plan") else None
    return None if plan is None else (int(plan[7]), int(plan[6]))


SLICED_SPLIT_FACTOR = 4.0   # rows beyond this multiple of the mean degree are split into lock-step parts (Zipf C4: 2-6 within 4 %)
SLICED_MAX_PARTS = 64       # one slot


class SlicedAdjacency:
    """Entry blocks + table + geometry of the feature-sliced product (include/difformer_hip.h, dif_sliced_*).
    order: rows by descending degree (skewed graphs); parts / n_pos: hub rows split into lock-step parts ("row
    positions" in the header) -- then order has n_pos entries (-1 = padding) and plan is the geometry of n_pos positions;
    The slice-major source copy is written with the same plan (its tiling)."""

    def __init__(self, plan, entries, table, order=None, parts=None, n_pos=None):
        self.plan, self.entries, self.table, self.order = plan, entries, table, order
        self.parts, self.n_pos = parts, n_pos


def split_positions(deg_sorted, order, cap, max_parts=SLICED_MAX_PARTS):
    """Row positions for rows sorted by descending degree `deg_sorted` (their ids in `order`): a row of degree d takes
    P = min(ceil(d / cap), max_parts) consecutive positions of ONE 64-position slot.  Rows with the same P are packed
    64 // P to a slot (every P-class starts on a fresh slot), the unsplit rows follow in order.
    -> (order2 int32 [n_pos] with -1 = empty, parts uint16-as-int16 [n_pos] = p | P << 8, n_pos).  Device tensor ops;
    one host sync for the sizes (cold path)."""
    dev = deg_sorted.device
    d = deg_sorted.to(torch.int64)
    P = torch.clamp((d + cap - 1) // cap, 1, max_parts)
    n_hub = int((P > 1).sum())                                   # a prefix: the degrees are sorted
    n = d.numel()
    Ph = P[:n_hub]
    vals, counts = torch.unique_consecutive(Ph, return_counts=True)
    per_slot = 64 // vals
    class_slots = (counts + per_slot - 1) // per_slot
    class_base = torch.cumsum(class_slots, 0) - class_slots      # first slot of every class
    class_first = torch.cumsum(counts, 0) - counts               # first hub row of every class
    cls = torch.repeat_interleave(torch.arange(vals.numel(), device=dev), counts)
    k = torch.arange(n_hub, device=dev) - class_first[cls]
    pos0 = (class_base[cls] + k // per_slot[cls]) * 64 + (k % per_slot[cls]) * Ph
    hub_slots = int(class_slots.sum())
    n_pos = hub_slots * 64 + (n - n_hub)
    order2 = torch.full((n_pos,), -1, dtype=torch.int32, device=dev)
    parts = torch.full((n_pos,), 0x0100, dtype=torch.int16, device=dev)
    row_of_part = torch.repeat_interleave(torch.arange(n_hub, device=dev), Ph)
    p_idx = torch.arange(row_of_part.numel(), device=dev) - (torch.cumsum(Ph, 0) - Ph)[row_of_part]
    pos = pos0[row_of_part] + p_idx
    order2[pos] = order[:n_hub][row_of_part].to(torch.int32)
    parts[pos] = (p_idx + (Ph[row_of_part] << 8)).to(torch.int16)
    order2[hub_slots * 64:] = order[n_hub:].to(torch.int32)
    return order2, parts, n_pos


class GraphCSR:
    """Normalised adjacency in CSR over destination rows (built once per graph, on device)."""

    def __init__(self, rowptr, blkptr, n_blocks, src, val, num_nodes, nnz, block_rows=0):
        self.rowptr, self.blkptr, self.n_blocks, self.src, self.val = rowptr, blkptr, int(n_blocks), src, val
        self.num_nodes, self.nnz = int(num_nodes), int(nnz)
        self.block_rows = int(block_rows) or -(-int(num_nodes) // int(n_blocks))     # source rows per block
        self._edges = None          # weak references to (edge_index, edge_weight) for the lazily built adjoint
        self._adjoint = None
        self._orders = {}           # (row_begin, n_rows) -> rows by descending degree (blocked SpMM load balance)
        self.weighted = self.transposed = False
        self.dinv = None            # adjoint CSR only: deg^-1/2 of the FORWARD graph (its own row lengths are out-degrees)
        self._sliced = {}           # (row_begin, n_rows, F) -> SlicedAdjacency | None
        self._row_sums = None
        self._format_checked = False # csr_cache.get(build_format=True) has built (or ruled out) the sliced format for this CSR
        self.weight_scale = 1.0      # a CONSTANT edge_weight (every entry equal: `--special_treat dense` / knn, spatial-temporal/
                                     # main.py:99,103) is not in `val`: the CSR is the unweighted one and every caller multiplies
                                     # its gcn_scale by this (csr_cache.get)
        self._max_degree = None      # longest row: known from the build (status[1] of dif_csr_build) ...
        self._max_source = None      # ... or (graph_utils._LongestRows, batch) for the batches of one subgraph_batches call

    def row_order(self, row_begin, n_rows):
        """(order, n_split) for the blocked SpMM over a shard: its rows by descending degree, the first n_split of them
        (degree > 4x mean) to be split over a whole quad.  Built once per shard on the device (one host sync for the two
        statistics).  None when the order would not help: unblocked kernels (they map rows to waves / lane groups one by
        one), or degrees about equal (max <= 2x mean: contiguous panels are faster then)."""
        if self.n_blocks <= 1 or self.nnz == 0:
            return None
        key = (int(row_begin), int(n_rows))
        if key not in self._orders:
            order, stats = get_backend().row_order(self.rowptr, *key)
            n_split, max_deg = (int(v) for v in stats.tolist())
            total = int(self.rowptr[key[0] + key[1]]) - int(self.rowptr[key[0]])
            skewed = max_deg * key[1] > 2 * total
            self._orders[key] = (order, n_split) if skewed else None
        return self._orders[key]

    @classmethod
    def build(cls, edge_index, edge_weight, num_nodes, n_blocks=1, transpose=False, block_rows=0):
        rowptr, blkptr, src, val, longest = get_backend().csr_build(edge_index, edge_weight, int(num_nodes), int(n_blocks),
                                                                    transpose, int(block_rows))
        csr = cls(rowptr, blkptr, n_blocks, src, val, num_nodes, edge_index.shape[1], block_rows)
        csr._max_degree = int(longest)
        csr.weighted, csr.transposed = edge_weight is not None, bool(transpose)
        if not transpose:
            csr._edges = (weakref.ref(edge_index), None if edge_weight is None else weakref.ref(edge_weight))
        return csr

    def sliced(self, row_begin, n_rows, F):
        """The feature-sliced LDS format of rows [row_begin, row_begin + n_rows) for F fp32 feature columns
        (csrc/gcn_sliced.hip), built on first use -- or None when this graph keeps the gather kernels: edge weights,
        an adjoint CSR, a blocking that is not the format's tiling, or a (row, tile) group beyond the 16-bit counters.
        With skewed degrees (max > 2x mean) the 64-row slots are formed in descending-degree order, so that the lock-step
        lanes of a slot carry lists of similar length."""
        key = (int(row_begin), int(n_rows), int(F))
        if key not in self._sliced:
            self._sliced[key] = self._build_sliced(*key)
        return self._sliced[key]

    def _build_sliced(self, row_begin, n_rows, F):
        if self.weighted or self.nnz == 0 or (self.transposed and self.dinv is None):
            return None
        be = get_backend()
        if self.num_nodes < SLICED_MIN_ROWS or self.nnz < SLICED_MIN_DEGREE * self.num_nodes or not hasattr(be, "sliced_plan"):
            return None
        plan = be.sliced_plan(self.num_nodes, n_rows, F)
        if plan is None:
            return None
        T, NT = int(plan[6]), int(plan[7])
        if NT != self.n_blocks or (NT > 1 and T != self.block_rows):
            return None
        deg = self.rowptr[row_begin + 1: row_begin + n_rows + 1] - self.rowptr[row_begin: row_begin + n_rows]
        max_deg, total = (int(v) for v in torch.stack([deg.max(), deg.sum()]).tolist())       # one sync, cold path
        order = parts = n_pos = None
        if max_deg * n_rows > 2 * total:
            order = be.row_order(self.rowptr, row_begin, n_rows)[0]
            cap = max(int(SLICED_SPLIT_FACTOR * total / n_rows), 64)
            if max_deg > cap:
                # hub rows run as several lock-step lanes of one slot instead of one long lane
                order, parts, n_pos = split_positions(deg[order.long()], order, cap)
                plan = be.sliced_plan(self.num_nodes, n_pos, F)
                if plan is None or int(plan[6]) != T or int(plan[7]) != NT:
                    return None
        built = be.sliced_build(self.rowptr, self.blkptr, self.src, self.num_nodes, self.nnz, row_begin, n_rows, F, plan, order,
                                parts, n_pos)
        if built is None:
            return None
        return SlicedAdjacency(plan, built[0], built[1], order, parts, n_pos)

    def row_sums(self):
        """A_hat 1 (float32 [N]): what the bias of the value projection turns into under the aggregation,
        A_hat (x Wv^T + 1 bv^T) = (A_hat x) Wv^T + (A_hat 1) bv^T.  One small product per graph, cached."""
        if self._row_sums is None:
            ones = torch.ones((self.num_nodes, 4), dtype=torch.float32, device=self.rowptr.device)
            be = get_backend()
            out = be.spmm(self.rowptr, self.blkptr, self.n_blocks, self.src, self.val, self.num_nodes, self.nnz, ones, 0,
                          self.num_nodes, None, 1.0, 1.0, None, self.row_order(0, self.num_nodes))
            self._row_sums = out[:, 0].contiguous()
        return self._row_sums

    def max_degree(self):
        """Longest row (entries).  `build` gets it from the CSR kernels with the host read a build has anyway
        (dif_csr_build status[1]) and graph_utils.subgraph_batches sets it for the batches it registers, so the kernel
        choice that depends on it (the closed-form layer kernel aggregates inside only when no row is long: its 16 rows of a
        tile walk in lock step) is a function of the graph alone -- identical calls launch identical kernels.  A CSR put
        together by hand pays one reduction and one host read here, once."""
        if self._max_degree is None and self._max_source is not None:
            rows, b = self._max_source
            self._max_degree, self._max_source = int(rows.resolve()[b]), None
        if self._max_degree is None:
            self._max_degree = int((self.rowptr[1:] - self.rowptr[:-1]).max().item()) if self.num_nodes > 0 else 0
        return self._max_degree

    def weight_leaf(self):
        """The edge_weight tensor this CSR was built from when the caller wants its gradient (difformer.py:73 is
        differentiable in it), else None."""
        if not self.weighted or not self._edges or self._edges[1] is None or not torch.is_grad_enabled():
            return None
        w = self._edges[1]()
        return w if (w is not None and w.requires_grad) else None

    def hold_edges(self):
        """Strong references to the tensors this CSR was built from (None if already freed).  The autograd node of the
        aggregation keeps them until backward(), so callers may pass temporaries (`model(x, ei.to(dev))`)."""
        if not self._edges:
            return None
        return (self._edges[0](), self._edges[1]() if self._edges[1] is not None else None)

    def adjoint(self):
        """CSR of A_hat^T (entries filed under their source row): the SpMM over it is the gradient of the
        aggregation.  Built on first use (training only) from the same edge tensors."""
        if self._adjoint is None:
            ei = self._edges[0]() if self._edges else None
            ew = self._edges[1]() if self._edges and self._edges[1] is not None else None
            if ei is None or (self._edges[1] is not None and ew is None):
                raise RuntimeError("difformer_amd: the edge_index this CSR was built from has been freed; keep it alive "
                                   "until backward()")
            self._adjoint = GraphCSR.build(ei, ew, self.num_nodes, self.n_blocks, transpose=True,
                                           block_rows=self.block_rows)
            if not self.weighted:
                # value_e = dinv[col] * dinv[row] with dinv from the in-degree (difformer.py:66-68): the adjoint product
                # factors the same way, grad_x[s] = dinv[s] * sum_{e: src = s} (dinv * g)[dst_e] -> the sliced kernel
                deg = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.float32)
                self._adjoint.dinv = torch.where(deg > 0, (1.0 / deg).sqrt(), torch.zeros_like(deg))
        return self._adjoint


class _CSRCache:
    """The reference rebuilds degree / values / SparseTensor in every layer of every forward
    (difformer.py:66-75).  We key the built CSR on the identity *and* version of the tensors the
    caller passes, so `forward(x, edge_index)` keeps its signature and in-place edits or new
    tensors still trigger a rebuild."""

    def __init__(self, capacity=8):
        self.capacity = capacity
        self.entries = OrderedDict()

    @staticmethod
    def _key(edge_index, edge_weight, num_nodes, n_blocks, block_rows=0):
        k = (id(edge_index), edge_index.data_ptr(), tuple(edge_index.shape), tensor_version(edge_index), int(num_nodes),
             int(n_blocks), int(block_rows), str(edge_index.device))
        if edge_weight is not None:
            k += (id(edge_weight), edge_weight.data_ptr(), tensor_version(edge_weight))
        return k

    def get(self, edge_index, edge_weight, num_nodes, row_bytes=256, shard=None, elem_size=4, build_format=True):
        """`row_bytes` = bytes of one feature row the SpMM will gather (H*D*elem_size); picks the blocking -- the
        tiling of the feature-sliced product for dense unweighted fp32 graphs, else ~2.5 MiB source blocks, aligned
        with the rank boundaries of `shard` when the run is row-sharded."""
        if edge_weight is not None and edge_weight.requires_grad and torch.is_grad_enabled() and shard is not None \
                and shard.world > 1:
            # the reference's gcn_conv is differentiable in edge_weight (value = w * d_in * d_out through
            # torch_sparse.matmul).  One GPU: autograd_ops._GcnAggregate returns that gradient (csrc/gcn_edge_grad.hip);
            # row-sharded it would need the gathered rows of both operands on every rank
            raise NotImplementedError("difformer_amd: gradients with respect to edge_weight are not implemented for "
                                      "row-sharded runs; detach() it or keep it a constant of the graph")
        self._purge()
        # every weight equal (one reduction, read with the host read a build has anyway): value_e = w d_in d_out
        # (difformer.py:73) = w x the unweighted value, so the graph takes every UNWEIGHTED kernel (feature-sliced product,
        # in-layer aggregation) and w rides in the callers' gcn_scale.  Keyed like a weighted graph, built like an unweighted one.
        given_weight, scale = edge_weight, 1.0
        if edge_weight is not None:
            u = self._uniform_weight(edge_weight, edge_index.shape[1], num_nodes)
            if u is not None:
                edge_weight, scale = None, u
        n_blocks, block_rows = self.blocking(edge_index, edge_weight, num_nodes, row_bytes, shard, elem_size)
        key = self._key(edge_index, given_weight, num_nodes, n_blocks, block_rows)
        hit = self.entries.get(key)
        csr = None
        if hit is not None:
            ei_ref, ew_ref, csr = hit
            if ei_ref() is edge_index and (given_weight is None or ew_ref() is given_weight):
                if edge_weight is not None and not csr.weighted:
                    # built through the uniform-weight shortcut (an eval pass before the first step of learnable weights
                    # initialised to a constant): this call wants the weighted CSR, whose values autograd differentiates
                    del self.entries[key]
                    csr = None
                else:
                    self.entries.move_to_end(key)
                    if csr._format_checked or not build_format:
                        return csr
            else:
                del self.entries[key]
                csr = None
        if csr is None:
            csr = GraphCSR.build(edge_index, edge_weight, num_nodes, n_blocks, block_rows=block_rows)
            csr.weight_scale = scale
        F = row_bytes // elem_size
        if build_format:
            # (also the first `build_format` request for a CSR that a pointers-only probe left here: _MixCache.get)
            csr._format_checked = True
        if build_format and sliced_tiling(num_nodes, F, edge_index.shape[1], edge_weight, shard, elem_size) is not None:
            # the blocking above is the tiling of the sliced product: build its format now, and if the graph turns out not
            # to fit it (a (row, tile) group beyond the 16-bit counters, a split plan with another tiling), fall back to a
            # CSR blocked for the gather kernels instead of running them on a blocking that was never tuned for them
            sharded = shard is not None and shard.world > 1
            rb, nr = (shard.row_begin, shard.n_local) if sharded else (0, num_nodes)
            if csr.sliced(rb, nr, F) is None:
                aligned = choose_shard_blocks(num_nodes, row_bytes, edge_index.shape[1], shard)
                nb2, br2 = aligned if aligned else (choose_source_blocks(num_nodes, row_bytes, edge_index.shape[1]), 0)
                if (nb2, br2 or -(-int(num_nodes) // nb2)) != (csr.n_blocks, csr.block_rows):
                    global _SLICED_FALLBACK_WARNED
                    if not _SLICED_FALLBACK_WARNED:
                        _SLICED_FALLBACK_WARNED = True
                        import warnings
                        warnings.warn("difformer_amd: this graph does not fit the feature-sliced product (see "
                                      "GraphCSR._build_sliced); using the gather kernels on their own blocking")
                    csr = GraphCSR.build(edge_index, edge_weight, num_nodes, nb2, block_rows=br2)
                    csr._format_checked = True
                    csr.weight_scale = scale
        self.entries[key] = (weakref.ref(edge_index), weakref.ref(given_weight) if given_weight is not None else None,
                             csr)
        while len(self.entries) > max(self.capacity, getattr(self, "_floor", 0)):
            self.entries.popitem(last=False)
        return csr

    UNIFORM_ALWAYS = False       # tests: run the check on graphs of any size
    UNIFORM_MIN_EDGES = 4096     # below this the CSR build is a handful of launches and the check would double its host cost
    # ... and it only buys something where the unweighted graph takes kernels the weighted one cannot: the feature-sliced
    # product (SLICED_MIN_ROWS nodes, SLICED_MIN_DEGREE entries per row).  A 1,068-node snapshot with fresh tensors per forward
    # (spatial-temporal/main.py:96-105) would pay two reductions and a host read per snapshot for the same gather kernels.

    def _uniform_weight(self, edge_weight, n_edges, num_nodes=None):
        """The constant all entries of `edge_weight` are equal to (0.0 when that constant is not finite: nan_to_num,
        difformer.py:74), or None: weights that vary, that want a gradient, or a graph too small for the check to pay.
        Remembered per weight tensor (identity + version)."""
        if n_edges < self.UNIFORM_MIN_EDGES or edge_weight.numel() != n_edges or not edge_weight.is_floating_point():
            return None
        if num_nodes is not None and (num_nodes < SLICED_MIN_ROWS or n_edges < SLICED_MIN_DEGREE * num_nodes) and not self.UNIFORM_ALWAYS:
            return None
        if edge_weight.requires_grad and torch.is_grad_enabled():
            return None
        memo = self.__dict__.setdefault("_uniform", OrderedDict())
        key = (id(edge_weight), edge_weight.data_ptr(), tensor_version(edge_weight), n_edges)
        hit = memo.get(key)
        if hit is not None and hit[0]() is edge_weight and key[2] >= 0:
            return hit[1]
        for k in [k for k, v in memo.items() if v[0]() is None]:
            del memo[k]
        w = edge_weight.detach()
        lo, hi = torch.stack([w.min(), w.max()]).tolist()            # NaN anywhere -> both NaN -> lo == hi is False
        if lo == hi:
            u = float(lo) if math.isfinite(lo) else 0.0
        elif math.isnan(lo) and bool(torch.isnan(w).all()):
            u = 0.0
        else:
            u = None
        memo[key] = (weakref.ref(edge_weight), u)
        while len(memo) > 64:
            memo.popitem(last=False)
        return u

    @staticmethod
    def blocking(edge_index, edge_weight, num_nodes, row_bytes=256, shard=None, elem_size=4):
        """(n_blocks, block_rows) of the CSR `get` builds for this graph: the tiling of the feature-sliced product for
        dense unweighted fp32 graphs, else ~2.5 MiB source blocks (aligned with the rank boundaries of a row shard)."""
        tiling = sliced_tiling(num_nodes, row_bytes // elem_size, edge_index.shape[1], edge_weight, shard, elem_size)
        if tiling is not None:
            return tiling if tiling[0] > 1 else (1, 0)
        aligned = choose_shard_blocks(num_nodes, row_bytes, edge_index.shape[1], shard)
        return aligned if aligned else (choose_source_blocks(num_nodes, row_bytes, edge_index.shape[1]), 0)

    def put(self, edge_index, edge_weight, num_nodes, csr):
        """Register a CSR built elsewhere (graph_utils.subgraph_batches: all batches of an epoch from one sort) under the
        key `get` would use for these tensors, so that `model(x_i, edge_index_i)` finds it without building anything."""
        key = self._key(edge_index, edge_weight, num_nodes, csr.n_blocks, 0)
        self.entries[key] = (weakref.ref(edge_index), weakref.ref(edge_weight) if edge_weight is not None else None, csr)
        self.entries.move_to_end(key)
        while len(self.entries) > max(self.capacity, getattr(self, "_floor", 0)):
            self.entries.popitem(last=False)

    def drop(self, edge_index):
        """Forget every CSR built from this edge_index tensor (frees its HBM once no autograd node holds it)."""
        for k in [k for k, (ei_ref, _, _) in self.entries.items() if ei_ref() is edge_index]:
            del self.entries[k]

    def reserve(self, n):
        """Keep room for at least n entries (an epoch's worth of registered batches must not evict one another)."""
        self._floor = max(int(n), 0)

    def _purge(self):
        """Drop entries whose edge tensors have been freed: a mini-batch loop (main-batch.py:126-131) makes a new
        edge_index per batch, and a dead entry would otherwise pin its CSR (and sliced format) in HBM until eight
        newer graphs push it out."""
        dead = [k for k, (ei_ref, ew_ref, _) in self.entries.items()
                if ei_ref() is None or (ew_ref is not None and ew_ref() is None)]
        for k in dead:
            del self.entries[k]

    def clear(self):
        self.entries.clear()


_SLICED_FALLBACK_WARNED = False
csr_cache = _CSRCache()


# ------------------------------------------------------------------------------------------
# graphs with community structure: the model runs in a MIXED node order
# ------------------------------------------------------------------------------------------
MIX_THRESHOLD = 2.5     # mean over rows of (largest per-tile share of the row's entries) x tiles: 1.3 on a uniform graph,
                        # ~8 when every row's entries sit in one or two source tiles


class MixedGraph:
    """A graph relabelled by a fixed random permutation of its nodes.  The sliced product sweeps the SOURCE rows tile by
    tile with all rows of a slot in lock step; when a row's entries sit in one or two tiles (the real ogbn-proteins: 8
    species that interact almost only among themselves, node ids grouped by species) most rounds of a wave are empty in
    any given tile and the rest are padded to them: 3.4-4.6 slots per entry instead of 1.56, 2-3x the time
    (profiles/r02_experiments.md).  Every per-node operation of the model is equivariant under a relabelling, so the model
    permutes x once on the way in and the logits once on the way out and runs everything else on `edge_index` = the
    relabelled graph: perm[i] = original node at position i, inv[perm] = arange."""

    def __init__(self, edge_index, num_nodes):
        dev = edge_index.device
        g = torch.Generator(device=dev).manual_seed(0x5EED)
        self.perm = torch.randperm(num_nodes, generator=g, device=dev)
        self.inv = torch.empty_like(self.perm)
        self.inv[self.perm] = torch.arange(num_nodes, device=dev)
        self.edge_index = self.inv[edge_index]              # same edges, same order, nodes renamed


class _MixCache:
    """MixedGraph (or the decision not to mix) per edge_index tensor, keyed like the CSR cache."""

    def __init__(self, capacity=8):
        self.capacity, self.entries = capacity, OrderedDict()

    def get(self, edge_index, num_nodes, F):
        key = (id(edge_index), edge_index.data_ptr(), tuple(edge_index.shape), tensor_version(edge_index), int(num_nodes), int(F))
        hit = self.entries.get(key)
        if hit is not None and hit[0]() is edge_index:
            self.entries.move_to_end(key)
            return hit[1]
        for k in [k for k, v in self.entries.items() if v[0]() is None]:
            del self.entries[k]
        mix = None
        if sliced_tiling(num_nodes, F, edge_index.shape[1], None, None, 4) is not None:
            csr = csr_cache.get(edge_index, None, num_nodes, F * 4, build_format=False)     # probe: pointers only
            if csr.n_blocks > 1 and csr.blkptr is not None:
                cnt = csr.blkptr.view(csr.n_blocks + 1, num_nodes)
                cnt = (cnt[1:] - cnt[:-1]).to(torch.float32)                    # entries per (tile, row)
                share = float((cnt.max(dim=0).values.sum() * csr.n_blocks / max(csr.nnz, 1)).item())     # one sync, cold path
                if share > MIX_THRESHOLD:
                    mix = MixedGraph(edge_index, num_nodes)
                    csr_cache.drop(edge_index)      # the forward runs on the relabelled graph: the probe CSR is dead weight
        self.entries[key] = (weakref.ref(edge_index), mix)
        while len(self.entries) > self.capacity:
            self.entries.popitem(last=False)
        return mix

    def clear(self):
        self.entries.clear()


mix_cache = _MixCache()


def gcn_aggregate(csr: GraphCSR, x, attn=None, attn_scale=1.0, gcn_scale=1.0, shard: Optional[RowShard] = None,
                  tail=None):
    """x [n,H,D] (this rank's rows of the value tensor) -> gcn_scale * A_hat x (+ attn_scale * attn).

    difformer.py:75-78 plus the combine of :130-134.  Sharded: all-gather x, SpMM over the local
    destination rows.  `tail` (H == 1 only) = dict(x0, prev, alpha, ln_weight, ln_bias, eps) fuses the
    layer tail of :139-140 / :200-203 into the SpMM epilogue; the result is then [n, 1, D]."""
    n, H, D = x.shape
    be = get_backend()
    args = (csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, csr.num_nodes, csr.nnz)
    flat = lambda a: None if a is None else (a.materialize() if isinstance(a, LazyAttention) else a).reshape(n, H * D)
    if shard is None or shard.world <= 1:
        a2 = flat(attn)
        x2 = x.reshape(n, H * D)
        sl = csr.sliced(0, n, H * D) if (x.dtype == torch.float32 and n == csr.num_nodes) else None
        if sl is not None:
            # dense unweighted graph: sources pre-scaled and staged slice by slice in LDS (csrc/gcn_sliced.hip); the
            # LayerNorm of the tail needs whole rows, which the slice workgroups do not have: it runs as its own pass
            ys = be.sliced_prescale(x2, csr.rowptr, csr.num_nodes, sl.plan, csr.dinv)
            out = be.sliced_spmm(sl, ys, csr.rowptr, csr.num_nodes, 0, n, H * D, a2, attn_scale, gcn_scale, csr.dinv)
            if tail is not None:
                out = be.layer_tail(out.reshape(n, H, D), tail.get("x0"), tail.get("prev"), tail.get("alpha", 0.5),
                                    tail.get("ln_weight"), tail.get("ln_bias"), tail.get("eps", 1e-5),
                                    tail.get("relu", False))
                return out.reshape(n, 1, D)
            return out.reshape(n, H, D)
        out = be.spmm(*args, x2, 0, n, a2, attn_scale, gcn_scale, tail, csr.row_order(0, n))
        return out.reshape(n, H, D)
    # row-sharded: the one exchange step of this operator is the all-gather of the value rows (N*H*D elements)
    if isinstance(x, GatheredRows):            # already started by project_simple_attention
        local, handle = x.local.reshape(n, H * D), x.handle
    else:
        local = x.reshape(n, H * D)
        handle = shard.all_gather_rows_async(local)
    row_begin, n_rows = shard.row_begin, shard.n_local
    F = H * D
    sl = csr.sliced(row_begin, n_rows, F) if local.dtype == torch.float32 else None
    if sl is not None:
        # dense unweighted graph: the shard's feature-sliced product (source tiles split over the workgroups of a panel)
        # over the gathered rows, as in the single-GPU branch above
        a2 = flat(attn)                        # waits for the record all-reduce, runs `apply` under the all-gather
        x2 = handle.wait()
        ys = be.sliced_prescale(x2, csr.rowptr, csr.num_nodes, sl.plan, csr.dinv)
        out = be.sliced_spmm(sl, ys, csr.rowptr, csr.num_nodes, row_begin, n_rows, F, a2, attn_scale, gcn_scale, csr.dinv)
        if tail is not None:
            out = be.layer_tail(out.reshape(n_rows, H, D), tail.get("x0"), tail.get("prev"), tail.get("alpha", 0.5),
                                tail.get("ln_weight"), tail.get("ln_bias"), tail.get("eps", 1e-5), tail.get("relu", False))
            return out.reshape(n_rows, 1, D)
        return out.reshape(n_rows, H, D)
    order = csr.row_order(row_begin, n_rows)
    rows = csr.block_rows
    split = (csr.n_blocks > 1 and csr.nnz > 0 and F % 4 == 0 and F <= 256 and row_begin % rows == 0 and
             (row_begin + n_rows == csr.num_nodes or (row_begin + n_rows) % rows == 0))
    if split:
        # blocks [own_lo, own_hi) hold exactly this rank's own value rows: sweep them now, under the collective, and
        # park the accumulators; the rest of the sweep and the epilogue follow once the other ranks' rows have landed
        own_lo, own_hi = row_begin // rows, -(-(row_begin + n_rows) // rows)
        local = local if local.is_contiguous() else local.contiguous()
        scratch = be.spmm(*args, local, row_begin, n_rows, None, attn_scale, gcn_scale, None, order,
                          (0, own_lo, own_hi, None, row_begin))
        a2 = flat(attn)                        # waits for the record all-reduce, runs `apply` -- after part 0 is queued
        x2 = handle.wait()
        out = be.spmm(*args, x2, row_begin, n_rows, a2, attn_scale, gcn_scale, tail, order,
                      (1, own_lo, own_hi, scratch, 0))
    else:
        a2 = flat(attn)
        x2 = handle.wait()
        out = be.spmm(*args, x2, row_begin, n_rows, a2, attn_scale, gcn_scale, tail, order)
    return out.reshape(n_rows, H, D)


class NarrowFactors:
    """Weight-only factors of the background coefficient chain (csrc/side_chain.hip), float32 [80 x 80] zero padded,
    augmented index 64: pt = W~q^T W~k, vtt = [W~v^T | e]^T, st = [W~q^T W~q ; W~k^T W~k] with W~ = [W | b].  Computed in
    float64 once per parameter version (the module caches the object)."""

    def __init__(self, Wq, bq, Wk, bk, Wv, bv):
        f64, B, A = torch.float64, 80, 64
        D, C = Wq.shape
        dev = Wq.device

        def aug(W, b):
            M = torch.zeros((W.shape[0], B), dtype=f64, device=dev)
            M[:, : W.shape[1]] = W.to(f64)
            M[:, A] = b.to(f64)
            return M
        Wq_, Wk_ = aug(Wq, bq), aug(Wk, bk)
        if Wv is not None:
            Wv_ = aug(Wv, bv)
        else:                                                           # use_weight = False: v = x (difformer.py:120)
            Wv_ = torch.zeros((C, B), dtype=f64, device=dev)
            Wv_[torch.arange(C), torch.arange(C)] = 1.0
        self.pt = (Wq_.t() @ Wk_).to(torch.float32).contiguous().reshape(-1)
        self.st = torch.stack([(Wq_.t() @ Wq_).reshape(-1), (Wk_.t() @ Wk_).reshape(-1)]).to(torch.float32).contiguous().reshape(-1)
        vtt = torch.zeros((B, B), dtype=f64, device=dev)
        vtt[: Wv_.shape[0]] = Wv_
        vtt[A, A] = 1.0
        self.vtt = vtt.to(torch.float32).contiguous().reshape(-1)


_SIDE_STREAMS = {}


def side_stream(dev):
    """The second stream the background coefficient chain runs on (one per device)."""
    key = (dev.type, dev.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(dev)
    return _SIDE_STREAMS[key]


SIDE_CHAIN = os.environ.get("DIFFORMER_SIDE_CHAIN", "1") != "0"
# DIFFORMER_EXACT_FP32=1: no product of the forward runs on split-bfloat16 operands -- the long-row input Linear takes the
# fp32 MFMA (the library reads the same variable), and the output Linear is not folded into the last layer kernel (whose
# extra product is a split-bfloat16 one) but runs as its own fp32-MFMA launch.  Results move by ~4e-6 of the logits' scale.
EXACT_FP32 = os.environ.get("DIFFORMER_EXACT_FP32", "0") == "1"

_F32_PARAMS = OrderedDict()


def f32_param(t):
    """Exact float32 copy of a bfloat16 parameter, cached until the parameter changes (identity + version keyed, like
    the CSR cache); float32 tensors pass through.  The closed-form kernels take their coefficients in float32."""
    if t is None or t.dtype == torch.float32:
        return t
    ver = tensor_version(t)
    if ver < 0:
        return t.detach().to(torch.float32).contiguous()       # version unreadable (inference tensor): never cached
    key = (id(t), t.data_ptr(), ver, tuple(t.shape), str(t.device))
    hit = _F32_PARAMS.get(key)
    if hit is not None and hit[0]() is t:
        _F32_PARAMS.move_to_end(key)
        return hit[1]
    c = t.detach().to(torch.float32).contiguous()
    _F32_PARAMS[key] = (weakref.ref(t), c)
    while len(_F32_PARAMS) > 256:
        _F32_PARAMS.popitem(last=False)
    return c


def invalidate_param_caches():
    """Forget the cached float32 copies of bfloat16 parameters and the packed weights of the long-row Linear (see
    param_key for when this is needed)."""
    _F32_PARAMS.clear()
    if _BACKEND is not None and hasattr(_BACKEND, "_packed"):
        _BACKEND._packed.clear()


def slice_sharded(shard, C, dtype=torch.float32):
    """True when a closed-form layer of width C on this shard splits its aggregation by feature slices (dist.py)."""
    return (shard is not None and shard.world > 1 and getattr(shard, "product", "row") == "slice" and
            dtype == torch.float32 and shard.slice_width(C) > 0)


def _closed_form_slice_sharded(be, x, Wq, bq, Wk, bk, Wv, bv, csr, attn_scale, gcn_scale, x0, residual, alpha, ln_weight,
                               ln_bias, eps, relu, shard, head):
    """The closed-form layer on a row shard with the aggregation split by FEATURE SLICES: this rank multiplies all
    n_global destination rows of its C / world columns (the sliced product of the whole graph at that width -- or the
    gather kernel where the graph does not take it), between two all-to-alls; Gram record, coefficients and the layer
    kernel stay on the rank's own rows exactly as in the row-sharded layer.  `csr` is the FULL graph's (built without a
    shard: csr_cache.get(..., shard=None) at the slice width)."""
    n, C = x.shape
    D = Wq.shape[0]
    N, w = shard.n_global, shard.slice_width(C)
    xs = shard.all_to_all_columns(x)                                        # [N, w]: my columns of every row
    record, _ = be.gram(x, None, None)                                      # local rows; runs while the exchange lands
    shard.all_reduce_sum(record)
    coef = be.simple_coeffs(record, N, C, D, Wq, bq, Wk, bk, Wv, bv, attn_scale)
    sl = csr.sliced(0, N, w) if hasattr(csr, "sliced") else None
    if sl is not None:
        ys = be.sliced_prescale(xs, csr.rowptr, N, sl.plan)
        axs = be.sliced_spmm(sl, ys, csr.rowptr, N, 0, N, w, None, 1.0, gcn_scale)
    else:
        axs = be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, N, csr.nnz, xs, 0, N, None, 1.0, gcn_scale,
                      None, csr.row_order(0, N))
    ax = shard.all_to_all_rows(axs)                                         # [n, C]: every column block of my rows
    rs = None
    if Wv is not None:
        rs = csr.row_sums()[shard.row_begin: shard.row_begin + n]
    return be.simple_layer(x, coef, D, ax, Wv, bv, rs, gcn_scale, x0, residual, alpha, ln_weight, ln_bias, eps, relu,
                           **({"head": head} if head is not None else {}))


def simple_layer_closed_form(x, Wq, bq, Wk, bk, Wv, bv, csr, attn_scale, gcn_scale, x0, residual, alpha, ln_weight,
                             ln_bias, eps, relu=False, carry=None, shard: Optional[RowShard] = None, factors=None, head=None,
                             keep=None):
    """One DIFFormer layer with the `simple` kernel, query == source == x [n, C] (this rank's rows), one head
    (csrc/simple_layer.hip): Gram record -> coefficients -> SpMM on x -> the layer kernel.  q, k, v and the attention
    output never reach memory.  csr = None: use_graph = False.  Wv = None: use_weight = False.
    Row-sharded: the two exchange steps of SURVEY 8e keep their place -- ONE all-reduce of the 4,160-float Gram record
    (instead of the KtV record) and ONE all-gather of the source rows (x instead of v), started first so that the
    Gram pass and the coefficients run under it.
    `carry` (dict, optional; single GPU) chains layers over the same graph: with carry["want_next"] the layer kernel also
    writes the slice-major pre-scaled copy of its output (the next layer's SpMM operand) from its registers, and with
    carry["next_record"] the Gram record of the output too; the next layer picks up what it finds.
    `keep` (dict, optional; the training forward, autograd_ops._ClosedFormLayer): receives the record, the coefficients, the
    aggregated rows and the row sums the backward pass starts from (the aggregation then stays a launch of its own)."""
    be = get_backend()
    n, C = x.shape
    D = Wq.shape[0]
    # bfloat16 activations (BASELINE config C5): the parameters go in as exact float32 copies, all arithmetic is float32
    Wq, bq, Wk, bk, Wv, bv, ln_weight, ln_bias = (f32_param(t) for t in (Wq, bq, Wk, bk, Wv, bv, ln_weight, ln_bias))
    sharded = shard is not None and shard.world > 1
    n_global, row_begin = (shard.n_global, shard.row_begin) if sharded else (n, 0)
    if sharded and csr is not None and slice_sharded(shard, C, x.dtype):
        return _closed_form_slice_sharded(be, x, Wq, bq, Wk, bk, Wv, bv, csr, attn_scale, gcn_scale, x0, residual, alpha,
                                          ln_weight, ln_bias, eps, relu, shard, head)
    sl = None
    if csr is not None and (sharded or n == csr.num_nodes) and x.dtype == torch.float32:
        sl = csr.sliced(row_begin, n, C)
    handle = shard.all_gather_rows_async(x) if (sharded and csr is not None) else None
    have = carry.get("products") if (carry is not None and not sharded) else None
    if have is not None and not (have["x"] is x and have["sl"] is sl):
        have = None
    record = have["record"] if have is not None else None
    ys = have["ys"] if have is not None else None
    # With the sliced product ahead (0.3 ms that does not need the coefficients) the chain Gram -> coefficients runs as
    # single-wave background kernels on a second stream, UNDER the product (csrc/side_chain.hip).
    background = (SIDE_CHAIN and factors is not None and sl is not None and not sharded and x.dtype == torch.float32 and
                  C % 4 == 0 and D % 4 == 0 and x.is_cuda and hasattr(be, "coeffs_bg"))
    coef = join = None
    if background:
        if record is None and ys is None:          # first layer: the Gram pass also writes the slice-major copy the product reads
            record, ys = be.gram(x, csr.rowptr, sl.plan)
        main, side = torch.cuda.current_stream(x.device), side_stream(x.device)
        side.wait_stream(main)                     # the chain may start once x (and the record) exist ...
        join = (main, side)                        # ... but is ENQUEUED after the product (below): its workgroups then take the
                                                   # wave slot the product leaves free on every CU instead of delaying its start
    else:
        need_ys = sl is not None and ys is None and not sharded
        if (record is None and not need_ys and not sharded and x.dtype == torch.float32 and C % 4 == 0 and
                n <= GRAM_COEFFS_MAX_ROWS and hasattr(be, "gram_coeffs")):
            # small graphs (node classification/run.sh: Cora ... PubMed): the few partial records of the Gram pass are summed
            # inside the coefficient kernel -- one launch (and one host call) less in a chain of four dependent ones
            record, coef = be.gram_coeffs(x, n_global, C, D, Wq, bq, Wk, bk, Wv, bv, attn_scale)
        else:
            if record is None:
                record, ys2 = be.gram(x, csr.rowptr if need_ys else None, sl.plan if need_ys else None)
                ys = ys2 if need_ys else ys
            if sharded:
                shard.all_reduce_sum(record)
            coef = be.simple_coeffs(record, n_global, C, D, Wq, bq, Wk, bk, Wv, bv, attn_scale)
    ax = rs = gather = None
    want_next = (carry is not None and not sharded and carry.get("want_next", False) and D % 4 == 0 and D == C and
                 x.dtype == torch.float32)
    want_rec = want_next and carry.get("next_record", False)
    if (csr is not None and sl is None and not sharded and not want_rec and keep is None and LAYER_GATHER and csr.n_blocks == 1 and n == csr.num_nodes and
            0 < csr.nnz <= LAYER_GATHER_MAX_DEGREE * n and hasattr(be, "_simple_layer_gather") and
            csr.max_degree() <= LAYER_GATHER_MAX_ROW):
        # a few entries per row: the layer kernel walks the CSR itself, no separate SpMM launch and no `ax` round trip
        gather = (csr.rowptr, csr.src, csr.val)
    elif csr is not None:
        x_src = handle.wait() if sharded else x
        if sl is not None:
            if ys is None:
                ys = be.sliced_prescale(x_src, csr.rowptr, csr.num_nodes, sl.plan)
            ax = be.sliced_spmm(sl, ys, csr.rowptr, csr.num_nodes, row_begin, n, C, None, 1.0, gcn_scale)
            if join is not None:
                with torch.cuda.stream(join[1]):
                    coef = be.coeffs_bg(None if record is not None else x, record, n_global, factors, C, D, attn_scale)
        else:
            ax = be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, csr.num_nodes, csr.nnz, x_src, row_begin, n,
                         None, 1.0, gcn_scale, None, csr.row_order(row_begin, n))
        if Wv is not None:
            rs = csr.row_sums()
            if sharded:
                rs = rs[row_begin: row_begin + n]
    if join is not None:                           # the layer kernel needs the coefficients: the side stream joins
        join[0].wait_stream(join[1])
        coef.record_stream(join[0])
    if carry is not None:
        carry["products"] = None
    if keep is not None:
        keep.update(record=record, coef=coef, ax=ax, row_sums=rs)
    if head is not None:                           # last layer: the model's output Linear rides in the same pass -> logits
        head = tuple(f32_param(t) for t in head)   # bfloat16 storage: exact float32 copies, as for the other parameters
        return be.simple_layer(x, coef, D, ax, Wv, bv, rs, gcn_scale, x0, residual, alpha, ln_weight, ln_bias, eps, relu,
                               head=head, gather=gather)
    if gather is not None:
        return be.simple_layer(x, coef, D, None, Wv, bv, None, gcn_scale, x0, residual, alpha, ln_weight, ln_bias, eps, relu,
                               gather=gather)
    if not (want_next and (sl is not None or want_rec)):
        return be.simple_layer(x, coef, D, ax, Wv, bv, rs, gcn_scale, x0, residual, alpha, ln_weight, ln_bias, eps, relu)
    out, ys2, record2 = be.simple_layer(x, coef, D, ax, Wv, bv, rs, gcn_scale, x0, residual, alpha, ln_weight, ln_bias, eps,
                                        relu, csr.rowptr if sl is not None else None, sl.plan if sl is not None else None,
                                        want_rec)
    carry["products"] = dict(x=out, sl=sl, record=record2, ys=ys2)
    return out


# The closed-form layer aggregates inside its own kernel (dif_simple_layer_gather_*) up to this many entries per row on
# average; above, the separate SpMM kernels (rows split over lanes, long rows over quads) balance the work better.
LAYER_GATHER = os.environ.get("DIFFORMER_LAYER_GATHER", "1") != "0"
LAYER_GATHER_MAX_DEGREE = int(os.environ.get("DIFFORMER_LAYER_GATHER_MAX_DEGREE", "12"))
# ... and only without long rows: the 16 rows of a tile walk in lock step to the longest (GraphCSR.max_degree); from ~64
# entries on the SpMM kernel, which hands long rows to a whole block, is faster (profiles/r03_experiments.md, 5b)
LAYER_GATHER_MAX_ROW = int(os.environ.get("DIFFORMER_LAYER_GATHER_MAX_ROW", "64"))

def closed_form_coeffs_backward(record, n_global, C, D, Wq, bq, Wk, bk, Wv, bv, attn_scale, d_MnT, d_cn, d_u, d_cd):
    """Backward of the coefficient stage (backend.simple_coeffs): gradients of a loss with respect to the Gram record and
    the projection parameters, given its gradients with respect to MnT [D, C], cn [D], u [C] and cd.  float64 tensor ops on
    (C + 1)-square matrices.  With the augmented matrices W~ = [W | b] and G~ = [[G, sx], [sx^T, N]] (difformer.py:18-38):
        KtV = W~k G~ W~v^T,  sum k = W~k G~ e,  sum v = W~v G~ e,  |Q|^2 = <W~q G~, W~q>,  |K|^2 = <W~k G~, W~k>,  s = (|Q|^2 |K|^2)^-1/2
        [Mn; cn - a sum v] = a s W~q^T KtV,    [u; cd - N] = s W~q^T sum k
    -> (S [C, C], t [C]) with d x = x S + 1 t^T (S, t = the symmetrised gradient of G~), d_Wq, d_bq, d_Wk, d_bk, d_Wv, d_bv
    (float32; the last two None when Wv is None: use_weight = False)."""
    f64 = torch.float64
    dev = record.device
    N = float(n_global)
    r = record.to(f64)
    Gt = torch.empty((C + 1, C + 1), dtype=f64, device=dev)
    Gt[:C, :C] = r[: C * C].view(C, C)
    Gt[:C, C] = r[C * C: C * C + C]
    Gt[C, :C] = r[C * C: C * C + C]
    Gt[C, C] = N
    aug = lambda W, b: torch.cat([W.to(f64), b.to(f64)[:, None]], dim=1)
    Wq_, Wk_ = aug(Wq, bq), aug(Wk, bk)
    if Wv is not None:
        Wv_ = aug(Wv, bv)
    else:
        Wv_ = torch.cat([torch.eye(D, C, dtype=f64, device=dev), torch.zeros(D, 1, dtype=f64, device=dev)], dim=1)
    a = float(attn_scale)
    Aq, Ak, Av = Wq_ @ Gt, Wk_ @ Gt, Wv_ @ Gt
    ktv, ksum = Ak @ Wv_.t(), Ak[:, C]
    q2, k2 = (Aq * Wq_).sum(), (Ak * Wk_).sum()
    s = (q2 * k2) ** -0.5
    P, rr = Wq_.t() @ ktv, Wq_.t() @ ksum
    Dm = torch.cat([d_MnT.to(f64).t(), d_cn.to(f64)[None, :]], dim=0)                   # [C + 1, D]
    Du = torch.cat([d_u.to(f64).reshape(-1), d_cd.to(f64).reshape(1)])                  # [C + 1]
    dP, dr = a * s * Dm, s * Du
    ds = a * (Dm * P).sum() + (Du * rr).sum()
    dvsum = a * d_cn.to(f64)
    dktv, dksum = Wq_ @ dP, Wq_ @ dr
    dq2, dk2 = -0.5 * ds * s / q2, -0.5 * ds * s / k2
    e_col = Gt[:, C]
    dWq_ = ktv @ dP.t() + torch.outer(ksum, dr) + 2.0 * dq2 * Aq
    dWk_ = 2.0 * dk2 * Ak + dktv @ Av + torch.outer(dksum, e_col)
    dWv_ = dktv.t() @ Ak + torch.outer(dvsum, e_col)
    dGt = dq2 * (Wq_.t() @ Wq_) + dk2 * (Wk_.t() @ Wk_) + Wk_.t() @ dktv @ Wv_
    dGt[:, C] += Wk_.t() @ dksum + Wv_.t() @ dvsum
    S = dGt + dGt.t()
    f32 = torch.float32
    out = [S[:C, :C].to(f32).contiguous(), S[C, :C].to(f32).contiguous(), dWq_[:, :C].to(f32), dWq_[:, C].to(f32),
           dWk_[:, :C].to(f32), dWk_[:, C].to(f32)]
    out += [dWv_[:, :C].to(f32), dWv_[:, C].to(f32)] if Wv is not None else [None, None]
    return out


XWIDE_MAX = int(os.environ.get("DIFFORMER_XWIDE_MAX", "416"))   # widest layer of the one-pass kernel with streamed weights (0: library GEMMs)
CLOSED_FORM_WIDE_MAX = 512      # widest layer the Gram-record formulation is used for (record = C x C floats)
# Layers wider than this take the Gram-record formulation at the scripts' widths: from 65 columns with the one-pass layer kernel
# of csrc/simple_layer_wide.hip (up to 128 columns: pokec-batch-h128 0.95 -> see profiles/r04_experiments.md); under
# DIFFORMER_EXACT_FP32=1 (that kernel runs split-bfloat16 products) from 129, where the library GEMMs around the tail pass win
# over the q / k / v operator path (at 128 they cost the same: 1.01 vs 1.02 ms)
CLOSED_FORM_WIDE_MIN = 128 if EXACT_FP32 else 64


def set_exact_fp32(flag):
    """Switch DIFFORMER_EXACT_FP32 at run time (bench.py's second pass) -> the previous setting.  Callers drop what they
    cached under the old setting (`DIFFormer.invalidate_caches()`: a captured forward bakes the kernel choice in)."""
    global EXACT_FP32, CLOSED_FORM_WIDE_MIN
    was = EXACT_FP32
    EXACT_FP32 = bool(flag)
    be = _BACKEND
    if be is not None and hasattr(be, "set_exact_fp32"):
        be.set_exact_fp32(EXACT_FP32)                       # the launchers' own choices (dif_set_exact_fp32)
    CLOSED_FORM_WIDE_MIN = 128 if EXACT_FP32 else 64
    return was


class WideCoefficients:
    """Weight-only factors of the closed form at the scripts' widths, float64, rebuilt when a parameter changes.
    With the augmented matrices  X~ = [X | 1],  W~ = [W | b]  (so q = X~ W~q^T etc.) and  G~ = X~^T X~ = [[G, sx], [sx^T, N]]:
        |Q|^2 = <W~q^T W~q, G~>,   |K|^2 = <W~k^T W~k, G~>                       (difformer.py:20-21)
        T = G~ V~,    V~ = [W~v^T | e_last | 0 0 0]     -> last row of T = [sum v | N | 0 0 0]
        R = P~ T,     P~ = W~q^T W~k                     -> rows 0..C-1 = [Mn | u] / s,  last row = [bq KtV | bq.ksum] / s
    num = x Mn + cn, den = x.u + cd  with  [cn | cd] = s R[C] + T[C]              (:25-38)"""

    def __init__(self, Wq, bq, Wk, bk, Wv, bv):
        f64 = torch.float64
        aug = lambda W, b: torch.cat([W.to(f64), b.to(f64)[:, None]], dim=1)
        Wq_, Wk_ = aug(Wq, bq), aug(Wk, bk)
        D, C1 = Wq_.shape
        if Wv is not None:
            Wv_ = aug(Wv, bv)
        else:                                                           # use_weight = False: v = x (difformer.py:120)
            Wv_ = torch.cat([torch.eye(C1 - 1, dtype=f64, device=Wq.device), torch.zeros(C1 - 1, 1, dtype=f64, device=Wq.device)], 1)
        self.D = Wv_.shape[0]
        self.S = torch.stack([(Wq_.t() @ Wq_).reshape(-1), (Wk_.t() @ Wk_).reshape(-1)])           # [2, (C+1)^2]
        self.P = (Wq_.t() @ Wk_).contiguous()                                                      # [(C+1), (C+1)]
        V = torch.zeros((C1, self.D + 4), dtype=f64, device=Wq.device)
        V[:, : self.D] = Wv_.t()
        V[C1 - 1, self.D] = 1.0
        self.V = V


WIDE_COEFFS_OWN = os.environ.get("DIFFORMER_WIDE_COEFFS_LIBRARY", "0") != "1"      # 1: the two float64 library GEMMs of round 4


def simple_layer_closed_form_wide(x, coeffs: WideCoefficients, Wv, bv, csr, attn_scale, gcn_scale, x0, residual, alpha,
                                  ln_weight, ln_bias, eps):
    """The closed form of simple_layer_closed_form at the widths the reference's scripts train with (hidden 128 / 300 /
    400: run.sh): query == source == x [n, C] fp32, one head, single GPU, inference.
        G~ = [X | 1]^T [X | 1]             one streaming pass (dif_gram_sym_f32: the upper 64-blocks of X^T X on the fp32 MFMA)
        [Mn | u], [cn | cd]                two float64 library GEMMs on (C + 1)-square matrices (WideCoefficients)
        Z = x [Mn | u] + [cn | cd]         ONE row GEMM: numerator | denominator
        gcn = (A_hat x) Wv^T               SpMM on x, one row GEMM  (+ (A_hat 1) bv^T inside the tail)
        tail                               num / den, combine, + x0, residual, LayerNorm in one pass (dif_layer_tail_mix_f32)
    instead of three projections (3 N C D), the K^T V reduce (N D^2) and the Q-side apply (N D^2): 2-3x fewer FLOPs and
    q, k, v never written."""
    be = get_backend()
    n, C = x.shape
    D = coeffs.D
    x3 = x.reshape(n, 1, C)
    rec = be.gram_sym(x)                                                # [X^T X (upper blocks) | sum x | ...]
    if hasattr(be, "wide_coeffs") and WIDE_COEFFS_OWN:
        B, bias = be.wide_coeffs(rec, C, n, coeffs.S, coeffs.V, coeffs.P)   # both float64 products + bookkeeping: two launches
    else:
        Gt, partial = be.wide_gram(rec, C, n, coeffs.S)                 # G~ (float64) and the partial sums of |Q|^2, |K|^2
        T = Gt @ coeffs.V                                               # [(C+1), D+4]
        R = coeffs.P @ T
        B, bias = be.wide_scale(R, T, partial, C)                       # [Mn | u | 0 0 0], [cn | cd | 0 0 0] (float32)
    if max(C, D) <= 128 and not EXACT_FP32 and hasattr(be, "simple_layer_wide"):
        # hidden 128 (node classification/run.sh:42-44): both row products, the division, the combine, the residual and the
        # LayerNorm in ONE pass over the rows (csrc/simple_layer_wide.hip) -- no library GEMM, no [n, D + 4] intermediate
        ax = rs = None
        if csr is not None:
            ax = gcn_aggregate(csr, x3, None, 1.0, 1.0).reshape(n, C)
            rs = csr.row_sums() if Wv is not None else None
        return be.simple_layer_wide(x, B, bias, D, attn_scale, ax, Wv if csr is not None else None,
                                    bv if csr is not None else None, rs, gcn_scale, x0, residual, alpha, ln_weight, ln_bias, eps)
    if max(C, D) <= XWIDE_MAX and not EXACT_FP32 and hasattr(be, "simple_layer_xwide"):
        # hidden 300 / 400 (image and text/run.sh:27): the same one pass with the rows in registers and the weights streamed
        # through LDS (csrc/simple_layer_xwide.hip) instead of a library GEMM per product around a tail pass
        ax = rs = None
        if csr is not None:
            ax = gcn_aggregate(csr, x3, None, 1.0, 1.0).reshape(n, C)
            rs = csr.row_sums() if Wv is not None else None
        return be.simple_layer_xwide(x, B, bias, D, attn_scale, ax, Wv if csr is not None else None,
                                     bv if csr is not None else None, rs, gcn_scale, x0, residual, alpha, ln_weight, ln_bias, eps)
    Z = torch.addmm(bias, x, B)                                         # [n, D + 4]: numerator | denominator
    gcn = rs = None
    if csr is not None:
        ax = gcn_aggregate(csr, x3, None, 1.0, 1.0).reshape(n, C)       # sliced or gather kernels, whichever the graph takes
        if Wv is not None:
            gcn, rs = torch.mm(ax, Wv.t()), csr.row_sums()
        else:
            gcn = ax
    return be.layer_tail_mix(Z, D, D, attn_scale, gcn, gcn_scale, rs, bv if rs is not None else None, x0,
                             x if residual else None, alpha, ln_weight, ln_bias, eps)


# ------------------------------------------------------------------------------------------
# f4: batch of graphs stored back to back (physical particle/difformer-v2.py:71-137)
# ------------------------------------------------------------------------------------------
class BatchLayout:
    """Index tables of a batch, derived once from `n_nodes` [B] (node count per graph), all on the device:
        graph_ptr    int32 [B+1]   graph b = rows [graph_ptr[b], graph_ptr[b+1])
        ranked_first int32 [B]     first row of the r-th largest graph        (sigmoid kernel)
        pos_count    int32 [maxn]  number of graphs with more than p nodes    (sigmoid kernel)
    They replace the reference's padding machinery (make_batch_mask / make_batch / to_pad, difformer-v2.py:8-27).
    Only B-sized integer bookkeeping happens here; the node-sized work is in the kernels."""

    def __init__(self, n_nodes, device):
        n = n_nodes.to(device=device, dtype=torch.int64).reshape(-1)
        if n.numel() == 0:
            raise ValueError("difformer_amd: n_nodes is empty")
        ptr = torch.zeros(n.numel() + 1, dtype=torch.int64, device=device)
        torch.cumsum(n, 0, out=ptr[1:])
        self.n_graphs = int(n.numel())
        total, biggest, smallest = (int(v) for v in torch.stack([ptr[-1], n.max(), n.min()]).tolist())   # one sync
        if smallest < 0:
            raise ValueError("difformer_amd: negative entry in n_nodes")
        if total >= 2 ** 31:
            raise ValueError("difformer_amd: batches are limited to 2^31 - 1 nodes")
        self.n_rows, self.max_nodes = total, biggest
        self.graph_ptr = ptr.to(torch.int32)
        order = torch.argsort(n, descending=True, stable=True)
        self.ranked_first = ptr[:-1][order].to(torch.int32).contiguous()
        le = torch.cumsum(torch.bincount(n, minlength=biggest + 1), 0)        # le[p] = #{graphs with n <= p}
        self.pos_count = (self.n_graphs - le[:biggest]).to(torch.int32).contiguous()


class _LayoutCache:
    """One BatchLayout per `n_nodes` tensor (identity + version), so the per-layer calls of one forward and
    repeated forwards over the same batch do the bookkeeping (and its host sync) once."""

    def __init__(self, capacity=16):
        self.capacity = capacity
        self.entries = OrderedDict()

    def get(self, n_nodes, device):
        if not torch.is_tensor(n_nodes):
            n_nodes = torch.as_tensor(n_nodes)
            return BatchLayout(n_nodes, device)
        key = (id(n_nodes), n_nodes.data_ptr(), tuple(n_nodes.shape), tensor_version(n_nodes), str(device))
        hit = self.entries.get(key)
        if hit is not None:
            ref, lay = hit
            if ref() is n_nodes:
                self.entries.move_to_end(key)
                return lay
            del self.entries[key]
        lay = BatchLayout(n_nodes, device)
        self.entries[key] = (weakref.ref(n_nodes), lay)
        while len(self.entries) > self.capacity:
            self.entries.popitem(last=False)
        return lay

    def clear(self):
        self.entries.clear()


layout_cache = _LayoutCache()


def _check_batch(qs, ks, vs, layout):
    if not (qs.shape[0] == ks.shape[0] == vs.shape[0] == layout.n_rows):
        raise RuntimeError(f"difformer_amd: n_nodes sums to {layout.n_rows} but q/k/v have "
                           f"{qs.shape[0]}/{ks.shape[0]}/{vs.shape[0]} rows (difformer-v2.py:26 would raise too)")


def batched_simple_attention(qs, ks, vs, layout: BatchLayout):
    """qs, ks [N,H,M], vs [N,H,D] -> [N,H,D]; attention inside each graph.  difformer-v2.py:80-111."""
    _check_batch(qs, ks, vs, layout)
    if layout.n_graphs == 1:          # one graph: exactly a1, whose reduce spreads the rows over the whole chip
        return simple_attention(qs, ks, vs)
    return get_backend().batched_simple_attention(qs, ks, vs, layout.graph_ptr)


def batched_sigmoid_attention(qs, ks, vs, layout: BatchLayout):
    """Attention among the nodes at equal positions of the graphs.  difformer-v2.py:113-135."""
    _check_batch(qs, ks, vs, layout)
    return get_backend().batched_sigmoid_attention(qs, ks, vs, layout.ranked_first, layout.pos_count)


# ------------------------------------------------------------------------------------------
# a4 / a5 tail
# ------------------------------------------------------------------------------------------
def layer_tail(conv, x0=None, prev=None, alpha=0.5, ln_weight=None, ln_bias=None, eps=1e-5, relu=False):
    """conv [n,H,D] -> [n,D]: head mean (+x0) -> alpha-residual with prev -> LayerNorm (-> ReLU)."""
    return get_backend().layer_tail(conv, x0, prev, alpha, ln_weight, ln_bias, eps, relu)


def linear(x, weight, bias, ln_weight=None, ln_bias=None, eps=1e-5, relu=False):
    """x W^T + b (-> LayerNorm) (-> ReLU) (difformer.py:188-191, :208; csrc/skinny_linear.hip: C_in <= 128, long rows into
    <= 64 features; csrc/simple_layer_xwide.hip: wide rows into 65..416 features, see linear_xwide_covers)."""
    return get_backend().linear(x, weight, bias, ln_weight, ln_bias, eps, relu)


GRAM_COEFFS_MAX_ROWS = 24576        # 48 partial Gram records: beyond, the separate finalize launch is the cheaper sum
LINEAR_XWIDE_MIN_ROWS = 16384       # below: too few 128-row blocks for the chip (15,000 x 512 -> 300: 64 us against 62 for GEMM + tail)


def linear_xwide_covers(x, weight):
    """Shapes dif_linear_xwide_f32 takes (float32, split-bf16 products: not under DIFFORMER_EXACT_FP32): more than 128 input
    channels in one product (<= 416) or two halves (C_in % 8 == 0, C_in / 2 <= 416), 65..416 output features, both multiples
    of 4, enough rows to fill the chip (a workgroup owns 128 rows)."""
    if EXACT_FP32 or x.dtype != torch.float32 or weight.dtype != torch.float32 or x.dim() != 2:
        return False
    n, c_in = x.shape
    d = weight.shape[0]
    if n < LINEAR_XWIDE_MIN_ROWS or not (64 < d <= XWIDE_MAX) or d % 4 or c_in <= 128 or c_in % 4:
        return False
    return c_in <= XWIDE_MAX or (c_in % 8 == 0 and c_in // 2 <= XWIDE_MAX)
