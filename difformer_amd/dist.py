"""Row sharding of the propagation layer over the GPUs of one node (one process per GPU).

SURVEY.md section 8e: nodes are split into contiguous row blocks; every per-node operation
(projections, LayerNorm, residual, head mean) is row-local, and each operator has exactly one
exchange step:
  a1 simple attention : local reduce -> ONE all-reduce(sum) of H*(M*D+M+D)+2 floats
                        (KtV, ksum, vsum, sum q^2, sum k^2; 16.9 KB at H=1, d=64) -> local apply
                        with the GLOBAL N in the denominator (difformer.py:22,38)
  a3 gcn_conv         : destination rows are local but sources are arbitrary -> ONE all-gather of
                        the value rows, then a local SpMM over this rank's CSR row range
  a2 sigmoid attention: query rows local; all-gather K and V; row sums are local
Closed-form layers (one head, inference) can also split the aggregation by FEATURE SLICES instead of rows
(`RowShard.product = "slice"`): rank p multiplies all N rows of columns [p C/P, (p+1) C/P) -- an all-to-all of its rows'
other column blocks in, one of the other ranks' rows of its block out (2 x N C 4 (P-1)/P^2 bytes per rank and layer,
7.4 MB at C4 on 8 ranks) instead of an all-gather that delivers N C 4 (P-1)/P bytes (30 MB) to every rank.
The collectives go through torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo"
in the CPU tests).  The reduce buffer is latency-bound (far below the per-link bandwidth
regime), the all-gather moves N*H*D*4 bytes per layer in one call.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.distributed as dist


def split_rows(n_global: int, world: int) -> List[int]:
    """Contiguous block sizes: every rank but the last takes c = ceil(n_global / world) rows, the last the
    remainder.  All blocks then start at multiples of c, so an all-gather of c-row (zero-padded) blocks lands
    every node at its global row index -- no compaction copy after the collective."""
    c = -(-n_global // world)
    if n_global >= 64 * world * world:
        # a multiple of 8, so that the blocked SpMM can cut every rank's rows into 1, 2, 4 or 8 whole source blocks
        # (block boundaries on rank boundaries: the product over a rank's OWN value rows runs before the all-gather lands)
        c = -(-c // 8) * 8
    counts = []
    left = n_global
    for _ in range(world):
        take = min(c, left)
        counts.append(take)
        left -= take
    return counts


@dataclass
class RowShard:
    """This rank's contiguous block of node rows."""
    n_global: int
    rank: int = 0
    world: int = 1
    group: Optional[object] = None
    counts: List[int] = field(default_factory=list)
    # second communicator for the small all-reduce, so it can run while the all-gather of the value rows
    # (issued first, on `group`) is still in flight; None = use `group` for both (serialised)
    side_group: Optional[object] = None
    # how the aggregation of a closed-form layer is split over the ranks: "row" = every rank gathers ALL source rows and
    # multiplies its own destination rows (one all-gather of N*C elements per layer); "slice" = every rank multiplies ALL
    # rows of ITS feature columns (two all-to-alls of N*C/world elements per rank and layer).  DIFFORMER_SHARD_PRODUCT.
    product: str = field(default_factory=lambda: os.environ.get("DIFFORMER_SHARD_PRODUCT", "row"))
    # diagnostics (bench.py --gpus N): a list makes every exchange step append (name, begin, end) -- HIP events recorded on
    # the compute stream around the call (what the compute stream WAITED, i.e. the exposed part of the collective), or
    # perf_counter seconds for host tensors; None = no bookkeeping
    timeline: Optional[list] = None

    def __post_init__(self):
        if not self.counts:
            self.counts = split_rows(self.n_global, self.world)
        if sum(self.counts) != self.n_global or len(self.counts) != self.world:
            raise ValueError("RowShard: counts must have one entry per rank and sum to n_global")
        if self.world > 1 and min(self.counts) <= 0:
            # a rank without rows would fail its kernels' argument checks while the others block in the collectives
            raise ValueError(f"RowShard: every rank needs at least one row (n_global={self.n_global}, world={self.world} "
                             f"gives blocks {self.counts}); use fewer ranks or pass explicit counts")
        self.offsets = [0]
        for c in self.counts:
            self.offsets.append(self.offsets[-1] + c)

    @classmethod
    def from_process_group(cls, n_global: int, group=None):
        if not dist.is_initialized():
            return cls(n_global)
        world = dist.get_world_size(group)
        side = None
        # Two communicators in flight at once (the record all-reduce on its own one while the all-gather of the value
        # rows runs) is an optimisation torch.distributed documents as unsafe unless the user orders the collectives;
        # it has not been validated on multi-GPU hardware here, so it is opt-in.  The default shares `group`: the
        # two collectives are then issued in the same order on every rank and serialise.
        if world > 1 and os.environ.get("DIFFORMER_OVERLAP_COLLECTIVES", "0") == "1":
            ranks = dist.get_process_group_ranks(group) if group is not None else list(range(world))
            side = dist.new_group(ranks=ranks)          # collective: every rank of `group` gets here; errors propagate
        return cls(n_global, dist.get_rank(group), world, group, side_group=side)

    @property
    def row_begin(self) -> int:
        return self.offsets[self.rank]

    @property
    def n_local(self) -> int:
        return self.counts[self.rank]

    def local_rows(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.row_begin:self.row_begin + self.n_local]

    def _mark(self, t: torch.Tensor):
        """A point on the compute stream (HIP event) or on the host clock, for `timeline`."""
        if self.timeline is None:
            return None
        if t.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(t.device))
            return ev
        import time
        return time.perf_counter()

    def _span(self, name: str, begin, t: torch.Tensor):
        if self.timeline is not None and begin is not None:
            self.timeline.append((name, begin, self._mark(t)))

    def timeline_ms(self):
        """{name: total milliseconds} of the recorded exchange steps (synchronises the device)."""
        out = {}
        if not self.timeline:
            return out
        if any(not isinstance(b, float) for _, b, _ in self.timeline):
            torch.cuda.synchronize()
        for name, b, e in self.timeline:
            out[name] = out.get(name, 0.0) + ((e - b) * 1e3 if isinstance(b, float) else b.elapsed_time(e))
        return out

    # ---- the two exchange steps -----------------------------------------------------------
    def all_reduce_sum(self, buf: torch.Tensor) -> torch.Tensor:
        """In-place sum over ranks of the small `reduced` record of the simple kernel."""
        if self.world > 1:
            t0 = self._mark(buf)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM,
                            group=self.side_group if self.side_group is not None else self.group)
            self._span("all_reduce(record)", t0, buf)
        return buf

    def all_reduce_gradients(self, params) -> None:
        """Training on row shards: every rank holds the parameter gradients of ITS rows' share of the loss; the
        gradient of the summed loss is their sum over ranks (one flat all-reduce; call it between backward() and
        optimizer.step(); divide the loss or the learning rate yourself if the reference's loss is a mean)."""
        grads = [p.grad for p in params if p is not None and p.grad is not None]
        if self.world <= 1 or not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        self.all_reduce_sum(flat)
        off = 0
        for g in grads:
            g.copy_(flat[off: off + g.numel()].view_as(g))
            off += g.numel()

    def all_reduce_sum_async(self, buf: torch.Tensor):
        """Start the in-place sum of the record and return the work handle (None when there is nothing to wait for)."""
        if self.world <= 1:
            return None
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM,
                               group=self.side_group if self.side_group is not None else self.group, async_op=True)

    # ---- slice-sharded product: columns out, columns back ------------------------------------
    def slice_width(self, C: int) -> int:
        """Feature columns per rank of the slice-sharded product, or 0 when C does not split into 16-byte slices."""
        return C // self.world if (self.world > 1 and C % (4 * self.world) == 0) else 0

    def all_to_all_columns(self, local: torch.Tensor) -> torch.Tensor:
        """[n_local, C] (this rank's rows) -> [n_global, C / world]: column block `rank` of EVERY row.  Rank r sends
        rank p the columns p*C/world .. of its rows: (world - 1) / world of N*C/world elements in and out per rank."""
        P, n = self.world, self.n_local
        w = local.shape[1] // P
        send = local.reshape(n, P, w).permute(1, 0, 2).contiguous()            # [P, n_local, w]: destination-major
        out = torch.empty((self.n_global, w), dtype=local.dtype, device=local.device)
        t0 = self._mark(local)
        dist.all_to_all_single(out, send.reshape(P * n, w), output_split_sizes=list(self.counts),
                               input_split_sizes=[n] * P, group=self.group)
        self._span("all_to_all(columns)", t0, local)
        return out

    def all_to_all_rows(self, cols: torch.Tensor) -> torch.Tensor:
        """The inverse exchange: [n_global, C / world] (this rank's column block of every row) -> [n_local, C]."""
        P, n = self.world, self.n_local
        w = cols.shape[1]
        recv = torch.empty((P * n, w), dtype=cols.dtype, device=cols.device)
        t0 = self._mark(cols)
        dist.all_to_all_single(recv, cols.contiguous(), output_split_sizes=[n] * P, input_split_sizes=list(self.counts),
                               group=self.group)
        self._span("all_to_all(rows)", t0, cols)
        return recv.reshape(P, n, w).permute(1, 0, 2).reshape(n, P * w)

    def all_gather_rows_async(self, local: torch.Tensor):
        """Start the all-gather of the value rows and return a handle; `handle.wait()` gives the gathered tensor.
        Lets the record all-reduce + the apply kernel run while the 4*N*H*D bytes move over xGMI."""
        return _GatherHandle(self, local)

    def all_gather_rows(self, local: torch.Tensor) -> torch.Tensor:
        """[n_local, ...] on every rank -> [n_global, ...] everywhere (rank order = row order)."""
        return _GatherHandle(self, local).wait()


class _GatherHandle:
    """All-gather of row blocks, possibly still in flight."""

    def __init__(self, shard: RowShard, local: torch.Tensor):
        self.shard, self.work, self.post = shard, None, None
        if shard.world == 1:
            self.result = local
            return
        local = local.contiguous()
        tail = tuple(local.shape[1:])
        c = max(shard.counts)
        uniform = all(shard.offsets[r] == r * c for r in range(shard.world))
        if shard.n_local != c:
            padded = torch.zeros((c,) + tail, dtype=local.dtype, device=local.device)
            padded[: shard.n_local] = local
            local = padded
        buf = torch.empty((shard.world * c,) + tail, dtype=local.dtype, device=local.device)
        self._issued = shard._mark(local)
        self.work = dist.all_gather_into_tensor(buf, local, group=shard.group, async_op=True)
        self._keep = local
        if uniform:
            # blocks start at multiples of c (split_rows): every node already sits at its global row index; the rows
            # past n_global at the end of the buffer are padding that no CSR entry refers to
            self.post = lambda: buf[: shard.n_global]
        else:
            # arbitrary caller-supplied blocks: compact the padded blocks
            self.post = lambda: torch.cat([buf[r * c: r * c + shard.counts[r]] for r in range(shard.world)], dim=0)

    def wait(self) -> torch.Tensor:
        if self.work is not None:
            t0 = self.shard._mark(self._keep)
            self.work.wait()
            self.work = None
            self.shard._span("all_gather(rows): exposed wait", t0, self._keep)
            self.shard._span("all_gather(rows): issue to landed", self._issued, self._keep)
            self.result = self.post()
        return self.result
