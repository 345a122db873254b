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
The collectives go through torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo"
in the CPU tests).  The reduce buffer is latency-bound (far below the per-link bandwidth
regime), the all-gather moves N*H*D*4 bytes per layer in one call.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.distributed as dist


def split_rows(n_global: int, world: int) -> List[int]:
    """Contiguous block sizes: every rank but the last takes c = ceil(n_global / world) rows, the last the
    remainder.  All blocks then start at multiples of c, so an all-gather of c-row (zero-padded) blocks lands
    every node at its global row index -- no compaction copy after the collective."""
    c = -(-n_global // world)
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

    def __post_init__(self):
        if not self.counts:
            self.counts = split_rows(self.n_global, self.world)
        if sum(self.counts) != self.n_global or len(self.counts) != self.world:
            raise ValueError("RowShard: counts must have one entry per rank and sum to n_global")
        self.offsets = [0]
        for c in self.counts:
            self.offsets.append(self.offsets[-1] + c)

    @classmethod
    def from_process_group(cls, n_global: int, group=None):
        if not dist.is_initialized():
            return cls(n_global)
        return cls(n_global, dist.get_rank(group), dist.get_world_size(group), group)

    @property
    def row_begin(self) -> int:
        return self.offsets[self.rank]

    @property
    def n_local(self) -> int:
        return self.counts[self.rank]

    def local_rows(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.row_begin:self.row_begin + self.n_local]

    # ---- the two exchange steps -----------------------------------------------------------
    def all_reduce_sum(self, buf: torch.Tensor) -> torch.Tensor:
        """In-place sum over ranks of the small `reduced` record of the simple kernel."""
        if self.world > 1:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        return buf

    def all_gather_rows(self, local: torch.Tensor) -> torch.Tensor:
        """[n_local, ...] on every rank -> [n_global, ...] everywhere (rank order = row order)."""
        if self.world == 1:
            return local
        local = local.contiguous()
        tail = tuple(local.shape[1:])
        c = max(self.counts)
        uniform = all(self.offsets[r] == r * c for r in range(self.world))
        if uniform:
            # blocks start at multiples of c (split_rows): gather c-row blocks straight into place; the rows past
            # n_global at the end of the buffer are padding that no CSR entry refers to
            if self.n_local != c:
                padded = torch.zeros((c,) + tail, dtype=local.dtype, device=local.device)
                padded[: self.n_local] = local
                local = padded
            full = torch.empty((self.world * c,) + tail, dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(full, local, group=self.group)
            return full[: self.n_global]
        # arbitrary user-supplied blocks: gather equal-sized padded blocks, then compact
        padded = torch.zeros((c,) + tail, dtype=local.dtype, device=local.device)
        padded[: self.n_local] = local
        buf = torch.empty((self.world * c,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(buf, padded, group=self.group)
        return torch.cat([buf[r * c: r * c + self.counts[r]] for r in range(self.world)], dim=0)
