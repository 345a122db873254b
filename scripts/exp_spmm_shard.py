"""Experiment: blocked SpMM over a row shard (what one of N ranks runs) on the C4 graph: one launch over all source
blocks vs the split product of the sharded path (part 0 = the rank's own blocks, runs under the all-gather; part 1 = the
rest + epilogue), with block boundaries aligned to the rank boundaries."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops
from difformer_amd.dist import RowShard
from bench import make_graph
dev = torch.device("cuda:0")
n = 132534
ei = make_graph(n, 39561252, dev)
e = ei.shape[1]
be = ops.get_backend()
x = torch.randn(n, 64, device=dev)
a = torch.randn(n, 64, device=dev)


def timeit(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


for world in (1, 2, 4, 8):
    sh = RowShard(n, rank=0, world=world)
    al = ops.choose_shard_blocks(n, 256, e, sh)
    nb, rows = al if al else (ops.choose_source_blocks(n, 256, e), 0)
    csr = ops.GraphCSR.build(ei, None, n, nb, block_rows=rows)
    args = (csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, e)
    for rank in sorted({0, world - 1}):
        s = RowShard(n, rank=rank, world=world)
        lo, cnt = s.row_begin, s.n_local
        t_one = timeit(lambda: be.spmm(*args, x, lo, cnt, a[lo:lo + cnt], 1.0, 1.0, None, None))
        line = f"world {world} rank {rank}: {cnt} rows, {nb} blocks  one launch {t_one:.3f} ms"
        if al:
            own_lo, own_hi = lo // csr.block_rows, -(-(lo + cnt) // csr.block_rows)
            own = x[lo:lo + cnt].contiguous()
            scratch = be.spmm(*args, own, lo, cnt, None, 1.0, 1.0, None, None, (0, own_lo, own_hi, None, lo))
            t0 = timeit(lambda: be.spmm(*args, own, lo, cnt, None, 1.0, 1.0, None, None, (0, own_lo, own_hi, scratch, lo)))
            t1 = timeit(lambda: be.spmm(*args, x, lo, cnt, a[lo:lo + cnt], 1.0, 1.0, None, None, (1, own_lo, own_hi, scratch, 0)))
            line += f"   split: part 0 (own blocks, hidden under the all-gather) {t0:.3f} + part 1 {t1:.3f} = {t0 + t1:.3f} ms"
        print(line, flush=True)
    del csr
