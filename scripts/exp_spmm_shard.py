"""Experiment: blocked SpMM over a row shard (what one of N ranks runs) on the C4 graph."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops
from difformer_amd.dist import split_rows
from bench import make_graph
dev = torch.device("cuda:0")
n = 132534
ei = make_graph(n, 39561252, dev)
e = ei.shape[1]
be = ops.get_backend()
x = torch.randn(n, 64, device=dev)
csr = ops.GraphCSR.build(ei, None, n, 13)
for world in (1, 2, 4, 8):
    cnt = split_rows(n, world)[0]
    f = lambda: be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, e, x, 0, cnt)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"world {world}: {cnt} rows  {dt*1e3:.3f} ms  (x{world} = {dt*world*1e3:.3f} ms)", flush=True)
