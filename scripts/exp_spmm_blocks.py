"""Experiment: blocked SpMM time on the C4 graph vs number of source blocks."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops
from bench import make_graph
dev = torch.device("cuda:0")
n = 132534
ei = make_graph(n, 39561252, dev)
e = ei.shape[1]
be = ops.get_backend()
x = torch.randn(n, 64, device=dev)
blocks = [int(b) for b in sys.argv[1:]] or [1, 6, 8, 10, 13, 16, 20, 26, 32]
for nb in blocks:
    csr = ops.GraphCSR.build(ei, None, n, nb)
    for _ in range(2):
        be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, e, x, 0, n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, e, x, 0, n)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"n_blocks {nb:3d} ({n*256/nb/2**20:5.2f} MiB/block, {e/n/nb:6.1f} entries/group): {dt*1e3:.3f} ms", flush=True)
    del csr
