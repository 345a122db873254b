"""Experiment: blocked SpMM (13 blocks, C4) with every gather folded into the first block's rows: what would a
perfectly L2-resident sweep cost with the same group structure?"""
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
for nb in (13, 20):
    csr = ops.GraphCSR.build(ei, None, n, nb)
    block_rows = -(-n // nb)
    def run(tag):
        f = lambda: be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, e, x, 0, n)
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize()
        print(f"n_blocks {nb} {tag:30s} {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
    run("true sources")
    csr.src = csr.src % block_rows
    run("sources folded into block 0")
    csr.src = csr.src % 1024
    run("sources folded into 1024 rows")
