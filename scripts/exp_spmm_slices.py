"""Experiment (profiles/r04_experiments.md section 4): the gather product of a mid-degree LARGE graph (full Pokec: 1.63 M rows,
19 entries per row, x = 418 MB: no cache holds it) swept in column slices, so that the gathered working set of a pass
(N x slice bytes) sits in the Infinity Cache: 1 x 64, 2 x 32, 4 x 16, 8 x 8 columns, from column views of x (row stride 256 B)
and from slice-major copies (dense working set)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops
from bench import make_graph, WORKLOADS

dev = torch.device("cuda:0")
n, pairs = WORKLOADS["pokec-full-s"][:2]
ei = make_graph(n, pairs, dev)
be = ops.get_backend()
csr = ops.csr_cache.get(ei, None, n, 256)
x = torch.randn(n, 64, device=dev)
args = (csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, csr.nnz)
order = csr.row_order(0, n)


def timed(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


ref = be.spmm(*args, x, 0, n, None, 1.0, 1.0, None, order)
print(f"n_blocks {csr.n_blocks}, longest row {csr.max_degree()}; one pass, 64 columns: {timed(lambda: be.spmm(*args, x, 0, n, None, 1.0, 1.0, None, order)):.0f} us", flush=True)
for w in (32, 16, 8):
    views = [x[:, c: c + w] for c in range(0, 64, w)]
    dense = [v.contiguous() for v in views]
    tv = timed(lambda: [be.spmm(*args, v, 0, n, None, 1.0, 1.0, None, order) for v in views])
    td = timed(lambda: [be.spmm(*args, v, 0, n, None, 1.0, 1.0, None, order) for v in dense])
    out = torch.cat([be.spmm(*args, v, 0, n, None, 1.0, 1.0, None, order) for v in dense], dim=1)
    print(f"{64 // w} passes of {w} columns: views {tv:.0f} us, slice-major copies {td:.0f} us (max diff {float((out - ref).abs().max()):.1e})", flush=True)
