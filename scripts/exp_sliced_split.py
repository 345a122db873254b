"""Experiment: threshold for splitting hub rows of the feature-sliced format (ops.SLICED_SPLIT_FACTOR x mean degree) on
the Zipf-degree C4 graph: launch time, row positions, geometry.   python scripts/exp_sliced_split.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from difformer_amd import ops
from bench import make_graph

dev = torch.device("cuda:0")
n, C = 132534, 64
ei = make_graph(n, 39561252, dev, zipf=True)
be = ops.get_backend()
x = torch.randn(n, C, device=dev)
csr = ops.csr_cache.get(ei, None, n, C * 4)
ref = be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, csr.nnz, x, 0, n, None, 1.0, 1.0, None, csr.row_order(0, n))
for factor in (1e9, 6.0, 4.0, 3.0, 2.0, 1.5, 1.25, 1.0, 0.75):
    ops.SLICED_SPLIT_FACTOR = factor
    csr._sliced.clear()
    torch.cuda.synchronize()
    sl = csr.sliced(0, n, C)
    if sl is None:
        print(f"factor {factor}: declined"); continue
    ys = be.sliced_prescale(x, csr.rowptr, n, sl.plan)
    f = lambda: be.sliced_spmm(sl, ys, csr.rowptr, n, 0, n, C, None, 1.0, 1.0)
    out = f()
    err = float((out - ref).abs().max() / ref.abs().max())
    for _ in range(3): f()
    ts = []
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): f()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 20 * 1e3)
    plan = [int(v) for v in sl.plan]
    nblk = int(sl.table[-1])
    print(f"factor {factor:g}: {np.median(ts):.1f} us  positions {sl.n_pos or n}  panels={plan[1]} W={plan[4]} R={plan[5]}  "
          f"blocks {nblk} ({nblk * 512 / csr.nnz:.2f} slots per entry)  err {err:.1e}", flush=True)
