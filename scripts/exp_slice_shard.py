"""Experiment: what ONE rank of a SLICE-sharded closed-form layer does per layer at C4 (no collectives, one GPU): the
pre-scale and the sliced product of the WHOLE graph at 64 / P feature columns, plus the two re-layout copies around the
all-to-alls; beside it the row shard's product (1 / P of the destination rows at 64 columns) on the same box.
    python scripts/exp_slice_shard.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops
from difformer_amd.dist import RowShard
from bench import make_graph

dev = torch.device("cuda:0")
n, C = 132534, 64
torch.manual_seed(0)
ei = make_graph(n, 39561252, dev)
be = ops.get_backend()
x = torch.randn(n, C, device=dev)


def timed(f, reps=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print("| ranks P | slice shard: columns | plan {slices, panels, .., W, R, T, NT} | pre-scale us | product us | re-layout copies us | row shard: product of N/P rows us |")
print("|---|---|---|---|---|---|---|")
for P in (1, 2, 4, 8):
    w = C // P
    csr = ops.csr_cache.get(ei, None, n, w * 4)
    sl = csr.sliced(0, n, w)
    xs = x[:, :w].contiguous()
    ys = be.sliced_prescale(xs, csr.rowptr, n, sl.plan)
    t_pre = timed(lambda: be.sliced_prescale(xs, csr.rowptr, n, sl.plan))
    t_prod = timed(lambda: be.sliced_spmm(sl, ys, csr.rowptr, n, 0, n, w, None, 1.0, 1.0))
    nl = -(-n // P)
    xl = x[:nl]
    recv = torch.empty(P * nl, w, device=dev)
    t_copy = timed(lambda: (xl.reshape(nl, P, w).permute(1, 0, 2).contiguous(), recv.reshape(P, nl, w).permute(1, 0, 2).reshape(nl, C))) if P > 1 else 0.0
    t_row = float("nan")
    if P > 1:
        sh = RowShard(n, 0, P)
        csr_r = ops.csr_cache.get(ei, None, n, C * 4, sh)
        slr = csr_r.sliced(sh.row_begin, sh.n_local, C)
        if slr is not None:
            ysr = be.sliced_prescale(x, csr_r.rowptr, n, slr.plan)
            t_row = timed(lambda: be.sliced_spmm(slr, ysr, csr_r.rowptr, n, sh.row_begin, sh.n_local, C, None, 1.0, 1.0))
    else:
        t_row = t_prod
    print(f"| {P} | {w} | {[int(v) for v in sl.plan]} | {t_pre:.1f} | {t_prod:.1f} | {t_copy:.1f} | {t_row:.1f} |", flush=True)
    ops.csr_cache.clear()
