"""Experiment: what ONE rank of an N-way row-sharded run does per layer on the C4 graph, timed on one GPU (no
collectives here: the all-gather of x and the all-reduce of the Gram record are the only exchange steps).
Feature-sliced product over the rank's destination rows (all source rows staged tile by tile), pre-scale pass over the
gathered rows, Gram pass / coefficients / layer kernel over the rank's rows."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops
from difformer_amd.dist import RowShard
from bench import make_graph
dev = torch.device("cuda:0")
n, C = 132534, 64
zipf = len(sys.argv) > 1 and sys.argv[1] == "zipf"
ei = make_graph(n, 39561252, dev, zipf=zipf)
be = ops.get_backend()
x = torch.randn(n, C, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
Wq, Wk, Wv = (torch.randn(C, C, device=dev, generator=g) / 8 for _ in range(3))
bq, bk, bv = (torch.randn(C, device=dev, generator=g) / 8 for _ in range(3))
lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)


def timeit(f, it=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


base = None
for world in (1, 2, 4, 8):
    for rank in sorted({0, world - 1}):
        s = RowShard(n, rank=rank, world=world)
        csr = ops.csr_cache.get(ei, None, n, C * 4, s)          # a shard has its own tiling (source splits)
        lo, cnt = s.row_begin, s.n_local
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sl = csr.sliced(lo, cnt, C)
        torch.cuda.synchronize(); t_build = (time.perf_counter() - t0) * 1e3
        assert sl is not None
        ys = be.sliced_prescale(x, csr.rowptr, n, sl.plan)
        t_pre = timeit(lambda: be.sliced_prescale(x, csr.rowptr, n, sl.plan))
        t_sp = timeit(lambda: be.sliced_spmm(sl, ys, csr.rowptr, n, lo, cnt, C, None, 1.0, 1.0))
        xl = x[lo:lo + cnt]
        t_gram = timeit(lambda: be.gram(xl, None, None))
        rec, _ = be.gram(xl, None, None)
        coef = be.simple_coeffs(rec, n, C, C, Wq, bq, Wk, bk, Wv, bv, 1.0)
        ax = be.sliced_spmm(sl, ys, csr.rowptr, n, lo, cnt, C, None, 1.0, 1.0)
        rs = csr.row_sums()[lo:lo + cnt]
        t_layer = timeit(lambda: be.simple_layer(xl, coef, C, ax, Wv, bv, rs, 1.0, None, True, 0.5, lw, lb, 1e-5, False))
        if base is None:
            base = t_sp
        plan = [int(v) for v in sl.plan]
        print(f"world {world} rank {rank}: {cnt} rows  plan panels={plan[1]} W={plan[4]} R={plan[5]} NT={plan[7]}  format build {t_build:.1f} ms | "
              f"sliced product {t_sp * 1e3:.0f} us ({t_sp / (base / world):.2f}x of 1/{world}), pre-scale of the gathered rows "
              f"{t_pre * 1e3:.0f} us, gram {t_gram * 1e3:.0f} us, layer kernel {t_layer * 1e3:.0f} us", flush=True)
