"""Experiment: the sliced product on a graph with COMMUNITY structure (the real ogbn-proteins has 8 species whose proteins
interact almost only among themselves; node ids are grouped by species): nodes in `nb` contiguous blocks, a fraction `intra`
of every node's edges stays inside its block.  Same size and mean degree as the C4 bench graph.
    python scripts/exp_sliced_blocks.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from difformer_amd import ops

dev = torch.device("cuda:0")
n, C, pairs = 132534, 64, 39561252
be = ops.get_backend()


def block_graph(nb, intra, skew, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    sizes = torch.full((nb,), n // nb, dtype=torch.int64)
    sizes[: n % nb] += 1
    if skew:                                              # unequal species
        w = torch.arange(1, nb + 1, dtype=torch.float64) ** 1.5
        sizes = (w / w.sum() * n).long()
        sizes[-1] += n - sizes.sum()
    start = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(sizes, 0)[:-1]]).to(dev)
    sizes = sizes.to(dev)
    a = torch.randint(0, n, (pairs,), generator=g, device=dev)
    blk = torch.searchsorted(torch.cumsum(sizes, 0), a, right=True).clamp_(max=nb - 1)
    inside = torch.rand(pairs, generator=g, device=dev) < intra
    b_in = start[blk] + (torch.rand(pairs, generator=g, device=dev, dtype=torch.float64) * sizes[blk]).long().clamp_(max=n - 1)
    b_out = torch.randint(0, n, (pairs,), generator=g, device=dev)
    b = torch.where(inside, b_in, b_out)
    loops = torch.arange(n, device=dev)
    return torch.stack([torch.cat([a, b, loops]), torch.cat([b, a, loops])]).contiguous()


def run(name, ei):
    x = torch.randn(n, C, device=dev)
    ops.csr_cache.clear()
    csr = ops.csr_cache.get(ei, None, n, C * 4)
    sl = csr.sliced(0, n, C)
    ref = be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, csr.nnz, x, 0, n, None, 1.0, 1.0, None, csr.row_order(0, n))
    if sl is None:
        print(f"{name}: declined"); return
    ys = be.sliced_prescale(x, csr.rowptr, n, sl.plan)
    f = lambda: be.sliced_spmm(sl, ys, csr.rowptr, n, 0, n, C, None, 1.0, 1.0)
    out = f()
    err = float((out - ref).abs().max() / ref.abs().max())
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    g = lambda: be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, csr.nnz, x, 0, n, None, 1.0, 1.0, None, csr.row_order(0, n))
    g(); c, d = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c.record()
    for _ in range(5): g()
    d.record(); torch.cuda.synchronize()
    plan = [int(v) for v in sl.plan]
    nblk = int(sl.table[-1])
    print(f"{name}: sliced {a.elapsed_time(b) / 20 * 1e3:.0f} us ({nblk * 512 / csr.nnz:.2f} slots per entry, order={'yes' if sl.order is not None else 'no'}, "
          f"parts={'yes' if sl.parts is not None else 'no'}, R={plan[5]}), gather kernel {c.elapsed_time(d) / 5 * 1e3:.0f} us, err {err:.1e}", flush=True)


run("uniform", block_graph(1, 0.0, False))
run("8 equal blocks, 90 % inside", block_graph(8, 0.9, False))
run("8 equal blocks, 99 % inside", block_graph(8, 0.99, False))
run("8 unequal blocks, 95 % inside", block_graph(8, 0.95, True))
run("40 blocks, 95 % inside", block_graph(40, 0.95, False))

# the whole model on the block graph: natural order against the mixed order the model picks itself
from difformer_amd import DIFFormer
torch.manual_seed(0)
model = DIFFormer(8, 64, 112, num_layers=4, kernel="simple").to(dev).eval()
xin = torch.randn(n, 8, device=dev)
for name, ei in (("uniform", block_graph(1, 0.0, False)), ("8 equal blocks, 95 % inside", block_graph(8, 0.95, False))):
    for thr in (1e9, ops.MIX_THRESHOLD):
        ops.MIX_THRESHOLD, saved = thr, ops.MIX_THRESHOLD
        ops.csr_cache.clear(); ops.mix_cache.clear()
        with torch.no_grad():
            for _ in range(3): model(xin, ei)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): model(xin, ei)
            b.record(); torch.cuda.synchronize()
        print(f"model forward, {name}, {'natural order' if thr > 100 else 'mixed if structured'}: {a.elapsed_time(b) / 20:.3f} ms "
              f"(mixed: {ops.mix_cache.get(ei, n, 64) is not None})", flush=True)
        ops.MIX_THRESHOLD = saved
