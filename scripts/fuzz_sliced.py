"""Differential fuzz of the feature-sliced product against the gather kernel (same CSR): random sizes, widths, degree
profiles, row shards, attention combine.  Not a test (minutes of GPU time); run by hand: python scripts/fuzz_sliced.py [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops

dev = torch.device("cuda:0")
be = ops.get_backend()
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = 0.0
for it in range(cases):
    n = int(torch.randint(8192, 45000, (1,), generator=g))
    deg = int(torch.randint(48, 130, (1,), generator=g))
    F = [4, 8, 16, 32, 64, 64, 64, 128, 256][int(torch.randint(0, 9, (1,), generator=g))]
    mode = int(torch.randint(0, 4, (1,), generator=g))          # 0/1 uniform, 2 hubs, 3 power law
    e = n * deg
    src = torch.randint(0, n, (e,), generator=g)
    if mode <= 1:
        dst = torch.randint(0, n, (e,), generator=g)
    elif mode == 2:
        dst = torch.randint(0, n, (e,), generator=g)
        hubs = int(torch.randint(1, 60, (1,), generator=g))
        dst[: e // 3] = torch.randint(0, hubs, (e // 3,), generator=g) * (n // hubs)
    else:
        dst = (torch.rand(e, generator=g) ** 2.5 * n).long().clamp_(max=n - 1)
    dst[: n // 7] = n - 1 - torch.arange(n // 7)                 # some rows get at least one entry, rest may be empty
    ei = torch.stack([src, dst]).to(dev)
    lo = int(torch.randint(0, n // 2, (1,), generator=g)) if it % 3 == 0 else 0
    cnt = int(torch.randint(1000, n - lo, (1,), generator=g)) if it % 3 == 0 else n
    if it % 6 == 3:                                              # a thin shard: the plan splits the source tiles
        cnt = max(1000, cnt // int(torch.randint(3, 9, (1,), generator=g)))

    class _Rows:                                                 # what csr_cache.get reads of a RowShard: the tiling of a shard
        world, n_local, counts, offsets, row_begin = 2, cnt, [cnt, n - cnt], [0, cnt, n], lo
    csr = ops.csr_cache.get(ei, None, n, F * 4, _Rows() if cnt < n else None)
    sl = csr.sliced(lo, cnt, F)
    x = torch.randn(n, F, generator=g).to(dev)
    a = torch.randn(cnt, F, generator=g).to(dev) if it % 2 else None
    ref = be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, csr.nnz, x, lo, cnt, a, 0.7, 1.3, None,
                  csr.row_order(lo, cnt))
    if sl is None:
        print(f"case {it}: n={n} deg={deg} F={F} mode={mode} rows=[{lo},{lo + cnt}) -> declined", flush=True)
        ops.csr_cache.clear()
        continue
    ys = be.sliced_prescale(x, csr.rowptr, n, sl.plan)
    out = be.sliced_spmm(sl, ys, csr.rowptr, n, lo, cnt, F, a, 0.7, 1.3)
    out2 = be.sliced_spmm(sl, ys, csr.rowptr, n, lo, cnt, F, a, 0.7, 1.3)
    err = float((out - ref).abs().max() / ref.abs().max())
    worst = max(worst, err)
    plan = [int(v) for v in sl.plan] + [int(be.lib.dif_sliced_spmm_workspace_bytes(n, cnt if sl.n_pos is None else sl.n_pos, F) > 0)]
    print(f"case {it}: n={n} deg={deg} F={F} mode={mode} rows=[{lo},{lo + cnt}) order={'yes' if sl.order is not None else 'no'} "
          f"plan={plan} err={err:.2e} bitwise={bool(torch.equal(out, out2))}", flush=True)
    assert err < 2e-5 and torch.equal(out, out2), "MISMATCH"
    ops.csr_cache.clear()
print("worst", worst)
