"""Slot order of the feature-sliced product on the C4 graph (the bounded attempt on its 1.56x padding): rows
that share a 64-row slot chosen by (largest per-tile block count, set of tiles that reach it) instead of natural order.
Prints, for both orders: the block count the row envelopes alone predict, the block count the format actually builds, and the
product's time.  Result (profiles/r06_experiments.md section 3): the order cuts the ROW envelope from 1.52 to 1.25 lane-steps
per entry, and the built format does not move (1.556 -> 1.569): a step must also hit 16 distinct bank quads per 16-row lane
group, and the largest bank-quad COLUMN of a lane group (a sum over its 16 rows, 46 +- 6.8 like a row) is the binding envelope.
    python scripts/exp_slot_order.py          (GPU)        python scripts/exp_sliced_slots.py    (the offline statistics)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_graph
from difformer_amd import ops
dev = torch.device("cuda:0")
n, pairs, F = 132534, 39561252, 64
ei = make_graph(n, pairs, dev)
be = ops.get_backend()
x = torch.randn(n, F, device=dev)
csr = ops.csr_cache.get(ei, None, n, F * 4)
plan = be.sliced_plan(n, n, F)
slices, panels, G, PW, W, R, T, NT = (int(v) for v in plan)


def slot_order():
    cnt = csr.blkptr.view(NT + 1, n)
    blocks = (cnt[1:] - cnt[:-1] + 7) // 8                                   # [tiles, rows] blocks of 8 steps
    top = blocks.max(dim=0).values.to(torch.int64)
    weights = torch.ones(NT, dtype=torch.int64, device=dev) << torch.arange(NT, device=dev)
    hot = ((blocks == top[None, :]).to(torch.int64) * weights[:, None]).sum(dim=0)
    key = (int(top.max().item()) - top) * (1 << NT) + hot
    return torch.argsort(key, stable=True).to(torch.int32)


def predicted(order):
    """blocks of the format if a round were as long as its longest ROW only"""
    cnt = csr.blkptr.view(NT + 1, n)
    cnt = (cnt[1:] - cnt[:-1]).t().contiguous()
    o = torch.full((G * 64,), -1, dtype=torch.int64, device=dev)
    o[:n] = order.long() if order is not None else torch.arange(n, device=dev)
    c = torch.where(o[:, None] >= 0, cnt[o.clamp(min=0)], torch.zeros_like(cnt[:1])).view(G, 64, NT)
    nb = (c.max(dim=1).values + 7) // 8
    j = torch.arange(R, device=dev)[:, None]
    pw = torch.arange(PW, device=dev)[None, :]
    s = j * PW + torch.where(j % 2 == 1, PW - 1 - pw, pw)
    r = torch.where((s < G)[..., None], nb[s.clamp(max=G - 1)], torch.zeros_like(nb[:1]))
    r = torch.flip(torch.cummax(torch.flip(r, [0]), 0).values, [0])
    return int(r.sum())


for name, order in (("natural order", None), ("ordered slots", slot_order()), ("natural order", None), ("ordered slots", slot_order())):
    built = be.sliced_build(csr.rowptr, csr.blkptr, csr.src, n, csr.nnz, 0, n, F, plan, order, None, None)
    sl = ops.SlicedAdjacency(plan, built[0], built[1], order, None, None)
    nb = sl.entries.numel() // 512
    ys = be.sliced_prescale(x, csr.rowptr, n, sl.plan)
    for _ in range(5):
        out = be.sliced_spmm(sl, ys, csr.rowptr, n, 0, n, F, None, 1.0, 1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        out = be.sliced_spmm(sl, ys, csr.rowptr, n, 0, n, F, None, 1.0, 1.0)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: row envelopes alone {predicted(order) * 512 / csr.nnz:.3f} lane-steps per entry, built format {nb * 512 / csr.nnz:.3f} "
          f"({nb} blocks), product {e0.elapsed_time(e1) / 30 * 1e3:.1f} us", flush=True)
