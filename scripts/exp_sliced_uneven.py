"""Experiment (profiles/r04_experiments.md section 1b): an UNEVEN deal of the 64-row slots over the waves of a workgroup.

The LDS serves a workgroup's waves oldest first: in every tile wave 0 reaches the barrier ~6 us before wave 14
(r02_experiments.md 3c).  Here the older waves get MORE rounds and the younger fewer, with the format as it is: the plan is
built for R = 10 rounds of every (panel, wave) pair over n_pos > n positions, and the slots a wave should not have are left
EMPTY (order = -1, the padding mechanism of the hub-row split) -- an empty last round has no blocks and costs no steps.
    python scripts/exp_sliced_uneven.py
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops
from bench import make_graph

dev = torch.device("cuda:0")
n, C = 132534, 64
zipf = len(sys.argv) > 1 and sys.argv[1] == "zipf"
torch.manual_seed(0)
ei = make_graph(n, 39561252, dev, zipf=zipf)
be = ops.get_backend()
x = torch.randn(n, C, device=dev)
csr = ops.csr_cache.get(ei, None, n, C * 4)
base = csr.sliced(0, n, C)
ys = be.sliced_prescale(x, csr.rowptr, n, base.plan)
ref = be.sliced_spmm(base, ys, csr.rowptr, n, 0, n, C, None, 1.0, 1.0)


def timed(sl):
    f = lambda: be.sliced_spmm(sl, ys, csr.rowptr, n, 0, n, C, None, 1.0, 1.0)
    out = f()
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            f()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 20 * 1e3)
    err = float((out - ref).abs().max() / ref.abs().max())
    return min(ts), sorted(ts)[2], err


def uneven(rounds):
    """rounds[w] = slots of wave w (every panel alike) -> SlicedAdjacency over n_pos = R * PW * 64 positions."""
    R = max(rounds)
    G_real = -(-n // 64)
    W = len(rounds)
    plan = be.sliced_plan(n, R * 16 * W * 64, C)
    slices, panels, G, PW, W2, R2, T, NT = (int(v) for v in plan)
    assert (W2, R2, PW) == (W, R, panels * W) and G == R * PW, (plan, rounds)
    assert sum(rounds) * panels >= G_real, (sum(rounds) * panels, G_real)
    j = torch.arange(R).view(R, 1)
    pw = torch.arange(PW).view(1, PW)
    slot = j * PW + torch.where(j % 2 == 1, PW - 1 - pw, pw)                   # slot_of(j, pw)
    real = (j < torch.tensor(rounds)[pw // panels])                             # wave = pw // panels
    real_slots = slot[real].sort().values[:G_real]                              # in slot order: rows in natural order
    order = torch.full((G * 64,), -1, dtype=torch.int32)
    rows = torch.arange(G_real * 64, dtype=torch.int32)
    rows[rows >= n] = -1
    order.view(G, 64)[real_slots] = rows.view(G_real, 64)
    parts = torch.full((G * 64,), 0x0100, dtype=torch.int16)
    order, parts = order.to(dev), parts.to(dev)
    built = be.sliced_build(csr.rowptr, csr.blkptr, csr.src, n, csr.nnz, 0, n, C, plan, order, parts, G * 64)
    assert built is not None
    return ops.SlicedAdjacency(plan, built[0], built[1], order, parts, G * 64)


lo, med, err = timed(base)
print(f"product's deal (plan {[int(v) for v in base.plan]}): {lo:.1f} / {med:.1f} us", flush=True)
profiles = {
    "even 9x10+5x8 via R=10 (control: same work per wave as the product)": [9] * 10 + [8] * 5,
    "10,10,10,10,9,9,9,9,8,8,8,8,8,7,7": [10, 10, 10, 10, 9, 9, 9, 9, 8, 8, 8, 8, 8, 7, 7],
    "10x5, 9x3, 8x4, 7x3": [10] * 5 + [9] * 3 + [8] * 4 + [7] * 3,
    "10x7, 9x2, 8x2, 7x2, 6x2": [10] * 7 + [9] * 2 + [8] * 2 + [7] * 2 + [6] * 2,
    "reverse (younger waves more): 7,7,8x5,9x4,10x4": [7, 7, 8, 8, 8, 8, 8, 9, 9, 9, 9, 10, 10, 10, 10],
}
for name, r in profiles.items():
    assert len(r) == 15
    try:
        sl = uneven(r)
        lo, med, err = timed(sl)
        blocks = int(sl.entries.numel()) // 512
        print(f"{name}: {lo:.1f} / {med:.1f} us (min / median of 5x20), {blocks * 512 / csr.nnz:.3f} padded lane-steps per entry, "
              f"max diff {err:.2e}", flush=True)
    except Exception as e:
        print(f"{name}: failed: {e}", flush=True)
lo, med, err = timed(base)
print(f"product's deal again: {lo:.1f} / {med:.1f} us", flush=True)
