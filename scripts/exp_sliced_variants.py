"""Experiment: variants of the sliced product's tile phase on the C4 graph (DIF_SLICED_VARIANT: bit 0 = tile loads
issued before the barrier, bit 1 = XCDs start on different tiles, bit 3 = panels start on different tiles, bit 2 =
per-tile wall-clock stamps of the first and last wave of every workgroup).  One process per variant (the library reads the
variable once):  python scripts/exp_sliced_variants.py [zipf]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from difformer_amd import ops
from bench import make_graph

dev = torch.device("cuda:0")
n, C = 132534, 64
variant = int(os.environ.get("DIF_SLICED_VARIANT", "0"))
zipf = len(sys.argv) > 1 and sys.argv[1] == "zipf"
ei = make_graph(n, 39561252, dev, zipf=zipf)
be = ops.get_backend()
x = torch.randn(n, C, device=dev)
csr = ops.csr_cache.get(ei, None, n, C * 4)
sl = csr.sliced(0, n, C)
plan = [int(v) for v in sl.plan]
slices, panels, NT, W = plan[0], plan[1], plan[7], plan[4]
trace = None
if variant & 4:
    trace = torch.zeros(slices * panels * 2 * NT * 4, dtype=torch.int64, device=dev)
    os.environ["DIF_SLICED_TRACE"] = str(trace.data_ptr())
ys = be.sliced_prescale(x, csr.rowptr, n, sl.plan)
f = lambda: be.sliced_spmm(sl, ys, csr.rowptr, n, 0, n, C, None, 1.0, 1.0)
out = f()
# reference: the gather kernel on the same CSR
ref = ops.gcn_aggregate(csr, x.view(n, 1, C), None, 1.0, 1.0).view(n, C) if False else None
chk = float(out.double().abs().sum())
for _ in range(5): f()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 20 * 1e3)
print(f"variant {variant}: sliced product {min(ts):.1f} / {np.median(ts):.1f} us (min / median of 5 x 20), checksum {chk:.6e}", flush=True)
if trace is not None:
    tr = trace.cpu().numpy().reshape(slices * panels, 2, NT, 4).astype(np.float64) / 100.0     # 100 MHz -> us
    t_begin = tr[:, :, 0, 0].min()
    wait = tr[..., 1] - tr[..., 0]          # barrier 1: waiting for the slowest wave of the previous tile
    load = tr[..., 2] - tr[..., 1]          # tile load + barrier 2
    comp = tr[..., 3] - tr[..., 2]          # own compute
    print(f"  per tile, mean over workgroups: wait at barrier (first wave / last wave) {wait[:, 0].mean():.2f} / {wait[:, 1].mean():.2f} us, "
          f"tile load {load.mean():.2f} us (p10 {np.percentile(load, 10):.2f}, p90 {np.percentile(load, 90):.2f}), "
          f"compute {comp.mean():.2f} us (p10 {np.percentile(comp, 10):.2f}, p90 {np.percentile(comp, 90):.2f})")
    print("  tile load by tile index (mean us):", np.round(load.mean(axis=(0, 1)), 2))
    print("  barrier wait by tile index, first wave (mean us):", np.round(wait[:, 0].mean(axis=0), 2))
    end = tr[:, :, NT - 1, 3].max(axis=1)
    start = tr[:, :, 0, 0].min(axis=1)
    print(f"  workgroup start spread {start.max() - start.min():.1f} us, end spread {end.max() - end.min():.1f} us, "
          f"span {end.max() - start.min():.1f} us; sum per workgroup: wait {wait.sum(axis=2).mean():.1f} load {load.sum(axis=2).mean():.1f} "
          f"compute {comp.sum(axis=2).mean():.1f} us")
