"""Experiment: where a launch of the sliced product spends its time, tile by tile.  A measurement build of the library
(-DDIF_SLICED_TRACE) stamps the wall clock (100 MHz) of the first and the last wave of every workgroup at four points of
every tile: before the barrier that ends the previous tile, after it, after the tile load, after the own compute.
    make -C difformer_amd/csrc EXTRA=-DDIF_SLICED_TRACE OBJDIR=/tmp/obj_trace OUT=$PWD/scripts/bin/libdifformer_hip_trace.so
    DIFFORMER_HIP_LIB=scripts/bin/libdifformer_hip_trace.so python scripts/exp_sliced_trace.py [zipf]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
sl = csr.sliced(0, n, C)
plan = [int(v) for v in sl.plan]
slices, panels, NT, W = plan[0], plan[1], plan[7], plan[4]
trace = torch.zeros(slices * panels * 2 * NT * 4, dtype=torch.int64, device=dev)
os.environ["DIF_SLICED_TRACE"] = str(trace.data_ptr())
ys = be.sliced_prescale(x, csr.rowptr, n, sl.plan)
f = lambda: be.sliced_spmm(sl, ys, csr.rowptr, n, 0, n, C, None, 1.0, 1.0)
for _ in range(5): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): f()
b.record(); torch.cuda.synchronize()
print(f"sliced product (trace build) {a.elapsed_time(b) / 20 * 1e3:.1f} us per launch, plan {plan}", flush=True)
tr = trace.cpu().numpy().reshape(slices * panels, 2, NT, 4).astype(np.float64) / 100.0     # 100 MHz -> us
wait = tr[..., 1] - tr[..., 0]          # barrier: waiting for the slowest wave of the previous tile
load = tr[..., 2] - tr[..., 1]          # rest of the tile load + second barrier
comp = tr[..., 3] - tr[..., 2]          # own compute
print(f"per tile, mean over workgroups: wait at the barrier (first wave / last wave) {wait[:, 0].mean():.2f} / {wait[:, 1].mean():.2f} us, "
      f"tile load {load.mean():.2f} us (p10 {np.percentile(load, 10):.2f}, p90 {np.percentile(load, 90):.2f}), "
      f"compute {comp.mean():.2f} us (p10 {np.percentile(comp, 10):.2f}, p90 {np.percentile(comp, 90):.2f})")
print("tile load by step (mean us):", np.round(load.mean(axis=(0, 1)), 2))
print("barrier wait of the first wave by step (mean us):", np.round(wait[:, 0].mean(axis=0), 2))
end = tr[:, :, NT - 1, 3].max(axis=1)
start = tr[:, :, 0, 0].min(axis=1)
print(f"workgroup start spread {start.max() - start.min():.1f} us, end spread {end.max() - end.min():.1f} us, span "
      f"{end.max() - start.min():.1f} us; sums per workgroup: wait {wait.sum(axis=2).mean():.1f}, load {load.sum(axis=2).mean():.1f}, "
      f"compute {comp.sum(axis=2).mean():.1f} us")
