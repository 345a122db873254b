"""Experiment: time (and check) the feature-sliced product of whatever build DIFFORMER_HIP_LIB points at, C4 shape.
    python scripts/exp_sliced_variant.py [zipf]
    DIFFORMER_HIP_LIB=difformer_amd/lib/libdifformer_hip_half.so python scripts/exp_sliced_variant.py [zipf]
(half = two 8-wave workgroups per CU with 80 KiB of LDS each: -DDIF_SLICED_TILE_ROWS=5104 -DDIF_SLICED_MAX_WAVES=8
-DDIF_SLICED_WG_PER_CU=2; profiles/r03_experiments.md)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops, _lib
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
ys = be.sliced_prescale(x, csr.rowptr, n, sl.plan)
f = lambda: be.sliced_spmm(sl, ys, csr.rowptr, n, 0, n, C, None, 1.0, 1.0)
out = f()
ref = be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, csr.nnz, x, 0, n)
err = float((out - ref).abs().max() / ref.abs().max())
for _ in range(10): f()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 20 * 1e3)
blocks = int(sl.entries.numel()) // 512
print(f"{os.path.basename(_lib.LIB_PATH)} {'zipf' if zipf else 'uniform'}: plan {plan}, {blocks} blocks = "
      f"{blocks * 512 / csr.nnz:.3f} padded lane-steps per entry, product {min(ts):.1f} / {sorted(ts)[2]:.1f} us (min / median of 5x20), "
      f"max diff vs the gather kernel {err:.2e}", flush=True)
