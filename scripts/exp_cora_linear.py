"""Input layer of the Cora configs (2,708 x 1,433 -> 64, LayerNorm, ReLU): one launch of the K-split kernel against the vendor
GEMM + tail pass it replaces (profiles/r04_experiments.md section 3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops
dev = torch.device("cuda:0")
be = ops.get_backend()


def timed(f, reps=200):
    for _ in range(20): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps // 20): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for n, ci in ((2708, 1433), (2708, 1432), (19717, 500), (10000, 512), (16000, 300), (50000, 512), (100000, 1433)):
    x = torch.randn(n, ci, device=dev)
    W, b = torch.randn(64, ci, device=dev) / ci ** 0.5, torch.randn(64, device=dev)
    lw, lb = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)
    t1 = timed(lambda: be.linear(x, W, b, lw, lb, 1e-5, True))
    t2 = timed(lambda: be.layer_tail(torch.nn.functional.linear(x, W, b).unsqueeze(1), None, None, 0.5, lw, lb, 1e-5, True))
    print(f"{n} x {ci} -> 64 + LayerNorm + ReLU: hand-written {t1:.1f} us, vendor GEMM + tail {t2:.1f} us "
          f"({(n * ci + 64 * ci + n * 64) * 4 / t1 / 1e6:.2f} TB/s)", flush=True)
