"""Input layer + Gram + slice-major copy: the fused kernel against the three it replaces (C4 row count)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops
dev = torch.device("cuda:0")
be = ops.get_backend()
n, d = 132534, 64
plan = be.sliced_plan(n, n, d)
rowptr = torch.arange(n + 1, dtype=torch.int32, device=dev) * 600


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


for c_in in (8, 32, 64):
    x = torch.randn(n, c_in, device=dev)
    W, b = torch.randn(d, c_in, device=dev), torch.randn(d, device=dev)
    lw, lb = torch.rand(d, device=dev) + 0.5, torch.randn(d, device=dev)
    t_fused = timed(lambda: be.input_gram(x, W, b, lw, lb, 1e-5, True, rowptr, plan))
    t_nocopy = timed(lambda: be.input_gram(x, W, b, lw, lb, 1e-5, True))
    t_lin = timed(lambda: be.linear(x, W, b, lw, lb, 1e-5, True))
    h = be.linear(x, W, b, lw, lb, 1e-5, True)
    t_gram = timed(lambda: be.gram(h, rowptr, plan))
    print(f"{n} x {c_in} -> {d}: fused {t_fused:.1f} us (without the copy {t_nocopy:.1f}); linear {t_lin:.1f} + gram with copy {t_gram:.1f} "
          f"= {t_lin + t_gram:.1f} us", flush=True)
