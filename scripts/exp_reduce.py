"""simple_reduce_kernel (stage 1 of the simple attention; Gram pass of the wide closed form) at the shapes that use it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops
dev = torch.device("cuda:0")
be = ops.get_backend()


def timed(f, reps=100):
    for _ in range(10): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps // 10): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for n, c in ((50000, 300), (50000, 400), (100000, 192)):
    x = torch.randn(n, c, device=dev)
    print(f"gram_sym {n} x {c}: {timed(lambda: be.gram_sym(x)):.1f} us", flush=True)
for n, h, d in ((132534, 1, 64), (100000, 1, 128), (100000, 2, 64), (2708, 1, 64)):
    q, k, v = (torch.randn(n, h, d, device=dev) for _ in range(3))
    print(f"simple_reduce {n} x {h} x {d}: {timed(lambda: be.simple_reduce(q, k, v)):.1f} us", flush=True)
