"""Narrow Linear kernel (csrc/skinny_linear.hip) on the shapes the models use; GB/s = algorithmic bytes / time."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difformer_amd import ops

dev = torch.device("cuda:0")
be = ops.get_backend()


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


for n in (163596, 132534):
    for ci, co, ln in ((7, 64, True), (8, 64, True), (64, 192, False), (64, 64, False), (64, 112, False), (64, 64, True)):
        x = torch.randn(n, ci, device=dev)
        w = torch.randn(co, ci, device=dev)
        b = torch.randn(co, device=dev)
        lw = torch.randn(co, device=dev) if ln else None
        us = timeit(lambda: be.linear(x, w, b, lw, lw, 1e-5, ln))
        gb = n * (ci + co) * 4 / us / 1e3
        print(f"n={n} {ci:3d}->{co:3d} ln={int(ln)}: {us:7.1f} us  {gb:7.0f} GB/s")
