"""Backward of the simple attention for heads wider than 64: HIP kernels (prep + reduce + wide row-GEMMs) against the
tensor-op re-derivation that used to serve those shapes (autograd_ops._grad_by_recompute over _simple_expr).
    python scripts/exp_wide_bwd.py            (on the GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difformer_amd import autograd_ops as ag, ops

dev = torch.device("cuda:0")
be = ops.get_backend()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for n, d in ((10000, 128), (132534, 128), (1632803, 128), (50000, 300), (132534, 64)):
    g = torch.Generator().manual_seed(0)
    q, k, v, go = (torch.randn(n, 1, d, generator=g).to(dev) for _ in range(4))
    reduced = be.simple_reduce(q, k, v)
    out = be.simple_apply(q, reduced, n, d)
    hip = timed(lambda: be.simple_backward(q, k, v, reduced, out, go))
    ten = timed(lambda: ag._grad_by_recompute(ag._simple_expr, (q, k, v), go))
    a = be.simple_backward(q, k, v, reduced, out, go)
    b = ag._grad_by_recompute(ag._simple_expr, (q, k, v), go)
    err = max(float((x - y).abs().max() / y.abs().max()) for x, y in zip(a, b))
    be.kernel_events = {}
    be.simple_backward(q, k, v, reduced, out, go)
    torch.cuda.synchronize()
    kt = be.kernel_times_ms()
    be.kernel_events = None
    print(f"n={n} d={d}: hip {hip:.1f} us, tensor ops {ten:.1f} us, max diff {err:.2e}; "
          + ", ".join(f"{k_} {'/'.join(f'{t * 1e3:.0f}' for t in v_)}" for k_, v_ in kt.items()), flush=True)
