"""Experiment: sigmoid attention backward (csrc/sigmoid_attn_bwd.hip) against re-deriving the gradient with tensor ops
(which materialises the [N,L,H] score tensor several times).   python scripts/exp_sigmoid_bwd.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops, autograd_ops as ag

dev = torch.device("cuda:0")
be = ops.get_backend()


def timeit(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


for n in (2708, 8192, 20000):
    q, k, v, g = (torch.randn(n, 1, 64, device=dev) * 0.5 for _ in range(4))
    out, den = be.sigmoid_attention(q, k, v, want_den=True)
    t_f = timeit(lambda: be.sigmoid_attention(q, k, v, want_den=True))
    t_b = timeit(lambda: be.sigmoid_backward(q, k, v, out, den, g))
    t_r = timeit(lambda: ag._grad_by_recompute(ag._sigmoid_expr, (q, k, v), g), it=5) if n <= 8192 else float("nan")
    fl = 14.0 * n * n * 64
    print(f"N = L = {n}: forward {t_f:.0f} us, backward kernel {t_b:.0f} us ({fl / t_b / 1e6:.1f} TFLOP/s fp32 MFMA), "
          f"tensor-op recompute {t_r:.0f} us", flush=True)
