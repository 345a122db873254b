"""Backward of the batched simple attention (DIFFormer_v2): three raw launches of the forward kernel against the
tensor-op recompute on the padded batch it replaces."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import autograd_ops as ag, ops
dev = torch.device("cuda:0")
be = ops.get_backend()


def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B, lo, hi, H, D in ((8192, 10, 40, 1, 64), (8192, 10, 40, 4, 16), (512, 100, 400, 1, 64)):
    g = torch.Generator().manual_seed(B)
    n_nodes = torch.randint(lo, hi + 1, (B,), generator=g)
    n = int(n_nodes.sum())
    q, k, v, go = (torch.randn(n, H, D, generator=g).to(dev) for _ in range(4))
    layout = ops.BatchLayout(n_nodes, dev)
    out, den, sumsq = be.batched_simple_attention(q, k, v, layout.graph_ptr, want_den=True)
    t_f = t(lambda: be.batched_simple_attention(q, k, v, layout.graph_ptr, want_den=True))
    t_hip = t(lambda: be.batched_simple_backward(q, k, v, out, den, sumsq, go, layout.graph_ptr))
    t_old = t(lambda: ag._grad_by_recompute(ag._batched_simple_expr(layout), (q, k, v), go), n=5)
    a = be.batched_simple_backward(q, k, v, out, den, sumsq, go, layout.graph_ptr)
    b = ag._grad_by_recompute(ag._batched_simple_expr(layout), (q, k, v), go)
    err = max(float((x - y).abs().max() / y.abs().max()) for x, y in zip(a, b))
    print(f"{B} graphs, {n} nodes, H={H}, D={D}: forward {t_f:.3f} ms, backward kernels {t_hip:.3f} ms, tensor-op recompute {t_old:.3f} ms, diff {err:.1e}")
