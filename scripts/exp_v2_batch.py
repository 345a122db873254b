"""f4 timing: DIFFormer_v2 forward on particle-shaped batches (physical particle/run.sh: batch_size 1024 / 8192 graphs,
hidden 64, 2 layers), per C-ABI entry point, next to the reference's padded formulation run with torch ops on the
same GPU (autograd_ops._batched_*_expr restates difformer-v2.py:80-135 for the backward pass).
    python scripts/exp_v2_batch.py
"""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difformer_amd import DIFFormer_v2, ops, autograd_ops as ag

dev = torch.device("cuda:0")


def batch(B, lo, hi, seed=0):
    rng = np.random.default_rng(seed)
    n_nodes = rng.integers(lo, hi + 1, size=B)
    offs = np.concatenate([[0], np.cumsum(n_nodes)])
    n = int(offs[-1])
    g = np.repeat(np.arange(B), 3 * n_nodes)
    src = (rng.random(g.shape[0]) * n_nodes[g]).astype(np.int64) + offs[g]
    dst = (rng.random(g.shape[0]) * n_nodes[g]).astype(np.int64) + offs[g]
    loops = np.arange(n)
    ei = np.stack([np.concatenate([src, dst, loops]), np.concatenate([dst, src, loops])])
    return torch.from_numpy(n_nodes).to(dev), torch.from_numpy(ei).to(dev), n


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for name, B, lo, hi, kernel in (("tau3mu-like 8192 x ~20", 8192, 8, 32, "simple"),
                                ("actstrack-like 1024 x ~110", 1024, 40, 180, "simple"),
                                ("plbind-like 256 x ~340", 256, 100, 600, "simple"),
                                ("actstrack-like 1024 x ~110", 1024, 40, 180, "sigmoid"),
                                ("tau3mu-like 8192 x ~20", 8192, 8, 32, "sigmoid")):
    n_nodes, ei, n = batch(B, lo, hi)
    torch.manual_seed(0)
    model = DIFFormer_v2(7, 64, 64, num_layers=2, kernel=kernel).to(dev).eval()
    x = torch.randn(n, 7, device=dev)
    be = ops.get_backend()
    with torch.no_grad():
        ms = timeit(lambda: model(x, ei, n_nodes))
        be.kernel_events = {}
        for _ in range(5):
            model(x, ei, n_nodes)
        torch.cuda.synchronize()
        kt = {k: float(np.mean(v)) * 1e3 for k, v in be.kernel_times_ms().items()}
        be.kernel_events = None
        lay = ops.layout_cache.get(n_nodes, dev)
        q, k, v = (torch.randn(n, 1, 64, device=dev) for _ in range(3))
        expr = (ag._batched_simple_expr if kernel == "simple" else ag._batched_sigmoid_expr)(lay)
        try:
            ref_ms = timeit(lambda: expr(q, k, v), iters=5)
        except RuntimeError as e:                      # the [B,B,max_node,H] tensor of :124 may not fit
            ref_ms = float("nan")
        fwd = ops.batched_simple_attention if kernel == "simple" else ops.batched_sigmoid_attention
        att_ms = timeit(lambda: fwd(q, k, v, lay))
    print(f"{name:28s} {kernel:8s} N={n:7d}  forward {ms:7.3f} ms   attention op {att_ms * 1e3:8.1f} us  "
          f"(padded torch formulation {ref_ms * 1e3:9.1f} us)   per entry point (us): "
          + ", ".join(f"{k[4:]} {v:.0f}" for k, v in sorted(kt.items())))
