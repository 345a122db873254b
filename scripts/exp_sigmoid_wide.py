"""Wide-head sigmoid attention (csrc/sigmoid_wide.hip) at the image-and-text scripts' sizes: forward, forward + backward, and the
paths it replaces (fp32-MFMA generic forward kernel; tensor-op gradient) under ops.set_exact_fp32(True).
    python scripts/exp_sigmoid_wide.py [--old]"""
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difformer_amd import autograd_ops as ag, ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def case(n, m, reps=10, old=False):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, 64, generator=g)
    q = ((x @ torch.randn(64, m, generator=g)) / 8 * 0.3).reshape(n, 1, m).to(dev)
    k = ((x @ torch.randn(64, m, generator=g)) / 8 * 0.3).reshape(n, 1, m).to(dev)
    v = torch.randn(n, 1, m, generator=g).to(dev)
    go = torch.randn(n, 1, m, generator=g).to(dev)
    be = ops.get_backend()
    with torch.no_grad():
        tf = timeit(lambda: be.sigmoid_attention(q, k, v), reps)
    qd, kd, vd = (a.clone().requires_grad_(True) for a in (q, k, v))

    def step():
        qd.grad = kd.grad = vd.grad = None
        ag.sigmoid_attention(qd, kd, vd).backward(go)
    ts = timeit(step, max(2, reps // 2))
    fl = 4.0 * n * n * m
    print(f"N={n} M=D={m} {'old' if old else 'new'}: forward {tf:.3f} ms = {fl / tf / 1e9:.1f} TFLOP/s (4NLD); forward+backward {ts:.3f} ms "
          f"(backward {ts - tf:.3f} ms = {14.0 * n * n * m / (ts - tf) / 1e9:.1f} TFLOP/s on 14NLD)", flush=True)


if __name__ == "__main__":
    old = "--old" in sys.argv
    if old:
        ops.set_exact_fp32(True)
    for n, m in ((15000, 300), (13000, 400), (18846, 300), (15000, 128), (15000, 512), (4000, 300)):
        case(n, m, reps=3 if old else 10, old=old)
