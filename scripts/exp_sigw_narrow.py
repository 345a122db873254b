"""Sigmoid attention at 64 columns per head: the fp32-chain kernels (training) against the split-bfloat16 plane kernels of
csrc/sigmoid_wide.hip instantiated at KS = 2 (DIF_SIGW_NARROW=1).  python scripts/exp_sigw_narrow.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difformer_amd import autograd_ops as ag, ops
dev = torch.device("cuda:0")
def timeit(fn, reps):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for n, m in ((2708, 64), (19717, 64), (20000, 64), (8192, 64), (20000, 48)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, 64, generator=g)
    q = ((x @ torch.randn(64, m, generator=g)) / 8 * 0.3).reshape(n, 1, m).to(dev)
    k = ((x @ torch.randn(64, m, generator=g)) / 8 * 0.3).reshape(n, 1, m).to(dev)
    v = torch.randn(n, 1, m, generator=g).to(dev)
    go = torch.randn(n, 1, m, generator=g).to(dev)
    be = ops.get_backend()
    with torch.no_grad():
        tf = timeit(lambda: be.sigmoid_attention(q, k, v), 10)
    qd, kd, vd = (a.clone().requires_grad_(True) for a in (q, k, v))
    def step():
        qd.grad = kd.grad = vd.grad = None
        ag.sigmoid_attention(qd, kd, vd).backward(go)
    ts = timeit(step, 5)
    print(f"{os.environ.get('DIF_SIGW_NARROW', '0')}: N={n} M=D={m}: inference forward {tf:.3f} ms; training forward+backward {ts:.3f} ms", flush=True)
