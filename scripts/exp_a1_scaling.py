"""Fixed vs per-row cost of the a1 kernels: back-to-back launches at several N (kernel time ~ a + b*N)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difformer_amd import ops
dev = torch.device("cuda:0")
be = ops.get_backend()
W = [torch.randn(64, 64, device=dev) / 8 for _ in range(3)]
b = [torch.randn(64, device=dev) * 0.1 for _ in range(3)]


def timeit(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


for n in (4096, 16384, 32768, 65536, 131072, 262144, 524288):
    x = torch.randn(n, 64, device=dev)
    q, v, rec = be.project_reduce(x, W[0], b[0], W[1], b[1], W[2], b[2], 1, 64)
    k = torch.randn(n, 1, 64, device=dev)
    t_pr = timeit(lambda: be.project_reduce(x, W[0], b[0], W[1], b[1], W[2], b[2], 1, 64))
    t_red = timeit(lambda: be.simple_reduce(q, k, v))
    t_ap = timeit(lambda: be.simple_apply(q, rec, n, 64))
    t_tail = timeit(lambda: be.layer_tail(q, None, x, 0.5, b[0], b[1], 1e-5))
    print(f"n={n:7d}  project_reduce(+finalize) {t_pr:7.1f} us   simple_reduce(+finalize) {t_red:7.1f} us   "
          f"apply {t_ap:6.1f} us   layer_tail {t_tail:6.1f} us", flush=True)
