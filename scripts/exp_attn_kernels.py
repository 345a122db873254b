"""Micro-benchmark of the simple-attention side kernels at C4 size (project_reduce, reduce, apply, tail)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops
dev = torch.device("cuda:0")
n, d = 132534, 64
be = ops.get_backend()
x = torch.randn(n, d, device=dev)
W = [torch.randn(d, d, device=dev) / 8 for _ in range(3)]
b = [torch.randn(d, device=dev) for _ in range(3)]
def bench(name, f, iters=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    print(f"{name:28s} {dt*1e6:8.1f} us", flush=True)
q, v, rec = be.project_reduce(x, W[0], b[0], W[1], b[1], W[2], b[2], 1, d)
# correctness vs torch
qr = torch.nn.functional.linear(x, W[0], b[0]); kr = torch.nn.functional.linear(x, W[1], b[1]); vr = torch.nn.functional.linear(x, W[2], b[2])
print("q err", float((q[:, 0] - qr).abs().max() / qr.abs().max()), "v err", float((v[:, 0] - vr).abs().max() / vr.abs().max()))
rec2 = be.simple_reduce(qr.view(n, 1, d), kr.view(n, 1, d), vr.view(n, 1, d))
print("record err", float((rec - rec2).abs().max() / rec2.abs().max()))
bench("project_reduce", lambda: be.project_reduce(x, W[0], b[0], W[1], b[1], W[2], b[2], 1, d))
q3, k3, v3 = qr.view(n, 1, d), kr.view(n, 1, d), vr.view(n, 1, d)
bench("simple_reduce", lambda: be.simple_reduce(q3, k3, v3))
bench("simple_apply", lambda: be.simple_apply(q3, rec, n, d))
bench("layer_tail", lambda: be.layer_tail(q3, None, vr, 0.5, b[0], b[1], 1e-5))
wcat = torch.cat(W); bcat = torch.cat(b)
bench("torch fused qkv GEMM", lambda: torch.nn.functional.linear(x, wcat, bcat))
