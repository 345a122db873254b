"""Experiment: how does the SpMM time depend on the span of source rows touched (L2 / MALL residency)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops
dev = torch.device("cuda:0")
n, e = 132534, 79255038
be = ops.get_backend()
x = torch.randn(n, 64, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
dst = torch.randint(0, n, (e,), generator=g, device=dev)
for span in (2048, 8192, 16384, 32768, 65536, n):
    src = torch.randint(0, span, (e,), generator=g, device=dev)
    ei = torch.stack([src, dst])
    csr = ops.GraphCSR.build(ei, None, n)
    for _ in range(2):
        be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, e, x, 0, n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, e, x, 0, n)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"span {span:7d} rows ({span*256/2**20:6.1f} MiB): {dt*1e3:.3f} ms  gather {e*256/dt/1e12:.2f} TB/s", flush=True)
    del csr, ei, src
