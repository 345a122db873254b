"""Experiment: L2-resident gather throughput of the wave-per-row SpMM vs feature width F (bytes- or request-bound?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops
dev = torch.device("cuda:0")
n, e = 132534, 40000000
be = ops.get_backend()
g = torch.Generator(device=dev).manual_seed(0)
dst = torch.randint(0, n, (e,), generator=g, device=dev)
for F in (16, 32, 64, 128, 256):
    span = max(256, int(2 * 2**20 / (F * 4)))      # 2 MiB of sources
    src = torch.randint(0, span, (e,), generator=g, device=dev)
    ei = torch.stack([src, dst])
    csr = ops.GraphCSR.build(ei, None, n)
    x = torch.randn(n, F, device=dev)
    for _ in range(2):
        be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, e, x, 0, n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, e, x, 0, n)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"F {F:4d} ({F*4:5d} B rows, span {span} rows): {dt*1e3:.3f} ms  gather {e*F*4/dt/1e12:.2f} TB/s  {e/dt/1e9:.1f} Grows/s", flush=True)
    del csr, ei, src, x
