"""Experiment: blocked SpMM on C4 under the conditions of the model (strided V view, attn operand, fused tail)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops
from bench import make_graph
dev = torch.device("cuda:0")
n = 132534
ei = make_graph(n, 39561252, dev)
e = ei.shape[1]
be = ops.get_backend()
qkv = torch.randn(n, 192, device=dev)
xc = torch.randn(n, 64, device=dev)
attn = torch.randn(n, 64, device=dev)
prev = torch.randn(n, 64, device=dev)
w, b = torch.rand(64, device=dev), torch.rand(64, device=dev)
csr = ops.GraphCSR.build(ei, None, n, 13)
def run(name, x, a, tail):
    f = lambda: be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, e, x, 0, n, a, 1.0, 1.0, tail)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{name:40s} {dt*1e3:.3f} ms", flush=True)
tail = dict(x0=None, prev=prev, alpha=0.5, ln_weight=w, ln_bias=b, eps=1e-5)
for rep in range(2):
    run("contiguous x", xc, None, None)
    run("contiguous x + attn", xc, attn, None)
    run("contiguous x + attn + tail", xc, attn, tail)
    run("strided x (ld 192)", qkv[:, 128:], None, None)
    run("strided x + attn + tail", qkv[:, 128:], attn, tail)
