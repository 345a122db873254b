"""Time of the forward entry point (packs + sweep + combine) at one shape, by HIP events: python scripts/exp_sigw_time.py N M"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difformer_amd import ops
dev = torch.device("cuda:0")
n, m = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator().manual_seed(0)
x = torch.randn(n, 64, generator=g)
q = ((x @ torch.randn(64, m, generator=g)) / 8 * 0.3).reshape(n, 1, m).to(dev)
k = ((x @ torch.randn(64, m, generator=g)) / 8 * 0.3).reshape(n, 1, m).to(dev)
v = torch.randn(n, 1, m, generator=g).to(dev)
be = ops.get_backend()
with torch.no_grad():
    for _ in range(3):
        be.sigmoid_attention(q, k, v)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        be.sigmoid_attention(q, k, v)
    e1.record(); torch.cuda.synchronize()
print(f"{os.environ.get('DIFFORMER_HIP_LIB', 'default').split('_')[-1]:>12}: N={n} M={m} forward {e0.elapsed_time(e1) / 20:.3f} ms")
