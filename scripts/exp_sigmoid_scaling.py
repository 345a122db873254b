"""a2 (sigmoid attention) at growing N: time and fp32-MFMA rate (4*N*L*H*D FLOP; peak 157 TF/s)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difformer_amd import ops
dev = torch.device("cuda:0")
be = ops.get_backend()
for n in (2708, 8192, 20000, 50000):
    q, k, v = (torch.randn(n, 1, 64, device=dev) * 0.3 for _ in range(3))
    for _ in range(3): be.sigmoid_attention(q, k, v)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    it = 20 if n <= 20000 else 5
    for _ in range(it): be.sigmoid_attention(q, k, v)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
    fl = 4.0 * n * n * 64
    print(f"N={n:6d}: {dt * 1e6:9.1f} us   {fl / dt / 1e12:6.1f} TF/s  ({fl / dt / 157e12 * 100:4.1f} % of the fp32 MFMA peak)", flush=True)
