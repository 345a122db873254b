"""One shape of the wide-head sigmoid forward (and optionally backward), repeated: the command rocprofv3 wraps.
    python scripts/exp_sigmoid_wide_fwd.py N M [bwd]"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difformer_amd import autograd_ops as ag, ops  # noqa: E402
dev = torch.device("cuda:0")
n, m = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator().manual_seed(0)
x = torch.randn(n, 64, generator=g)
q = ((x @ torch.randn(64, m, generator=g)) / 8 * 0.3).reshape(n, 1, m).to(dev)
k = ((x @ torch.randn(64, m, generator=g)) / 8 * 0.3).reshape(n, 1, m).to(dev)
v = torch.randn(n, 1, m, generator=g).to(dev)
go = torch.randn(n, 1, m, generator=g).to(dev)
be = ops.get_backend()
if len(sys.argv) > 3:
    qd, kd, vd = (a.clone().requires_grad_(True) for a in (q, k, v))
    for _ in range(10):
        qd.grad = kd.grad = vd.grad = None
        ag.sigmoid_attention(qd, kd, vd).backward(go)
else:
    with torch.no_grad():
        for _ in range(20):
            be.sigmoid_attention(q, k, v)
torch.cuda.synchronize()
