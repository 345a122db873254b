import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, numpy as np
from difformer_amd import autograd_ops as ag
from conftest import rel_err
dev=torch.device('cuda:0')
for n,h,d in [(300,1,64),(50000,1,64),(132534,1,64)]:
    g=torch.Generator().manual_seed(n+d)
    q,k,v=(torch.randn(n,h,d,generator=g) for _ in range(3)); go=torch.randn(n,h,d,generator=g)
    qd,kd,vd=(t.to(dev).requires_grad_(True) for t in (q,k,v))
    ag.simple_attention(qd,kd,vd).backward(go.to(dev))
    q64,k64,v64=(t.double().requires_grad_(True) for t in (q,k,v))
    ag._simple_expr(q64,k64,v64).backward(go.double())
    q32,k32,v32=(t.to(dev).requires_grad_(True) for t in (q,k,v))
    ag._simple_expr(q32,k32,v32).backward(go.to(dev))
    for a,b,c,nm in ((qd.grad,q32.grad,q64.grad,'dq'),(kd.grad,k32.grad,k64.grad,'dk'),(vd.grad,v32.grad,v64.grad,'dv')):
        print(n,nm,'hip',rel_err(a.cpu().numpy(),c.numpy()),'torch32',rel_err(b.cpu().numpy(),c.numpy()))
