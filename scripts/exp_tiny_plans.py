"""One `sigmoid` case of tests/test_gpu_tiny.py under both launch plans: per-parameter gradient error against the float64 oracle
(and the float32 CPU run of the same oracle), to tell a plan's bug from the case's own cancellation."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_tiny as T  # noqa: E402
from conftest import grad_err  # noqa: E402
from difformer_amd import DIFFormer, tiny  # noqa: E402
og = T.og

want = sys.argv[1] if len(sys.argv) > 1 else "sigmoid-n513-d3-L4-s7"
c = [c for c in T._cases() if T._ID(c) == want][0]
dev = torch.device("cuda:0")
n, d, L = c["n"], c["hidden"], c["layers"]
res = {}
for plan in (1, 2):
    tiny.PLAN = plan
    torch.manual_seed(100 + c["seed"])
    model = DIFFormer(c["f_in"], d, c["c"], num_layers=L, num_heads=1, kernel=c["kernel"], alpha=c["alpha"], dropout=0.0,
                      use_bn=c["use_bn"], use_residual=c["use_residual"], use_weight=c["use_weight"], use_graph=c["use_graph"],
                      graph_weight=c["graph_weight"], use_source=c["use_source"])
    with torch.no_grad():
        for bn in model.bns:
            bn.weight.add_(0.2 * torch.randn(bn.weight.shape))
            bn.bias.add_(0.2 * torch.randn(bn.bias.shape))
    g = torch.Generator().manual_seed(c["seed"])
    x = torch.randn(n, c["f_in"], generator=g)
    iso = min(c["iso"], n - 1)
    ei = T._graph(n, n * c["deg"], seed=c["seed"], isolated=iso)
    ei = torch.cat([ei, torch.arange(n - iso).repeat(2, 1)], dim=1)
    w = (torch.rand(ei.shape[1], generator=g) * 2 + 0.05) if c["weighted"] else None
    go = torch.randn(n, c["c"], generator=g)
    cfg = dict(in_channels=c["f_in"], hidden_channels=d, out_channels=c["c"], num_layers=L, num_heads=1, kernel=c["kernel"],
               alpha=c["alpha"], use_bn=c["use_bn"], use_residual=c["use_residual"], use_weight=c["use_weight"],
               use_graph=c["use_graph"], graph_weight=c["graph_weight"], use_source=c["use_source"])
    model = model.to(dev).train()
    xd = x.to(dev).requires_grad_(True)
    out = model(xd, ei.to(dev) if c["use_graph"] else None, w.to(dev) if (w is not None and c["use_graph"]) else None)
    out.backward(go.to(dev))
    res[plan] = {k: p.grad.cpu().numpy() for k, p in model.named_parameters() if p.grad is not None}
    res[plan]["out"] = out.detach().cpu().numpy()
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
p64 = og.leaves(sd)
x64 = x.double().requires_grad_(True)
ref = og.difformer_forward(p64, x64, ei if c["use_graph"] else None, None if w is None else w.double(), cfg)
ref.backward(go.double())
p32 = og.leaves(sd, torch.float32)
ref32 = og.difformer_forward(p32, x.clone(), ei if c["use_graph"] else None, w, cfg)
ref32.backward(go)
grads = {k: v.grad for k, v in p64.items() if v.grad is not None}
gmax = max(float(v.abs().max()) for v in grads.values())
print(f"# {want}: gmax {gmax:.3e}")
print(f"{'parameter':28s} {'|ref|max':>10s} {'one-wg':>10s} {'grid':>10s} {'cpu-f32':>10s} {'grid-vs-one':>12s}")
for k, r in grads.items():
    r = r.numpy()
    e1, e2 = grad_err(res[1][k], r, gmax, 2e-6), grad_err(res[2][k], r, gmax, 2e-6)
    e32 = grad_err(p32[k].grad.numpy(), r, gmax, 2e-6)
    print(f"{k:28s} {np.abs(r).max():10.3e} {e1:10.2e} {e2:10.2e} {e32:10.2e} {np.abs(res[1][k] - res[2][k]).max():12.3e}")
