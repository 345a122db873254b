"""Per-tensor errors of the tiny path for chosen cases of tests/test_gpu_tiny.py (debug aid)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_tiny as T
from conftest import grad_err, rel_err
from oracle import difformer_oracle_grad as og
from difformer_amd import DIFFormer

dev = torch.device("cuda:0")
want = set(int(v) for v in sys.argv[1:])
for c in T._cases():
    if c["seed"] not in want:
        continue
    print(c)
    n, d, L = c["n"], c["hidden"], c["layers"]
    torch.manual_seed(100 + c["seed"])
    model = DIFFormer(c["f_in"], d, c["c"], num_layers=L, num_heads=1, kernel=c["kernel"], alpha=c["alpha"], dropout=c["dropout"],
                      use_bn=c["use_bn"], use_residual=c["use_residual"], use_weight=c["use_weight"], use_graph=c["use_graph"],
                      graph_weight=c["graph_weight"], use_source=c["use_source"])
    with torch.no_grad():
        for bn in model.bns:
            bn.weight.add_(0.2 * torch.randn(bn.weight.shape)); bn.bias.add_(0.2 * torch.randn(bn.bias.shape))
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
    eid, wd = ei.to(dev), (None if w is None else w.to(dev))
    masks = None
    if c["dropout"] > 0:
        torch.manual_seed(4242)
        state = torch.cuda.get_rng_state(dev)
        rnd = torch.rand((L + 1, n, d), device=dev)
        torch.cuda.set_rng_state(state, dev)
        masks = ((rnd >= c["dropout"]).double() / (1.0 - c["dropout"])).cpu()
    out = model(xd, eid if c["use_graph"] else None, wd if c["use_graph"] else None)
    out.backward(go.to(dev))
    p64 = og.leaves({k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
    x64 = x.double().requires_grad_(True)
    ref = T._oracle(p64, x64, ei if c["use_graph"] else None, None if w is None else w.double(), cfg, masks)
    ref.backward(go.double())
    print("  out", rel_err(out.detach().cpu().numpy(), ref.detach().numpy()))
    grads = {k: v.grad for k, v in p64.items()}
    gmax = max([float(v.abs().max()) for v in grads.values() if v is not None] + [1e-30])
    print("  gmax", gmax, " dx", grad_err(xd.grad.cpu().numpy(), x64.grad.numpy(), gmax), float(x64.grad.abs().max()))
    for k, prm in model.named_parameters():
        if grads[k] is None:
            continue
        print(f"  {k:24s} err {grad_err(prm.grad.cpu().numpy(), grads[k].numpy(), gmax):.3e}  max|ref| {float(grads[k].abs().max()):.3e}  max|diff| {float((prm.grad.cpu().double()-grads[k]).abs().max()):.3e}")
